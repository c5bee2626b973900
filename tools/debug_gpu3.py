import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmx_compressor_amd as d, oracle as O
dev = torch.device("cuda:0")
bits = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16); vals = bits.view(torch.bfloat16); vals = vals[torch.isfinite(vals.float())]
solo = torch.zeros(vals.numel(), 16, dtype=torch.bfloat16); solo[:, 3] = vals
for wl in (4, 8, 11):
    got = d.ops.bfp_qdq(solo.to(dev), wl, 16).cpu().float(); want = O.bfp_cast(solo, wl, 16).bfloat16().float()
    bad = ((got.view(torch.int32) != want.view(torch.int32)) & ~(torch.isnan(got) & torch.isnan(want))).nonzero()
    print("wl", wl, "bad", len(bad))
    for r, c in bad[:8].tolist():
        print("   row", r, "col", c, "x", float(solo[r, 3]), hex(int(solo[r,3].view(torch.int16)) & 0xFFFF), "got", float(got[r, c]), "want", float(want[r, c]))
