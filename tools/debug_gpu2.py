import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmx_compressor_amd as d, oracle as O
from _data import make
dev = torch.device("cuda:0")
def show(tag, x, got, want, n=5, B=16):
    g, w = got.cpu().float().flatten(), want.cpu().float().flatten()
    bad = ((g.view(torch.int32) != w.view(torch.int32)) & ~(torch.isnan(g) & torch.isnan(w))).nonzero().flatten()
    print(tag, "mismatches", len(bad), "of", g.numel())
    xf = x.float().flatten()
    for i in bad[:n].tolist():
        b0 = i // B * B
        print("   idx", i, "x", float(xf[i]), "got", float(g[i]), "want", float(w[i]), "blockmax", float(xf[b0:b0+B].abs().max()))
x = make("mixed", (64, 512), seed=8*131+64, dtype=torch.float32, block=64)
show("asym fp32 B64 wl8", x, d.ops.bfp_qdq(x.to(dev), 8, 64, -1, False), O.bfp_cast(x, 8, 64, -1, False), B=64)
x = make("denormal", (128, 1024), seed=3, dtype=torch.float32, block=16)
show("denormal fp32", x, d.ops.bfp_qdq(x.to(dev), 8, 16), O.bfp_cast(x, 8, 16))
bits = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16); vals = bits.view(torch.bfloat16); vals = vals[torch.isfinite(vals.float())]
for top in [1.0, 2.0**-126, 3e-39, 1e-41, 2.0**100, 2.0**112]:
    t = torch.tensor(top).bfloat16(); keep = vals[vals.float().abs() <= float(t.float())]
    n = (keep.numel() + 14)//15; blk = torch.zeros(n,16,dtype=torch.bfloat16); blk[:,0]=t
    flat = torch.zeros(n*15, dtype=torch.bfloat16); flat[:keep.numel()] = keep; blk[:,1:] = flat.reshape(n,15)
    show(f"exh bf16 wl4 top {top}", blk, d.ops.bfp_qdq(blk.to(dev), 4, 16), O.bfp_cast(blk, 4, 16).bfloat16(), n=3)
