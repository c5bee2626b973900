import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmx_compressor_amd as d, oracle as O
from _data import make
dev = torch.device("cuda:0")
def show(tag, x, got, want, n=6):
    g, w = got.cpu().float().flatten(), want.cpu().float().flatten()
    bad = (g.view(torch.int32) != w.view(torch.int32)).nonzero().flatten()
    print(tag, "mismatches", len(bad), "of", g.numel())
    for i in bad[:n].tolist():
        print("   idx", i, "x", float(x.flatten()[i]), "got", float(g[i]), "want", float(w[i]))
x = make("normal", (37, 50, 3), seed=21, dtype=torch.float16)
sc, zp = torch.tensor([0.0173]), torch.tensor([3])
show("affine fp16 per-tensor", x, d.ops.fixed_qdq(x.to(dev), 8, 0, True, False, scale=sc, zero_point=zp), O.fixed_point_affine_cast(x, 8, 0, True, False, sc, zp).half())
g = torch.Generator().manual_seed(1)
sc = torch.rand(37, generator=g) * 0.05 + 1e-3; zp = torch.randint(-5, 6, (37,), generator=g)
show("affine fp16 ch0", x, d.ops.fixed_qdq(x.to(dev), 8, 0, True, True, scale=sc, zero_point=zp, ch_axis=0), O.fixed_point_affine_cast(x, 8, 0, True, True, sc, zp, ch_axis=0).half())
show("affine fp16 ch0 ->f32", x, d.ops.fixed_qdq(x.to(dev), 8, 0, True, True, scale=sc, zero_point=zp, ch_axis=0, out_dtype=torch.float32), O.fixed_point_affine_cast(x, 8, 0, True, True, sc, zp, ch_axis=0))
