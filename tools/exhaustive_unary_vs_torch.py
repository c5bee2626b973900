"""tools/exhaustive_unary_vs_torch.py -- every one of the 65536 bf16 / fp16 bit patterns through this library's exact-function
kernels (approximator algorithm "dmxq") and through torch's own GPU ops: how many patterns differ, and by how many code points.
Output: profiles/r02_unary_vs_torch_exhaustive.txt"""
import os, sys, torch
import torch.nn.functional as F
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R)
import dmx_compressor_amd as dmx
dev = torch.device('cuda:0')
for dt in (torch.bfloat16, torch.float16):
    x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dt).repeat(2).to(dev)
    for name, mine, ref in (("silu", lambda t: dmx.ops.silu(t), lambda t: F.silu(t)),
                            ("gelu", lambda t: dmx.ops.gelu(t), lambda t: F.gelu(t)),
                            ("gelu_tanh", lambda t: dmx.ops.gelu(t, approximate="tanh"), lambda t: F.gelu(t, approximate="tanh")),
                            ("quick_gelu", lambda t: dmx.ops.quick_gelu(t), lambda t: t * torch.sigmoid(1.702 * t))):
        try:
            a, b = mine(x), ref(x)
        except Exception as e:
            print(dt, name, "ERR", type(e).__name__, str(e)[:80]); continue
        nan_both = torch.isnan(a) & torch.isnan(b)
        diff = (a.view(torch.int16) != b.view(torch.int16)) & ~nan_both
        n = int(diff[:65536].sum())
        worst = 0
        if n:
            ai, bi = a.view(torch.int16)[:65536].int(), b.view(torch.int16)[:65536].int()
            worst = int((ai - bi).abs()[diff[:65536]].max())
        print(dt, name, "differing patterns:", n, "max bit distance:", worst)
