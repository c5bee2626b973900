#!/usr/bin/env python3
"""tools/probe_partial.py -- does the LAST, partial tile of a flat tensor cost a step?  Every flat-stream op at row counts around a
size class boundary whose last tile is 0 / 25 / 50 / 75 / 94 % full (4096-column bf16), events over 200 launches on rotating buffers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dmx_compressor_amd as dmx  # noqa: E402

ops = dmx.ops
dev = torch.device("cuda:0")
OPS = {   # (outputs come from torch's caching allocator: ~nbuf blocks in rotation)
    "bfp16": lambda x: ops.bfp_qdq(x, 8, 16),
    "float E4M3": lambda x: ops.float_qdq(x, 3, 4, 7, False),
    "fixed INT8": lambda x: ops.fixed_qdq(x, 8, 0),
    "silu": lambda x: ops.silu(x),
}


def run(name, fn, R, C=4096, nbuf=10):
    xs = [torch.randn(R, C, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
    best = 1e9
    for rep in range(3):
        for i in range(50):
            fn(xs[i % nbuf])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            fn(xs[i % nbuf])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    print(f"{name:12s} {R:5d}x{C}: {best:7.2f} us  {R * C * 4 / best / 1e3 / 8000 * 100:5.1f}%", flush=True)


rows = [int(a) for a in sys.argv[1:]] or [3840, 3900, 3950, 4000, 4050, 4090, 4096]
for name, fn in OPS.items():
    for R in rows:
        run(name, fn, R)
