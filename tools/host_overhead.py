#!/usr/bin/env python3
"""tools/host_overhead.py — host time per call of the path from `CastTo.forward` down to the C ABI (GPU box), the table of
profiles/r0N_host_overhead.txt:  median host time of 1000 calls in bursts of 100 on an idle stream (the GPU keeps up: what is timed is
the host's cost of ISSUING a call, not the kernel).
    python tools/host_overhead.py
"""
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dmx_compressor_amd as d  # noqa: E402
from dmx_compressor_amd import _lib  # noqa: E402


def host_us(fn, bursts=10, per=100):
    ts = []
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for _ in range(bursts):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            fn()
        ts.append((time.perf_counter() - t0) / per * 1e6)
        torch.cuda.synchronize()
    return statistics.median(ts)


def main():
    dev = torch.device("cuda:0")
    x = torch.randn(256, 768, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    L = _lib.lib()
    vp = ctypes.c_void_p
    sp = vp(torch.cuda.current_stream().cuda_stream)
    args = (vp(x.data_ptr()), vp(out.data_ptr()), _lib.BF16, _lib.BF16, 256, 768, 1, 64, 8, 2, 1, 0, sp)
    rows = []
    rows.append(("C ABI dmxq_bfp_qdq alone (prebuilt ctypes arguments, no allocation)", host_us(lambda: L.dmxq_bfp_qdq(*args))))
    rows.append(("torch.empty_like (the output allocation alone)", host_us(lambda: torch.empty_like(x))))
    if os.environ.get("DMXQ_BINDING", "torch") != "ctypes":
        rows.append(("torch.ops.dmxq.bfp_qdq (STE autograd wrapper)", host_us(lambda: torch.ops.dmxq.bfp_qdq(x, 8, 64, -1, True, 2, None, 0))))
        rows.append(("torch.ops.dmxq.bfp_qdq_nograd", host_us(lambda: torch.ops.dmxq.bfp_qdq_nograd(x, 8, 64, -1, True, 2, None, 0))))
    rows.append(("ops.bfp_qdq", host_us(lambda: d.ops.bfp_qdq(x, 8, 64))))
    cast = d.CastTo(format="BFP[8|8]{64}(SN)").to(dev)
    with torch.no_grad():
        rows.append(("CastTo.forward(x) (BFP16_64, no_grad)", host_us(lambda: cast.forward(x))))
        rows.append(("CastTo(x) through nn.Module.__call__ (no_grad)", host_us(lambda: cast(x))))
        c16 = d.CastTo(format="FP[1|5|10,15](FN)").to(dev)
        rows.append(("CastTo.forward(x) (FLOAT16 activation cast, no_grad)", host_us(lambda: c16.forward(x))))
        lin = d.nn.Linear(768, 768).to(dev).to(torch.bfloat16).eval()
        for r in d.config_rules.BASIC:
            if isinstance(lin, r.module_types):
                lin.configure(r.module_config)
        rows.append(("dmx nn.Linear(768, 768) BASIC forward, 256 tokens (4 casts + weight chain + F.linear)", host_us(lambda: lin(x), per=50)))
        raw = torch.nn.Linear(768, 768).to(dev).to(torch.bfloat16)
        rows.append(("torch.nn.Linear(768, 768) forward, 256 tokens (the GEMM call alone)", host_us(lambda: raw(x), per=50)))
    for name, us in rows:
        print(f"{name}: {us:.2f} us per call")


if __name__ == "__main__":
    main()
