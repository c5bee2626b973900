#!/usr/bin/env python3
"""tools/bench_rows.py — Softmax / LayerNorm (and BFP on the same tensor, as the copy-like yardstick) over row lengths
256 .. 8192, bf16, direct C-ABI launches with rotating buffers > 256 MiB (GPU box).  Output: profiles/r01_row_ops.txt"""
import sys, ctypes, math, torch
sys.path.insert(0, '.')
from dmx_compressor_amd import _lib
L = _lib.lib(); vp = ctypes.c_void_p
dev = torch.device('cuda:0')
s = vp(torch.cuda.current_stream().cuda_stream)
def bench(name, f, nb, byt, iters=200):
    for i in range(30): f(i % nb)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): f(i % nb)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / iters
    print(f"{name:44s} {us:8.2f} us {byt/us/1e3:8.1f} GB/s {byt/us/1e3/80:5.1f}%", flush=True)
for rows, cols in ((18000, 1500), (18000, 1536), (16384, 2048), (8192, 4096), (24000, 768), (32768, 1024), (4096, 8192), (65536, 256)):
    nb = max(2, min(24, math.ceil(600e6 / (rows * cols * 4))))
    x = [torch.randn(rows, cols, device=dev).to(torch.bfloat16) for _ in range(nb)]
    y = [torch.empty_like(t) for t in x]
    w = torch.ones(cols, device=dev, dtype=torch.bfloat16)
    bench(f"softmax bf16 {rows}x{cols}", lambda i: L.dmxq_softmax(vp(x[i].data_ptr()), vp(y[i].data_ptr()), _lib.BF16, _lib.BF16, rows, cols, ctypes.c_float(-math.inf), s), nb, rows * cols * 4)
    bench(f"layernorm bf16 {rows}x{cols}", lambda i: L.dmxq_layernorm(vp(x[i].data_ptr()), vp(y[i].data_ptr()), _lib.BF16, _lib.BF16, rows, cols, vp(w.data_ptr()), vp(w.data_ptr()), _lib.BF16, ctypes.c_float(1e-5), s), nb, rows * cols * 4)
    bench(f"  (bfp_qdq same tensor)", lambda i: L.dmxq_bfp_qdq(vp(x[i].data_ptr()), vp(y[i].data_ptr()), _lib.BF16, _lib.BF16, rows, cols, 1, 16, 8, 2, 1, 0, s), nb, rows * cols * 4)
    del x, y
