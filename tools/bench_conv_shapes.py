#!/usr/bin/env python3
"""tools/bench_conv_shapes.py — BFP along a NON-contiguous dim on the shapes convolutional / attention models produce
(blocks along channels of [N, C, H, W], along the sequence of [B, H, S, D]); bf16, one tensor re-used (small tensors are
cache resident: read the large ones for bandwidth, the small ones for launch cost).  Output: profiles/r01_conv_shapes.txt"""
import sys

import torch

sys.path.insert(0, ".")
import dmx_compressor_amd as dmx  # noqa: E402

dev = torch.device("cuda:0")


def bench(name, f, byt, iters=200):
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / iters
    print(f"{name:58s} {us:8.2f} us {byt / us / 1e3:8.1f} GB/s {byt / us / 1e3 / 80:5.1f}%", flush=True)


for shape, dim, B in (((64, 256, 56, 56), 1, 64), ((64, 512, 28, 28), 1, 64), ((64, 1024, 14, 14), 1, 64), ((256, 1024, 14, 14), 1, 64),
                      ((64, 2048, 7, 7), 1, 64), ((64, 3, 224, 224), 1, 64), ((512, 512, 3, 3), 1, 64), ((1280, 1280, 3), 1, 64),
                      ((8, 12, 1500, 64), -2, 64), ((8, 12, 64, 1500), -1, 64), ((8, 32, 2048, 128), -2, 64)):
    x = torch.randn(*shape, device=dev).to(torch.bfloat16)
    bench(f"bfp bf16 {shape} block_dim={dim} B={B}", lambda: dmx.ops.bfp_qdq(x, 8, B, dim), x.numel() * 4)
