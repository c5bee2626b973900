#!/usr/bin/env python3
"""tools/bench_slab_shapes.py — BFP along a non-contiguous dim on conv / attention shapes with ROTATING buffers (> 600 MB per shape, so that
nothing is served by the Infinity Cache; tools/bench_conv_shapes.py re-uses one tensor): the A/B harness of csrc/bfp_slab.hip.
    DMXQ_SLAB=0 python tools/bench_slab_shapes.py     the column kernel everywhere
    DMXQ_SLAB=1 python tools/bench_slab_shapes.py     the LDS slab kernel wherever it applies
    python tools/bench_slab_shapes.py                  the library's own routing
Output: profiles/r06_slab_ab3 / ab4 / ab5."""
import sys, torch
sys.path.insert(0, ".")
import dmx_compressor_amd as dmx
dev = torch.device("cuda:0")
def bench(name, f, byt, iters=100):
    for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / iters
    print(f"{name:48s} {us:8.2f} us {byt / us / 1e3 / 80:5.1f}%", flush=True)
for shape, dim, B in (((64, 256, 56, 56), 1, 64), ((64, 512, 28, 28), 1, 64), ((64, 1024, 14, 14), 1, 64), ((256, 1024, 14, 14), 1, 64), ((64, 3, 224, 224), 1, 64),
                      ((8, 12, 1500, 64), -2, 64), ((8, 32, 2048, 128), -2, 64), ((64, 64, 112, 112), 1, 64),
                      ((16, 16, 1024, 256), -2, 64), ((8, 64, 1024, 192), -2, 64), ((8, 32, 2048, 128), -2, 128), ((32, 32, 256, 128), -2, 32)):
    xs = [torch.randn(*shape, device=dev).to(torch.bfloat16) for _ in range(max(2, int(6e8 // (torch.tensor(shape).prod().item() * 2))))]
    i = [0]
    def f():
        i[0] = (i[0] + 1) % len(xs)
        dmx.ops.bfp_qdq(xs[i[0]], 8, B, dim)
    bench(f"{shape} dim={dim} B={B} ({len(xs)} bufs)", f, xs[0].numel() * 4)
