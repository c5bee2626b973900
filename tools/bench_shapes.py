#!/usr/bin/env python3
"""tools/bench_shapes.py — BFP16 Q->DQ (bf16 -> bf16) over the secondary shapes of SURVEY.md §8(d): Llama-3-8B and
opt-125m weight shapes, activation shapes, B in {16, 64}.  Eager C-ABI launches (tiny tensors are therefore host-launch-bound: ~2 us per call) that rotate over enough buffers to exceed the Infinity Cache where the tensor allows it."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dmx_compressor_amd import _lib  # noqa: E402

SHAPES = [(2048, 4096), (3072, 4096), (4096, 4096), (4100, 4096), (4608, 4096), (5000, 4096), (6144, 4096), (8192, 4096), (10240, 4096), (14336, 4096), (4096, 14336), (1024, 4096), (128256, 4096), (16384, 4096), (2048, 14336), (768, 768),
          (3072, 768), (768, 3072), (50272, 768), (12 * 1500, 1500), (1500, 768), (64, 4096), (120, 400)]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--mid":   # the 16 - 64 MiB band only
        global SHAPES
        SHAPES = [(r, 4096) for r in (2048, 2560, 3072, 3584, 4096, 4100, 4352, 4608, 5000, 5632, 6144, 6200, 8192)]
    dev = torch.device("cuda:0")
    L = _lib.lib()
    vp = ctypes.c_void_p
    stream = torch.cuda.Stream()
    sp = vp(stream.cuda_stream)
    print(f"{'shape':>16s} {'B':>4s} {'us':>9s} {'GB/s':>9s} {'%8TB/s':>7s}  nbuf")
    for R, C in SHAPES:
        torch.cuda.empty_cache()
        n = R * C
        nbuf = max(2, min(16, math.ceil(600 * 2 ** 20 / (n * 4))))
        # buffers FIRST, in a clean cache (every one its own allocation), then the heavy-tailed values slab by slab: with whole-tensor
        # float32 temporaries in between, the bf16 tensors were carved back to back out of recycled blocks, and the one-round plans
        # are sensitive to that placement (tools/probe_deep.py: 4100 x 4096 11.1 us or 12.6 us with the same kernel and data)
        xs = [torch.empty(R, C, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
        ys = [torch.empty_like(x) for x in xs]
        for x in xs:
            for r0 in range(0, R, 1024):
                r1 = min(R, r0 + 1024)
                x[r0:r1] = (torch.randn(r1 - r0, C, device=dev) * torch.exp(2 * torch.randn(r1 - r0, C, device=dev))).to(torch.bfloat16)
        for B in (16, 64):
            iters = 200
            with torch.cuda.stream(stream):
                def launch(i):
                    rc = L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, B, 8, 2, 1, 0, sp)
                    assert rc == 0
                us = float("inf")
                for rep in range(3):   # best of three (round 3 printed a single pass: 3072 x 4096 read 10.3 us there, 9.3-9.4 in every later run)
                    for i in range(100):   # long warm-up: clocks and caches settle (10 launches read 10-30 % slow)
                        launch(i % nbuf)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for i in range(iters):   # eager C-ABI launches (a hipGraph replay of big kernels measured ~30 % slower here)
                        launch(i % nbuf)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    us = min(us, e0.elapsed_time(e1) * 1e3 / iters)
            gbs = n * 4 / (us * 1e-6) / 1e9
            print(f"{R:>9d}x{C:<6d} {B:>4d} {us:9.2f} {gbs:9.1f} {100 * gbs / 8000:6.1f}%  {nbuf}", flush=True)
        del xs, ys
        torch.cuda.empty_cache()  # fresh allocations per shape: blocks carved out of a fragmented cache measured up to 50 % slower

    if len(sys.argv) > 1 and sys.argv[1] == "--mid":
        return
    # many small weights: one launch per tensor vs the multi-tensor entry point (opt-125m's 73 Linear weights, BFP16_64).
    # Straight C-ABI calls with prebuilt arguments / descriptors (the Python front end adds ~3 us per tensor for allocation
    # and descriptor filling, which is a one-off at fold time and would hide the kernels here).
    shapes = []
    for _ in range(12):
        shapes += [(768, 768)] * 4 + [(3072, 768), (768, 3072)]
    shapes += [(50272, 768)]
    sets = [[(torch.randn(s, device=dev) * 0.05).to(torch.bfloat16) for s in shapes] for _ in range(3)]  # 3 x 248 MB
    outs = [[torch.empty_like(w) for w in ws] for ws in sets]
    n = sum(a * b for a, b in shapes)
    single_args = [[(vp(w.data_ptr()), vp(o.data_ptr()), _lib.BF16, _lib.BF16, w.shape[0], w.shape[1], 1, 64, 8, 2, 1, 0, sp) for w, o in zip(ws, os_)]
                   for ws, os_ in zip(sets, outs)]
    descs = []
    for ws, os_ in zip(sets, outs):
        d = (_lib.TensorDesc * len(ws))()
        for e, w, o in zip(d, ws, os_):
            e.in_, e.out, e.outer, e.L, e.inner = w.data_ptr(), o.data_ptr(), w.shape[0], w.shape[1], 1
        descs.append(d)

    def one_by_one(i):
        for a in single_args[i]:
            L.dmxq_bfp_qdq(*a)

    def multi(i):
        assert L.dmxq_bfp_qdq_multi(descs[i], len(shapes), _lib.BF16, _lib.BF16, 64, 8, 2, 1, 0, sp) == 0

    with torch.cuda.stream(stream):
        for name, fn in (("one dmxq_bfp_qdq launch per tensor (73 launches)", one_by_one), ("dmxq_bfp_qdq_multi (2 launches)", multi)):
            for i in range(6):
                fn(i % 3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(30):
                fn(i % 3)
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 30
            print(f"opt-125m 73 Linear weights ({n / 1e6:.1f} M elements) BFP16_64, {name}: {us:9.1f} us  "
                  f"{n * 4 / (us * 1e-6) / 1e9:8.1f} GB/s  {100 * n * 4 / (us * 1e-6) / 8e12:5.1f}%", flush=True)
    # the INT8 group-quant twin (BASELINE.json configs[2]: group_size = 128 rows): one dmxq_fixed_qdq per tensor vs dmxq_fixed_qdq_multi
    scales = [[(torch.rand(-(-s[0] // 128), device=dev) * 0.002 + 0.0005) for s in shapes] for _ in range(3)]
    zps = [[torch.zeros(-(-s[0] // 128), dtype=torch.int64, device=dev) for s in shapes] for _ in range(3)]
    f_args = [[(vp(w.data_ptr()), vp(o.data_ptr()), _lib.BF16, _lib.BF16, 1, w.shape[0], w.shape[1], 8, 0, 1, 1, 2, vp(sc.data_ptr()), vp(z.data_ptr()), 128, 0, sp)
               for w, o, sc, z in zip(ws, os_, scs, zs)] for ws, os_, scs, zs in zip(sets, outs, scales, zps)]
    adescs = []
    for ws, os_, scs, zs in zip(sets, outs, scales, zps):
        d = (_lib.AffineDesc * len(ws))()
        for e, w, o, sc, z in zip(d, ws, os_, scs, zs):
            e.in_, e.out, e.scale, e.zero_point, e.outer, e.C, e.inner = w.data_ptr(), o.data_ptr(), sc.data_ptr(), z.data_ptr(), 1, w.shape[0], w.shape[1]
        adescs.append(d)

    def f_one_by_one(i):
        for a in f_args[i]:
            L.dmxq_fixed_qdq(*a)

    def f_multi(i):
        assert L.dmxq_fixed_qdq_multi(adescs[i], len(shapes), _lib.BF16, _lib.BF16, 8, 0, 1, 1, 2, 128, 0, sp) == 0

    with torch.cuda.stream(stream):
        for name, fn in (("one dmxq_fixed_qdq launch per tensor (73 launches)", f_one_by_one), ("dmxq_fixed_qdq_multi (4 launches)", f_multi)):
            for i in range(6):
                fn(i % 3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(30):
                fn(i % 3)
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 30
            print(f"opt-125m 73 Linear weights ({n / 1e6:.1f} M elements) INT8 group 128, {name}: {us:9.1f} us  "
                  f"{n * 4 / (us * 1e-6) / 1e9:8.1f} GB/s  {100 * n * 4 / (us * 1e-6) / 8e12:5.1f}%", flush=True)


if __name__ == "__main__":
    main()
