#!/usr/bin/env python3
"""tools/bench_shard_sets.py -- BASELINE.json configs[3] under row sharding, every rank's work measured on ONE GPU.

A Llama-3-8B decoder layer has seven Linear weights (218.1 M elements); sharded by rows over N ranks, rank r's step
is the weight chain (BTOPK{2:4,-1} mask -> BFP[8|8]{64}(SN), bf16 weight / score / result: 6 B per element) over ITS seven shards.
For N = 1, 2, 4, 8 this script builds rank 0's shard set (all ranks' sets have the same sizes: every row count divides by 8) and
times one step two ways -- one dmxq_weight_hypernet launch per shard (rounds 2-3) and the whole set in ONE
dmxq_weight_hypernet_multi launch -- with HIP events over eager C-ABI launches that rotate over enough copies to exceed the 256 MiB
Infinity Cache.  Strong scaling is linear iff  (per-rank us at N) x N == (us at N = 1): the last column.  This is the per-rank
kernel-side ceiling of `bench.py --workload llama-shard`; it says nothing about RCCL or the driver's 8-GPU node (never available
to this builder), only that the library adds no launch floor of its own as the shards shrink."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dmx_compressor_amd import _lib  # noqa: E402
from dmx_compressor_amd.parallel import row_shards  # noqa: E402

LLAMA = [("q", 4096, 4096), ("k", 1024, 4096), ("v", 1024, 4096), ("o", 4096, 4096), ("gate", 14336, 4096), ("up", 14336, 4096), ("down", 4096, 14336)]
PEAK = 8.0e12


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    vp = ctypes.c_void_p
    stream = torch.cuda.Stream()
    sp = vp(stream.cuda_stream)
    print(f"{'N':>2s} {'elements/rank':>14s} {'copies':>6s} | {'7 launches us':>13s} {'%8TB/s':>7s} {'x N':>8s} | {'1 launch us':>11s} {'%8TB/s':>7s} {'x N':>8s} {'vs N=1':>7s}")
    base = {}
    for N in (1, 2, 4, 8):
        torch.cuda.empty_cache()
        shapes = [(row_shards(r, N)[0][1] - row_shards(r, N)[0][0], c) for _, r, c in LLAMA]
        n = sum(a * b for a, b in shapes)
        copies = max(2, math.ceil(700 * 2 ** 20 / (n * 6)))
        sets = []
        for _ in range(copies):
            ws = [(torch.randn(s, device=dev) * 0.02).to(torch.bfloat16) for s in shapes]
            ss = [torch.rand(s, device=dev).to(torch.bfloat16) for s in shapes]
            os_ = [torch.empty_like(w) for w in ws]
            d = (_lib.HypernetDesc * len(ws))()
            for e, w, s, o in zip(d, ws, ss, os_):
                e.w, e.score, e.sq_scale, e.out, e.rows, e.L = w.data_ptr(), s.data_ptr(), None, o.data_ptr(), w.shape[0], w.shape[1]
            each = [(vp(w.data_ptr()), _lib.BF16, vp(s.data_ptr()), _lib.BF16, 2, 4, None, vp(o.data_ptr()), _lib.BF16, w.shape[0], w.shape[1], 64, 8, 1, sp)
                    for w, s, o in zip(ws, ss, os_)]
            sets.append((ws, ss, os_, d, each))

        def run_each(i):
            for a in sets[i][4]:
                assert L.dmxq_weight_hypernet(*a) == 0

        def run_multi(i):
            assert L.dmxq_weight_hypernet_multi(sets[i][3], len(shapes), _lib.BF16, _lib.BF16, 2, 4, _lib.BF16, 64, 8, 1, sp) == 0

        res = {}
        with torch.cuda.stream(stream):
            # the two forms must agree bit for bit before anything is timed
            run_each(0)
            torch.cuda.synchronize()
            ref = [o.clone() for o in sets[0][2]]
            for o in sets[0][2]:
                o.zero_()
            run_multi(0)
            torch.cuda.synchronize()
            assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ref, sets[0][2])), "multi != per-tensor launches"
            del ref
            iters = max(40, min(400, int(20000 / (n * 6 / 6e12 * 1e6))))
            for name, fn in (("each", run_each), ("multi", run_multi)):
                best = float("inf")
                for rep in range(3):
                    for i in range(max(20, iters // 4)):
                        fn(i % copies)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for i in range(iters):
                        fn(i % copies)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
                res[name] = best
        if N == 1:
            base = dict(res)
        pct = lambda us: 100 * n * 6 / (us * 1e-6) / PEAK
        print(f"{N:>2d} {n:>14d} {copies:>6d} | {res['each']:13.2f} {pct(res['each']):6.1f}% {res['each'] * N:8.1f} | "
              f"{res['multi']:11.2f} {pct(res['multi']):6.1f}% {res['multi'] * N:8.1f} {res['multi'] * N / base['multi']:7.3f}", flush=True)
        del sets


if __name__ == "__main__":
    main()
