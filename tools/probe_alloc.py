#!/usr/bin/env python3
"""tools/probe_alloc.py -- does the PLACEMENT of the rotating buffers explain the 3072 x 4096 discrepancy (10.34 us
through tools/bench_shapes.py, 9.35 us in tools/tune_bfp for the same 512 x 16 kernel)?  The same dmxq_bfp_qdq launches over buffer
sets that differ only in where they live: separate torch allocations (bench_shapes.py), one block carved at the tensor's own size,
carved with a pad that breaks the power-of-two-ish stride, and raw hipMalloc (tune_bfp)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dmx_compressor_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    vp = ctypes.c_void_p
    stream = torch.cuda.Stream()
    sp = vp(stream.cuda_stream)
    C = 4096

    def time_set(ins, outs, R, iters=300):
        k = len(ins)
        with torch.cuda.stream(stream):
            def launch(i):
                assert L.dmxq_bfp_qdq(vp(ins[i]), vp(outs[i]), _lib.BF16, _lib.BF16, R, C, 1, 16, 8, 2, 1, 0, sp) == 0
            best = 1e9
            for rep in range(3):
                for i in range(100):
                    launch(i % k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for i in range(iters):
                    launch(i % k)
                e1.record(stream)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        return best

    print(f"{'rows':>6s} {'MiB':>6s} {'nbuf':>4s} | {'separate torch':>14s} {'carved tight':>13s} {'carved +1 MiB':>13s} {'carved +64 KiB':>14s} {'hipMalloc':>10s}   (us per launch, best of 3)")
    for R in (1024, 2048, 2560, 3072, 3584, 4096, 4608, 6144):
        torch.cuda.empty_cache()
        n = R * C
        nbytes = n * 2
        k = max(4, min(16, -(-700 * 2 ** 20 // (nbytes * 2))))
        src = (torch.randn(R, C, device=dev) * torch.exp(2 * torch.randn(R, C, device=dev))).to(torch.bfloat16)
        res = []
        # (1) separate torch tensors
        xs = [src.clone() for _ in range(k)]
        ys = [torch.empty_like(src) for _ in range(k)]
        res.append(time_set([t.data_ptr() for t in xs], [t.data_ptr() for t in ys], R))
        del xs, ys
        torch.cuda.empty_cache()
        # (2-4) one block, carved
        for pad in (0, 1 << 20, 1 << 16):
            stride = nbytes + pad
            blk = torch.empty(2 * k * stride + 4096, dtype=torch.uint8, device=dev)
            base = (blk.data_ptr() + 4095) & ~4095
            ins = [base + (2 * i) * stride for i in range(k)]
            outs = [base + (2 * i + 1) * stride for i in range(k)]
            for p in ins:
                hip.hipMemcpyAsync(vp(p), vp(src.data_ptr()), ctypes.c_size_t(nbytes), 3, sp)
            torch.cuda.synchronize()
            res.append(time_set(ins, outs, R))
            del blk
            torch.cuda.empty_cache()
        # (5) raw hipMalloc
        ptrs = []
        for _ in range(2 * k):
            p = vp()
            assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
            ptrs.append(p.value)
        for p in ptrs[:k]:
            hip.hipMemcpy(vp(p), vp(src.data_ptr()), ctypes.c_size_t(nbytes), 3)
        res.append(time_set(ptrs[:k], ptrs[k:], R))
        for p in ptrs:
            hip.hipFree(vp(p))
        print(f"{R:>6d} {nbytes / 2**20:6.1f} {k:>4d} | {res[0]:14.2f} {res[1]:13.2f} {res[2]:13.2f} {res[3]:14.2f} {res[4]:10.2f}", flush=True)


if __name__ == "__main__":
    main()
