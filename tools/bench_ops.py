#!/usr/bin/env python3
"""tools/bench_ops.py — per-op HBM roofline table for every libdmxq entry point (GPU box).

For each op: direct C-ABI launches (ctypes, preallocated outputs) on one stream, rotating over enough buffer
sets to exceed the 256 MiB Infinity Cache, HIP-event timing; reports us/launch, algorithmic GB/s, % of 8 TB/s.
    python tools/bench_ops.py [--rows 4096 --cols 4096] [--json out.json]
"""
import argparse
import ctypes
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dmx_compressor_amd as d  # noqa: E402
from dmx_compressor_amd import _lib  # noqa: E402

PEAK = 8.0e12
DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
vp = ctypes.c_void_p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=200, help="launches per timed group; the median of 5 groups is reported")
    ap.add_argument("--warm-ms", type=float, default=40.0, help="untimed launches of the op until this much GPU time has passed")
    ap.add_argument("--json", default=None)
    ap.add_argument("--data", default="heavy", choices=("heavy", "randn"),
                    help="input distribution of the [rows, cols] operands: randn * exp(2 randn) (default: outliers in every block, ~40 "
                         "binades) or plain randn; the chip streams plain randn 10-15 %% faster through the SAME instruction stream "
                         "(profiles/r05_warmup_effect.txt)")
    ap.add_argument("--only", default=None, help="comma-separated substrings: run only the ops whose name contains one of them")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    R, C = args.rows, args.cols
    n = R * C
    stream = torch.cuda.Stream()
    sp = vp(stream.cuda_stream)
    results = []

    def bufs(dtype, count, shape=(R, C), gen=None):
        out = []
        for i in range(count):
            g = torch.Generator(device=dev).manual_seed(i)
            t = torch.randn(*shape, generator=g, device=dev)
            if args.data == "heavy":
                t = t * torch.exp(2 * torch.randn(*shape, generator=g, device=dev))
            out.append(t.to(dtype))
        return out

    only = [t.strip() for t in args.only.split(",")] if args.only else None

    def run(name, launch, nbuf, bytes_per_launch, check=None):
        if only is not None and not any(t in name for t in only):
            return
        # warm-up by GPU TIME, not by count: the first few hundred launches after the input generation (Philox kernels, allocations)
        # run 5-15 % slow whatever the op -- same instruction counts (SQ_INSTS_VALU / SALU identical), the chip's clocks settling --
        # and 20 launches hid that in the first op measured on every new buffer set (profiles/r05_warmup_effect.txt)
        with torch.cuda.stream(stream):
            warmed = 0.0
            while warmed < args.warm_ms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for i in range(100):
                    launch(i % nbuf)
                e1.record(stream)
                torch.cuda.synchronize()
                warmed += e0.elapsed_time(e1)
            groups = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for i in range(args.iters):
                    launch(i % nbuf)
                e1.record(stream)
                torch.cuda.synchronize()
                groups.append(e0.elapsed_time(e1) * 1e3 / args.iters)
        us = sorted(groups)[2]
        gbs = bytes_per_launch / (us * 1e-6) / 1e9
        results.append({"op": name, "us": round(us, 2), "GB/s": round(gbs, 1), "frac": round(gbs * 1e9 / PEAK, 4),
                        "bytes": bytes_per_launch, "us_groups": [round(g, 2) for g in groups]})
        print(f"{name:58s} {us:9.2f} us {gbs:9.1f} GB/s {100 * gbs * 1e9 / PEAK:6.1f}%", flush=True)

    def nb(per_set_bytes):  # buffer sets needed to exceed 512 MiB total
        return max(2, min(24, math.ceil(512 * 2 ** 20 / per_set_bytes)))

    # ---------------------------------------------------------------- BFP, rows
    for din, dout in ((torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16), (torch.float32, torch.float32),
                      (torch.bfloat16, torch.float32), (torch.float32, torch.bfloat16)):
        byt = n * (din.itemsize + dout.itemsize)
        k = nb(byt)
        xs = bufs(din, k)
        ys = [torch.empty(R, C, dtype=dout, device=dev) for _ in range(k)]
        for B, wl, sym, rnd in ((16, 8, 1, 2), (64, 8, 1, 2), (128, 8, 1, 2), (64, 8, 0, 2), (16, 16, 1, 2), (16, 8, 1, 3), (64, 8, 1, 1)):
            if din != torch.bfloat16 and (B, wl, sym, rnd) not in ((16, 8, 1, 2), (64, 8, 1, 2)):
                continue
            run(f"bfp_qdq {str(din)[6:]}->{str(dout)[6:]} B={B} wl={wl} {'sym' if sym else 'asym'} rnd={rnd}",
                lambda i: L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), DT[din], DT[dout], R, C, 1, B, wl, rnd, sym, 7, sp),
                k, byt)
        del xs, ys
    # ---------------------------------------------------------------- BFP, column blocks and ragged
    k = nb(n * 4)
    xs = bufs(torch.bfloat16, k)
    ys = [torch.empty_like(x) for x in xs]
    run("bfp_qdq bf16 block_dim=-2 B=64 ([R,C] blocks along R)",
        lambda i: L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, 1, R, C, 64, 8, 2, 1, 0, sp), k, n * 4)
    run("bfp_qdq bf16 block_dim=1 of [R/64,64,C] B=16",
        lambda i: L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R // 64, 64, C, 16, 8, 2, 1, 0, sp), k, n * 4)
    run("bfp_qdq bf16 conv weight [512,512,3,3] along in-channels B=64 (4.7 MB: launch-bound; LDS sub-slab kernel)",
        lambda i: L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, 512, 512, 9, 64, 8, 2, 1, 0, sp), k, 512 * 512 * 9 * 4)
    run("bfp_qdq bf16 feature map [64,2048,7,7] along channels B=64 (LDS sub-slab kernel)",
        lambda i: L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, 64, 2048, 49, 64, 8, 2, 1, 0, sp), k, 64 * 2048 * 49 * 4)
    # conv activations whose rows are not whole 128-byte lines (round 6: csrc/bfp_slab.hip), each on its own rotating buffers
    for shp in ((256, 1024, 14, 14), (64, 512, 28, 28)):
        nel = shp[0] * shp[1] * shp[2] * shp[3]
        kk = nb(nel * 4)
        cx = bufs(torch.bfloat16, kk, shape=shp)
        cy = [torch.empty_like(t) for t in cx]
        run(f"bfp_qdq bf16 feature map [{shp[0]},{shp[1]},{shp[2]},{shp[3]}] along channels B=64 (LDS slab kernel: rows of {shp[2] * shp[3] * 2} bytes)",
            lambda i: L.dmxq_bfp_qdq(vp(cx[i].data_ptr()), vp(cy[i].data_ptr()), _lib.BF16, _lib.BF16, shp[0], shp[1], shp[2] * shp[3], 64, 8, 2, 1, 0, sp), kk, nel * 4)
        del cx, cy
    run("bfp_qdq bf16 rows ragged L=C-8 B=64 (generic path)",
        lambda i: L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C - 8, 1, 64, 8, 2, 1, 0, sp), k, R * (C - 8) * 4)
    # ---------------------------------------------------------------- float / fixed / scale / gelu
    run("float_qdq bf16 FP16(FN) [BASIC activation cast]",
        lambda i: L.dmxq_float_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, n, 10, 5, 15, 1, 0, 2, 0, sp), k, n * 4)
    run("float_qdq bf16 E4M3 (AFLOAT8)",
        lambda i: L.dmxq_float_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, n, 3, 4, 7, 0, 0, 2, 0, sp), k, n * 4)
    run("fixed_qdq bf16 INT8 no affine",
        lambda i: L.dmxq_fixed_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, 1, 1, n, 8, 0, 1, 1, 2, None, None, 1, 0, sp), k, n * 4)
    G = R // 128
    sc = (torch.rand(G, device=dev) * 0.05 + 0.01)
    zp = torch.zeros(G, dtype=torch.int64, device=dev)
    run("fixed_qdq bf16 INT8 group_size=128 along dim0 (opt-125m style)",
        lambda i: L.dmxq_fixed_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, 1, R, C, 8, 0, 1, 1, 2, vp(sc.data_ptr()), vp(zp.data_ptr()), 128, 0, sp), k, n * 4)
    scc = (torch.rand(C, device=dev) + 0.5)
    zpc = torch.zeros(C, dtype=torch.int64, device=dev)
    run("fixed_qdq bf16 INT8 per-channel along last dim",
        lambda i: L.dmxq_fixed_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, 8, 0, 1, 1, 2, vp(scc.data_ptr()), vp(zpc.data_ptr()), 1, 0, sp), k, n * 4)
    run("scale_channels bf16 divide along last dim (SmoothQuant input)",
        lambda i: L.dmxq_scale_channels(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, vp(scc.data_ptr()), 1, sp), k, n * 4)
    run("gelu bf16 (erf)", lambda i: L.dmxq_gelu(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, n, 0, sp), k, n * 4)
    # ---------------------------------------------------------------- N:M
    ss = bufs(torch.float32, max(2, k // 2))
    k2 = len(ss)
    yf = [torch.empty(R, C, dtype=torch.float32, device=dev) for _ in range(k2)]
    for K_, M_ in ((2, 4), (4, 8)):
        run(f"nm_mask {K_}:{M_} fp32 score -> fp32 mask",
            lambda i: L.dmxq_nm_mask(vp(ss[i].data_ptr()), _lib.F32, None, 0, vp(yf[i].data_ptr()), _lib.F32, None, 0, R, C, 1, K_, M_, sp), k2, n * 8)
        run(f"nm_sparsify {K_}:{M_} fp32 score, bf16 x -> bf16 y (fused apply)",
            lambda i: L.dmxq_nm_mask(vp(ss[i].data_ptr()), _lib.F32, vp(xs[i].data_ptr()), _lib.BF16, None, 0, vp(ys[i].data_ptr()), _lib.BF16, R, C, 1, K_, M_, sp), k2, n * 8)
    # a |w| score is its own tensor (Sparsify's score_func materialises it, sparse.py:287-294): score + x + y = 6 B/element.  (Rounds 1-2
    # passed the SAME buffer as score and x and counted 4 B/element: the second read was an L2 hit and the row read 56 %.)
    run("nm_sparsify 2:4 bf16 score(|w|), bf16 x -> bf16 y (6 B/elem)",
        lambda i: L.dmxq_nm_mask(vp(xs[(i + 1) % k].data_ptr()), _lib.BF16, vp(xs[i].data_ptr()), _lib.BF16, None, 0, vp(ys[i].data_ptr()), _lib.BF16, R, C, 1, 2, 4, sp), k, n * 6)
    # ---------------------------------------------------------------- fused weight hypernet (mask -> scale -> BFP)
    sq = (torch.rand(C, device=dev) + 0.5)
    run("weight_hypernet 2:4 mask + BFP16_64, bf16 w, fp32 score -> bf16 (vs 3 unfused passes)",
        lambda i: L.dmxq_weight_hypernet(vp(xs[i].data_ptr()), _lib.BF16, vp(ss[i % k2].data_ptr()), _lib.F32, 2, 4, None, vp(ys[i].data_ptr()), _lib.BF16, R, C, 64, 8, 1, sp), k2, n * 8)
    run("weight_hypernet 2:4 mask + SmoothQuant scale + BFP16_64 -> bf16",
        lambda i: L.dmxq_weight_hypernet(vp(xs[i].data_ptr()), _lib.BF16, vp(ss[i % k2].data_ptr()), _lib.F32, 2, 4, vp(sq.data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, R, C, 64, 8, 1, sp), k2, n * 8)
    run("weight_hypernet dense + SmoothQuant scale + BFP16_64, bf16 -> bf16",
        lambda i: L.dmxq_weight_hypernet(vp(xs[i].data_ptr()), _lib.BF16, None, 0, 0, 0, vp(sq.data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, R, C, 64, 8, 1, sp), k, n * 4)
    run("input_hypernet SmoothQuant x / s + BFP16_64, bf16 -> fp32 (vs scale_channels + bfp_qdq: 14 B/elem)",
        lambda i: L.dmxq_input_hypernet(vp(xs[i].data_ptr()), _lib.BF16, vp(sq.data_ptr()), vp(yf[i % k2].data_ptr()), _lib.F32, R, C, 64, 8, 1, sp), k2, n * 6)
    run("  the two launches it replaces: scale_channels bf16 -> fp32, then bfp_qdq fp32 -> fp32",
        lambda i: (L.dmxq_scale_channels(vp(xs[i].data_ptr()), vp(yf[i % k2].data_ptr()), _lib.BF16, _lib.F32, R, C, 1, vp(sq.data_ptr()), 1, sp),
                   L.dmxq_bfp_qdq(vp(yf[i % k2].data_ptr()), vp(yf[(i + 1) % k2].data_ptr()), _lib.F32, _lib.F32, R, C, 1, 64, 8, 2, 1, 0, sp)), k2, n * 6)
    f16 = _lib.FloatFmt(10, 5, 15, 1)
    pf = ctypes.cast(ctypes.pointer(f16), ctypes.c_void_p)
    run("binary_cast ResAdd: FLOAT16 casts on both inputs and the output, bf16 (one launch, 6 B/elem)",
        lambda i: L.dmxq_binary_cast(vp(xs[i].data_ptr()), vp(xs[(i + 1) % k].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, n, 0, pf, pf, pf, sp), k, n * 6)
    run("  the four launches it replaces (3 x float_qdq FLOAT16 + torch add)",
        lambda i: (L.dmxq_float_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, n, 10, 5, 15, 1, 0, 2, 0, sp),
                   L.dmxq_float_qdq(vp(xs[(i + 1) % k].data_ptr()), vp(ys[(i + 1) % k].data_ptr()), _lib.BF16, _lib.BF16, n, 10, 5, 15, 1, 0, 2, 0, sp),
                   torch.add(ys[i], ys[(i + 1) % k], out=ys[(i + 2) % k]),
                   L.dmxq_float_qdq(vp(ys[(i + 2) % k].data_ptr()), vp(ys[(i + 3) % k].data_ptr()), _lib.BF16, _lib.BF16, n, 10, 5, 15, 1, 0, 2, 0, sp)), k, n * 6)
    f32a = [torch.randn(R, C, device=dev) for _ in range(6)]
    f32o = [torch.empty(R, C, device=dev) for _ in range(6)]
    run("binary_cast ResAdd on float32 tensors, three FLOAT16 casts (general form: one launch, 12 B/elem)",
        lambda i: L.dmxq_binary_cast(vp(f32a[i % 6].data_ptr()), vp(f32a[(i + 1) % 6].data_ptr()), vp(f32o[i % 6].data_ptr()), _lib.F32, n, 0, pf, pf, pf, sp), 6, n * 12)
    run("  the four launches it replaces (3 x float_qdq FLOAT16 fp32 + torch add)",
        lambda i: (L.dmxq_float_qdq(vp(f32a[i % 6].data_ptr()), vp(f32o[i % 6].data_ptr()), _lib.F32, _lib.F32, n, 10, 5, 15, 1, 0, 2, 0, sp),
                   L.dmxq_float_qdq(vp(f32a[(i + 1) % 6].data_ptr()), vp(f32o[(i + 1) % 6].data_ptr()), _lib.F32, _lib.F32, n, 10, 5, 15, 1, 0, 2, 0, sp),
                   torch.add(f32o[i % 6], f32o[(i + 1) % 6], out=f32o[(i + 2) % 6]),
                   L.dmxq_float_qdq(vp(f32o[(i + 2) % 6].data_ptr()), vp(f32o[(i + 3) % 6].data_ptr()), _lib.F32, _lib.F32, n, 10, 5, 15, 1, 0, 2, 0, sp)), 6, n * 12)
    run("relu_cast ReLU module: FLOAT16 input and output casts, bf16 (one launch, 4 B/elem; 3 launches unfused)",
        lambda i: L.dmxq_relu_cast(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, n, pf, pf, sp), k, n * 4)
    ws = torch.empty(L.dmxq_topk_workspace_bytes(n) // 8 + 1, dtype=torch.int64, device=dev)
    run("topk_sparsify TOPK{0.5} fp32 score, bf16 x -> bf16 y (radix select + apply; 4 score reads)",
        lambda i: L.dmxq_topk_mask(vp(ss[i % k2].data_ptr()), _lib.F32, vp(xs[i].data_ptr()), _lib.BF16, None, 0, vp(ys[i].data_ptr()), _lib.BF16, n, n // 2, vp(ws.data_ptr()), sp), k2, n * 8)
    # ---------------------------------------------------------------- composite block formats, packed BFP
    run("sbfp_qdq bf16 SBFP12_16 (XP[4,0] codes, FP[0|4|4,7] scaler) [weight storage rule]",
        lambda i: L.dmxq_sbfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, 16, 4, 1, 1, 4, 4, 7, 1, sp), k, n * 4)
    run("mxfp_qdq bf16 MXFP8[E4M3]{32}",
        lambda i: L.dmxq_mxfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, 32, 3, 4, sp), k, n * 4)
    run("mxfp_qdq bf16 MXFP4[E2M1]{32}",
        lambda i: L.dmxq_mxfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, 32, 1, 2, sp), k, n * 4)
    mant = [torch.empty(R, C, dtype=torch.int8, device=dev) for _ in range(k)]
    exps = [torch.empty(R, C // 16, dtype=torch.uint8, device=dev) for _ in range(k)]
    run("bfp_pack bf16 -> int8 codes + uint8 exponents, B=16",
        lambda i: L.dmxq_bfp_pack(vp(xs[i].data_ptr()), _lib.BF16, vp(mant[i].data_ptr()), vp(exps[i].data_ptr()), R, C, 16, 8, 1, sp), k, n * 3 + n // 16)
    run("bfp_unpack int8 codes + uint8 exponents -> bf16, B=16",
        lambda i: L.dmxq_bfp_unpack(vp(mant[i].data_ptr()), vp(exps[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, R, C, 16, 8, sp), k, n * 3 + n // 16)
    run("bfp_pack float32 -> int8 codes + uint8 exponents, B=64 (5 B/elem)",
        lambda i: L.dmxq_bfp_pack(vp(f32a[i % 6].data_ptr()), _lib.F32, vp(mant[i % k].data_ptr()), vp(exps[i % k].data_ptr()), R, C, 64, 8, 1, sp), 6, n * 5 + n // 64)
    run("bfp_unpack int8 codes + uint8 exponents -> float32, B=64",
        lambda i: L.dmxq_bfp_unpack(vp(mant[i % k].data_ptr()), vp(exps[i % k].data_ptr()), vp(f32o[i % 6].data_ptr()), _lib.F32, R, C, 64, 8, sp), 6, n * 5 + n // 64)
    run("scale_channels bf16 -> float32, x / s along last dim (the unfused SmoothQuant input scaling of a bf16 model; 6 B/elem)",
        lambda i: L.dmxq_scale_channels(vp(xs[i].data_ptr()), vp(f32o[i % 6].data_ptr()), _lib.BF16, _lib.F32, R, C, 1, vp(scc.data_ptr()), 1, sp), 6, n * 6)
    run("float_qdq bf16 -> float32 E4M3 (widening output; 6 B/elem)",
        lambda i: L.dmxq_float_qdq(vp(xs[i].data_ptr()), vp(f32o[i % 6].data_ptr()), _lib.BF16, _lib.F32, n, 3, 4, 7, 0, 0, 2, 0, sp), 6, n * 6)
    run("unary silu bf16", lambda i: L.dmxq_unary(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, n, 2, ctypes.c_float(0.0), sp), k, n * 4)
    run("unary quick_gelu bf16", lambda i: L.dmxq_unary(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, n, 3, ctypes.c_float(0.0), sp), k, n * 4)
    wr = torch.ones(C, device=dev, dtype=torch.bfloat16)
    run("rmsnorm bf16 rows of 4096 (Llama hidden)", lambda i: L.dmxq_rmsnorm(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, _lib.BF16, R, C, vp(wr.data_ptr()), _lib.BF16, ctypes.c_float(1e-5), sp), k, n * 4)
    # ---------------------------------------------------------------- reductions
    mn = torch.empty(R, device=dev)
    mx = torch.empty(R, device=dev)
    run("group_minmax bf16 group_size=128 along dim0",
        lambda i: L.dmxq_group_minmax(vp(xs[i].data_ptr()), _lib.BF16, 1, R, C, 128, vp(mn.data_ptr()), vp(mx.data_ptr()), sp), k, n * 2)
    mn.fill_(float("inf")); mx.fill_(float("-inf"))
    run("group_minmax_accumulate bf16 group_size=128 (a MinMaxObserver step in ONE launch: reduction + running min / max)",
        lambda i: L.dmxq_group_minmax_accumulate(vp(xs[i].data_ptr()), _lib.BF16, 1, R, C, 128, vp(mn.data_ptr()), vp(mx.data_ptr()), sp), k, n * 2)
    run("group_minmax_accumulate bf16 per-tensor",
        lambda i: L.dmxq_group_minmax_accumulate(vp(xs[i].data_ptr()), _lib.BF16, 1, 1, n, 1, vp(mn.data_ptr()), vp(mx.data_ptr()), sp), k, n * 2)
    run("group_minmax bf16 per-tensor",
        lambda i: L.dmxq_group_minmax(vp(xs[i].data_ptr()), _lib.BF16, 1, 1, n, 1, vp(mn.data_ptr()), vp(mx.data_ptr()), sp), k, n * 2)
    ma = torch.empty(C, device=dev)
    run("channel_maxabs bf16 along last dim (reduce over rows)",
        lambda i: L.dmxq_channel_maxabs(vp(xs[i].data_ptr()), _lib.BF16, R, C, 1, vp(ma.data_ptr()), sp), k, n * 2)
    hist = torch.empty(2048, device=dev)
    run("histc bf16 2048 bins over [-4, 4] (HistogramObserver pass)",
        lambda i: L.dmxq_histc(vp(xs[i].data_ptr()), _lib.BF16, n, 2048, -4.0, 4.0, vp(hist.data_ptr()), sp), k, n * 2)
    # ---------------------------------------------------------------- row ops (Whisper shapes)
    rows, cols = 12 * 1500, 1500
    xr = [torch.randn(rows, cols, device=dev).to(torch.bfloat16) for _ in range(10)]
    yr = [torch.empty_like(t) for t in xr]
    run("softmax bf16 rows of 1500 (12 heads x 1500)",
        lambda i: L.dmxq_softmax(vp(xr[i].data_ptr()), vp(yr[i].data_ptr()), _lib.BF16, _lib.BF16, rows, cols, ctypes.c_float(-math.inf), sp), 10, rows * cols * 4)
    rows2, cols2 = 16 * 1500, 768
    xl = [torch.randn(rows2, cols2, device=dev).to(torch.bfloat16) for _ in range(14)]
    yl = [torch.empty_like(t) for t in xl]
    w = torch.ones(cols2, device=dev, dtype=torch.bfloat16)
    run("layernorm bf16 rows of 768",
        lambda i: L.dmxq_layernorm(vp(xl[i].data_ptr()), vp(yl[i].data_ptr()), _lib.BF16, _lib.BF16, rows2, cols2, vp(w.data_ptr()), vp(w.data_ptr()), _lib.BF16, ctypes.c_float(1e-5), sp), 14, rows2 * cols2 * 4)
    # ---------------------------------------------------------------- activation / normalisation MODULES in one launch (row a9)
    # cast_in -> f -> cast_out with the BASIC rules' FLOAT16 casts; next to each, the three launches it replaces (this library's two
    # casts around torch's own GPU function), both at the fused kernel's algorithmic bytes (in + out)
    F = torch.nn.functional
    fq = lambda a, o, dt, m: L.dmxq_float_qdq(vp(a.data_ptr()), vp(o.data_ptr()), dt, dt, m, 10, 5, 15, 1, 0, 2, 0, sp)
    for kind, nm, tf in ((0, "gelu", lambda a, o: torch.nn.functional.gelu(a)), (2, "silu", lambda a, o: torch.nn.functional.silu(a))):
        run(f"unary_cast {nm} module: FLOAT16 -> {nm} -> FLOAT16, bf16 (one launch, 4 B/elem)",
            lambda i: L.dmxq_unary_cast(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, n, kind, ctypes.c_float(0.0), pf, pf, sp), k, n * 4)
        run(f"  the three launches it replaces (float_qdq, torch {nm}, float_qdq)",
            lambda i: (fq(xs[i], ys[i], _lib.BF16, n), fq(tf(ys[i], None), ys[(i + 1) % k], _lib.BF16, n)), k, n * 4)
    # the same modules as a 65,536-entry table lookup (csrc/lut16.hip): the table is built once (not timed), any function costs the same
    lut = torch.empty(65536, dtype=torch.int16, device=dev)
    for kind, nm in ((0, "gelu"), (3, "quick_gelu")):
        assert L.dmxq_unary_cast_table(_lib.BF16, kind, ctypes.c_float(0.0), pf, pf, vp(lut.data_ptr()), sp) == 0
        torch.cuda.synchronize()
        run(f"lut16_apply {nm} module: FLOAT16 -> {nm} -> FLOAT16, bf16, as a table lookup (correctly rounded; 4 B/elem)",
            lambda i: L.dmxq_lut16_apply(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), n, vp(lut.data_ptr()), sp), k, n * 4)
    run("unary_cast gelu module on float32 tensors (general form, 8 B/elem)",
        lambda i: L.dmxq_unary_cast(vp(f32a[i % 6].data_ptr()), vp(f32o[i % 6].data_ptr()), _lib.F32, n, 0, ctypes.c_float(0.0), pf, pf, sp), 6, n * 8)
    run("  the three launches it replaces (float_qdq fp32, torch gelu, float_qdq fp32)",
        lambda i: (fq(f32a[i % 6], f32o[i % 6], _lib.F32, n), fq(F.gelu(f32o[i % 6]), f32o[(i + 1) % 6], _lib.F32, n)), 6, n * 8)
    run("softmax_cast module bf16 rows of 1500: FLOAT16 -> softmax -> FLOAT16 (one launch)",
        lambda i: L.dmxq_softmax_cast(vp(xr[i].data_ptr()), vp(yr[i].data_ptr()), _lib.BF16, rows, cols, ctypes.c_float(-math.inf), pf, pf, sp), 10, rows * cols * 4)
    run("  the three launches it replaces (float_qdq, torch softmax, float_qdq)",
        lambda i: (fq(xr[i], yr[i], _lib.BF16, rows * cols), fq(torch.softmax(yr[i], -1), yr[(i + 1) % 10], _lib.BF16, rows * cols)), 10, rows * cols * 4)
    xr32 = [torch.randn(rows, cols, device=dev) for _ in range(5)]
    yr32 = [torch.empty_like(t) for t in xr32]
    run("softmax_cast module float32 rows of 1500 (Whisper attention, general form, 8 B/elem)",
        lambda i: L.dmxq_softmax_cast(vp(xr32[i].data_ptr()), vp(yr32[i].data_ptr()), _lib.F32, rows, cols, ctypes.c_float(-math.inf), pf, pf, sp), 5, rows * cols * 8)
    run("  the three launches it replaces (float_qdq fp32, torch softmax, float_qdq fp32)",
        lambda i: (fq(xr32[i], yr32[i], _lib.F32, rows * cols), fq(torch.softmax(yr32[i], -1), yr32[(i + 1) % 5], _lib.F32, rows * cols)), 5, rows * cols * 8)
    run("bfp_qdq float32 rows of 1500, B=64 (ragged last block: Whisper attention probabilities)",
        lambda i: L.dmxq_bfp_qdq(vp(xr32[i].data_ptr()), vp(yr32[i].data_ptr()), _lib.F32, _lib.F32, rows, cols, 1, 64, 8, 2, 1, 0, sp), 5, rows * cols * 8)
    run("softmax_cast_bfp float32 rows of 1500: the softmax module AND the consumer's BFP16_64 input cast in one launch (8 B/elem)",
        lambda i: L.dmxq_softmax_cast_bfp(vp(xr32[i].data_ptr()), vp(yr32[i].data_ptr()), _lib.F32, rows, cols, ctypes.c_float(-math.inf), pf, pf, 64, 8, sp), 5, rows * cols * 8)
    run("  the two launches it replaces (softmax_cast, then bfp_qdq of the probabilities: 16 B/elem)",
        lambda i: (L.dmxq_softmax_cast(vp(xr32[i].data_ptr()), vp(yr32[i].data_ptr()), _lib.F32, rows, cols, ctypes.c_float(-math.inf), pf, pf, sp),
                   L.dmxq_bfp_qdq(vp(yr32[i].data_ptr()), vp(yr32[(i + 1) % 5].data_ptr()), _lib.F32, _lib.F32, rows, cols, 1, 64, 8, 2, 1, 0, sp)), 5, rows * cols * 8)
    del xr32, yr32
    run("float_qdq float32 FP16(FN) [BASIC activation cast of a float32 model]",
        lambda i: fq(f32a[i % 6], f32o[i % 6], _lib.F32, n), 6, n * 8)
    xl32 = [torch.randn(rows2, cols2, device=dev) for _ in range(8)]
    yl32 = [torch.empty_like(t) for t in xl32]
    w32 = torch.ones(cols2, device=dev)
    run("layernorm_cast module float32 rows of 768 (opt-125m / Whisper, general form, 8 B/elem)",
        lambda i: L.dmxq_layernorm_cast(vp(xl32[i].data_ptr()), vp(yl32[i].data_ptr()), _lib.F32, rows2, cols2, vp(w32.data_ptr()), vp(w32.data_ptr()), ctypes.c_float(1e-5), pf, pf, sp), 8, rows2 * cols2 * 8)
    del xl32, yl32
    run("layernorm_cast module bf16 rows of 768 (one launch)",
        lambda i: L.dmxq_layernorm_cast(vp(xl[i].data_ptr()), vp(yl[i].data_ptr()), _lib.BF16, rows2, cols2, vp(w.data_ptr()), vp(w.data_ptr()), ctypes.c_float(1e-5), pf, pf, sp), 14, rows2 * cols2 * 4)
    run("  the three launches it replaces (float_qdq, torch layer_norm, float_qdq)",
        lambda i: (fq(xl[i], yl[i], _lib.BF16, rows2 * cols2), fq(F.layer_norm(yl[i], (cols2,), w, w, 1e-5), yl[(i + 1) % 14], _lib.BF16, rows2 * cols2)), 14, rows2 * cols2 * 4)
    run("rmsnorm_cast module bf16 rows of 4096 (Llama hidden, one launch)",
        lambda i: L.dmxq_rmsnorm_cast(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), _lib.BF16, R, C, vp(wr.data_ptr()), ctypes.c_float(1e-5), pf, pf, sp), k, n * 4)
    run("  the three launches it replaces (float_qdq, torch rms_norm, float_qdq)",
        lambda i: (fq(xs[i], ys[i], _lib.BF16, n), fq(F.rms_norm(ys[i], (C,), wr, 1e-5), ys[(i + 1) % k], _lib.BF16, n)), k, n * 4)
    if args.json:
        json.dump(results, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
