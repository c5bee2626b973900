#!/usr/bin/env python3
"""tools/bench_tier2.py — the SECOND TIER of bench.py's line (round 6): the kernels BASELINE.json configs 3 / 4 / 5
exercise, timed inside the driver's own `python bench.py` run right after the headline (config 2) measurement, so that their numbers
are driver-observed and not only builder-run tables under profiles/.

    python bench.py                       -> the one JSON line gains  "ops": {name: {us, bytes, GB/s, frac, check, ...}}
                                                                      "layers": {model: {live: {eager, graph}, ...}}
    python tools/bench_tier2.py [--only c3,c5] [--no-layers]          the same dict, stand-alone (pretty-printed)

Method per op = tools/bench_ops.py's (the table under profiles/ it is compared with): direct C-ABI launches (include/dmxq.h) on one
stream with preallocated outputs, rotating over buffer sets that together exceed the 256 MiB Infinity Cache, warmed by GPU time, HIP
events on the launch stream around 5 groups of `iters` launches, the median group; `bytes` = ALGORITHMIC bytes per launch (inputs
read once + outputs written once), `frac` = bytes / us / 8 TB/s.  Inputs are device-generated N(0,1) * exp(2 N(0,1)) ("heavy") unless
noted.  Every op is CHECKED outside the timed region on rotation slot 0: bit-exact against the CPU oracle (oracle/, the checker --
never the thing measured) for the Q->DQ / mask kernels, and for the approximator-slot modules (softmax / LayerNorm / GELU between two
FLOAT16 casts) within the contract tests/test_gpu_act_cast.py states (float64 truth on the oracle-cast input, oracle output cast,
tests/_data.py outside_cast_bracket).  A failing check or a failing launch is recorded in the op's entry ("check": "FAILED: ...") and
never raises: the headline line must come out whatever happens here.

Reference paths these ops replace: numerical/cast.py:278-296 (INT8 group affine), sparse.py:163-180 + numerical/format.py:304-343
(2:4 mask -> BFP), modeling/nn/core.py:178-198 (weight hypernet), :228-232 (SmoothQuant input scaling -> input cast),
modeling/nn/torch_modules.py:989-998 (Softmax), :1062-1082 (LayerNorm), functional/approximate.py:300-327 (GELU module).
"""
import ctypes
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK = 8.0e12
vp = ctypes.c_void_p
LLAMA_LAYER = [("q_proj", 4096, 4096), ("k_proj", 1024, 4096), ("v_proj", 1024, 4096), ("o_proj", 4096, 4096),
               ("gate_proj", 14336, 4096), ("up_proj", 14336, 4096), ("down_proj", 4096, 14336)]
OPT_LAYER = [("q_proj", 768, 768), ("k_proj", 768, 768), ("v_proj", 768, 768), ("out_proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]


def heavy(shape, seed, dev, dtype, spread=2.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    t = torch.randn(*shape, generator=g, device=dev)
    if spread:
        t = t * torch.exp(spread * torch.randn(*shape, generator=g, device=dev))
    return t.to(dtype)


def sets_for(per_set_bytes, lo=2, hi=24):
    """buffer sets needed for the rotation to exceed 512 MiB (2 x the Infinity Cache)"""
    return max(lo, min(hi, math.ceil(512 * 2 ** 20 / per_set_bytes)))


class Timer:
    def __init__(self, dev, iters=100, warm_ms=25.0):
        self.dev, self.iters, self.warm_ms = dev, iters, warm_ms
        self.stream = torch.cuda.Stream(device=dev)
        self.sp = vp(self.stream.cuda_stream)

    def time(self, launch, nbuf, iters=None):
        """us per launch (median of 5 event-timed groups) and the groups"""
        iters = iters or self.iters
        st = self.stream
        with torch.cuda.stream(st):
            warmed = 0.0
            while warmed < self.warm_ms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for i in range(max(10, iters // 2)):
                    launch(i % nbuf)
                e1.record(st)
                torch.cuda.synchronize(self.dev)
                warmed += max(e0.elapsed_time(e1), 0.05)
            groups = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for i in range(iters):
                    launch(i % nbuf)
                e1.record(st)
                torch.cuda.synchronize(self.dev)
                groups.append(e0.elapsed_time(e1) * 1e3 / iters)
        return statistics.median(groups), groups


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} returned {rc}")


def run_ops(dev, only=None, iters=100, log=None):
    """-> {name: {...}} for the second-tier ops.  `only`: iterable of substrings (config tags c3 / c4 / c5 / basic or op names)"""
    import oracle as O
    from _data import bits_equal, outside_cast_bracket

    from dmx_compressor_amd import _lib

    L = _lib.lib()
    BF16, F32 = _lib.BF16, _lib.F32
    T = Timer(dev, iters)
    sp = T.sp
    out = {}

    def want(name):
        """an op name, or a block tag (c3 / c4 / c5): a filter term selects the ops containing it and the blocks that may hold them"""
        return only is None or any(t in name or (len(name) == 2 and (t.startswith(name) or t[:2] not in ("c3", "c4", "c5"))) for t in only)

    def record(name, config, nbytes, launch, nbuf, check, note=None, iters=None):
        if not want(name):
            return
        t0 = time.perf_counter()
        ent = {"config": config, "bytes": int(nbytes)}
        try:
            us, groups = T.time(launch, nbuf, iters)
            ent.update({"us": round(us, 2), "GB/s": round(nbytes / us / 1e3, 1), "frac": round(nbytes / (us * 1e-6) / PEAK, 4),
                        "us_groups": [round(g, 2) for g in groups], "buffer_sets": nbuf})
        except Exception as e:   # noqa: BLE001 -- recorded, never raised (see the module docstring)
            ent["error"] = f"{type(e).__name__}: {str(e)[:200]}"
        try:
            with torch.cuda.stream(T.stream):
                launch(0)
            torch.cuda.synchronize(dev)
            ent["check"] = check()
        except Exception as e:   # noqa: BLE001
            ent["check"] = f"FAILED: {type(e).__name__}: {str(e)[:300]}"
        if note:
            ent["note"] = note
        ent["wall_s"] = round(time.perf_counter() - t0, 2)
        out[name] = ent
        if log:
            log(f"{name:64s} {ent.get('us', float('nan')):9.2f} us {100 * ent.get('frac', 0):6.1f} %  {ent['check'][:70]}")

    def exact(got, wanted, what):
        bad = bits_equal(got.cpu(), wanted)
        if bad:
            raise AssertionError(f"{bad} of {got.numel()} elements differ from {what}")
        return f"{got.numel()} elements bit-exact vs {what}"

    R = C = 4096
    n = R * C
    f16 = _lib.FloatFmt(10, 5, 15, 1)
    pf = ctypes.cast(ctypes.pointer(f16), vp)
    cast16 = lambda t: O.floating_point_cast(t, 10, 5, 15, True).to(t.dtype)   # noqa: E731  (CastTo with FLOAT16 = FP[1|5|10,15](FN))

    # ------------------------------------------------------------------ config 3: opt-125m, INT8 group-128 Linear weights
    if want("c3"):
        # (a) one decoder layer's six float32 weights AND its six bias casts (BASIC: BFP32_1 = float_quantize with 22 mantissa bits) in ONE
        #     launch: what nn.LiveWeightBatch issues per forward for un-folded weights
        k = sets_for(2 * 4 * sum(r * c for _, r, c in OPT_LAYER))
        layers = []
        for s in range(k):
            ws = [heavy((r, c), 100 * s + i, dev, torch.float32, spread=0.0) * 0.05 for i, (_, r, c) in enumerate(OPT_LAYER)]
            bs = [heavy((r,), 100 * s + 50 + i, dev, torch.float32, spread=0.0) * 0.02 for i, (_, r, c) in enumerate(OPT_LAYER)]
            scs = [(w.reshape(-1, 128, w.shape[1]).abs().amax(dim=(1, 2)) / 127.0).contiguous() for w in ws]   # MinMax, per_tensor_symmetric per slab
            zps = [torch.zeros(sc.numel(), dtype=torch.int64, device=dev) for sc in scs]
            wo, bo = [torch.empty_like(w) for w in ws], [torch.empty_like(b) for b in bs]
            ad = (_lib.AffineDesc * len(ws))()
            for d, w, o, sc, zp in zip(ad, ws, wo, scs, zps):
                d.in_, d.out, d.scale, d.zero_point, d.outer, d.C, d.inner = w.data_ptr(), o.data_ptr(), sc.data_ptr(), zp.data_ptr(), 1, w.shape[0], w.shape[1]
            fd = (_lib.TensorDesc * len(bs))()
            for d, b, o in zip(fd, bs, bo):
                d.in_, d.out, d.outer, d.L, d.inner = b.data_ptr(), o.data_ptr(), 1, b.numel(), 1
            layers.append((ws, bs, scs, zps, wo, bo, ad, fd))

        def launch_c3(i):
            ws, bs, scs, zps, wo, bo, ad, fd = layers[i]
            _ok(L.dmxq_fixed_float_qdq_multi(ad, len(ws), 8, 0, 1, 1, 2, 128, fd, len(bs), 22, 8, 127, 0, 0, 2, F32, 0, sp), "dmxq_fixed_float_qdq_multi")

        def check_c3():
            ws, bs, scs, zps, wo, bo, _, _ = layers[0]
            for w, o, sc, zp in zip(ws, wo, scs, zps):
                exact(o, O.fixed_point_affine_cast(w.cpu(), 8, 0, True, True, sc.cpu(), zp.cpu(), ch_axis=0, group_size=128), "oracle.fixed_point_affine_cast")
            for b, o in zip(bs, bo):
                exact(o, O.float_quantize(b.cpu(), 22, 8, 127, False), "oracle.float_quantize")
            return "6 weights + 6 biases bit-exact vs the oracle (fixed_point_affine_cast group_size 128 / float_quantize)"

        nel = sum(r * c + r for _, r, c in OPT_LAYER)
        record("c3.int8_group128_layer_multi", "opt-125m decoder layer: 6 float32 weights INT8 group-128 (rows) + 6 BFP32_1 bias casts, ONE launch "
               "(dmxq_fixed_float_qdq_multi)", 8 * nel, launch_c3, k, check_c3)
        del layers
        # (b) the same cast on the headline operand
        k = sets_for(n * 4)
        xs = [heavy((R, C), 10 + i, dev, torch.bfloat16) for i in range(k)]
        ys = [torch.empty_like(x) for x in xs]
        sc = (torch.rand(R // 128, device=dev) * 0.05 + 0.01)
        zp = torch.zeros(R // 128, dtype=torch.int64, device=dev)
        record("c3.int8_group128_4096x4096_bf16", "INT8 group_size 128 along dim 0, 4096x4096 bf16 -> bf16 (dmxq_fixed_qdq)", n * 4,
               lambda i: _ok(L.dmxq_fixed_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), BF16, BF16, 1, R, C, 8, 0, 1, 1, 2, vp(sc.data_ptr()), vp(zp.data_ptr()), 128, 0, sp), "dmxq_fixed_qdq"),
               k, lambda: exact(ys[0], O.fixed_point_affine_cast(xs[0].cpu(), 8, 0, True, True, sc.cpu(), zp.cpu(), ch_axis=0, group_size=128).to(torch.bfloat16),
                                "oracle.fixed_point_affine_cast"))
    else:
        k = sets_for(n * 4)
        xs = [heavy((R, C), 10 + i, dev, torch.bfloat16) for i in range(k)]
        ys = [torch.empty_like(x) for x in xs]

    # ------------------------------------------------------------------ BASIC rules on the headline operand
    record("basic.float16_activation_cast_bf16", "FLOAT16 = FP[1|5|10,15](FN) activation cast, 4096x4096 bf16 -> bf16 (dmxq_float_qdq)", n * 4,
           lambda i: _ok(L.dmxq_float_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), BF16, BF16, n, 10, 5, 15, 1, 0, 2, 0, sp), "dmxq_float_qdq"),
           k, lambda: exact(ys[0], cast16(xs[0].cpu()), "oracle.floating_point_cast"))
    record("c4.bfp16_64_activation_cast_bf16", "BFP[8|8]{64}(SN) input cast of a Linear, 4096x4096 bf16 -> bf16 (dmxq_bfp_qdq)", n * 4,
           lambda i: _ok(L.dmxq_bfp_qdq(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), BF16, BF16, R, C, 1, 64, 8, 2, 1, 0, sp), "dmxq_bfp_qdq"),
           k, lambda: exact(ys[0], O.bfp_cast(xs[0].cpu(), 8, 64).to(torch.bfloat16), "oracle.bfp_cast"))

    # ------------------------------------------------------------------ config 4: Llama-3-8B, 2:4 N:M weight mask -> BFP16_64
    if want("c4"):
        ss = [torch.rand(R, C, generator=torch.Generator(device=dev).manual_seed(70 + i), device=dev) for i in range(max(2, k // 2))]
        k2 = len(ss)
        record("c4.nm_sparsify_2:4_fp32_score_bf16", "BTOPK{2:4,-1} mask from a float32 score applied to a bf16 weight, 4096x4096 (dmxq_nm_mask, fused apply)", n * 8,
               lambda i: _ok(L.dmxq_nm_mask(vp(ss[i % k2].data_ptr()), F32, vp(xs[i].data_ptr()), BF16, None, 0, vp(ys[i].data_ptr()), BF16, R, C, 1, 2, 4, sp), "dmxq_nm_mask"),
               k2, lambda: exact(ys[0], O.sparsify(xs[0].cpu(), ss[0].cpu(), 2, 4).to(torch.bfloat16), "oracle.sparsify"))
        del ss
        rows, cols = 14336, 4096
        ws = [heavy((rows, cols), 200 + i, dev, torch.bfloat16) for i in range(2)]
        sf = [torch.rand(rows, cols, generator=torch.Generator(device=dev).manual_seed(210 + i), device=dev) for i in range(2)]
        wo = [torch.empty_like(w) for w in ws]
        record("c4.weight_hypernet_2:4_bfp16_64_14336x4096", "gate/up_proj weight [14336, 4096] bf16, float32 score: 2:4 mask -> BFP[8|8]{64}(SN) in one launch "
               "(dmxq_weight_hypernet; 2 + 4 + 2 B/element)", rows * cols * 8,
               lambda i: _ok(L.dmxq_weight_hypernet(vp(ws[i].data_ptr()), BF16, vp(sf[i].data_ptr()), F32, 2, 4, None, vp(wo[i].data_ptr()), BF16, rows, cols, 64, 8, 1, sp), "dmxq_weight_hypernet"),
               2, lambda: exact(wo[0], O.bfp_cast(O.sparsify(ws[0].cpu(), sf[0].cpu(), 2, 4).to(torch.bfloat16), 8, 64).to(torch.bfloat16), "oracle sparsify -> bfp_cast"),
               iters=50)
        del ws, sf, wo
        # the whole decoder layer's seven weights in ONE launch (bf16 scores, as `bench.py --workload llama-shard`)
        copies = []
        for c in range(2):
            ts = []
            for t, (_, r, cc) in enumerate(LLAMA_LAYER):
                w = heavy((r, cc), 300 + 10 * c + t, dev, torch.bfloat16)
                s = torch.rand(r, cc, generator=torch.Generator(device=dev).manual_seed(400 + 10 * c + t), device=dev).to(torch.bfloat16)
                ts.append((w, s, torch.empty_like(w)))
            hd = (_lib.HypernetDesc * len(ts))()
            for d, (w, s, o) in zip(hd, ts):
                d.w, d.score, d.sq_scale, d.out, d.rows, d.L = w.data_ptr(), s.data_ptr(), None, o.data_ptr(), w.shape[0], w.shape[1]
            copies.append((ts, hd))
        nel = sum(r * c for _, r, c in LLAMA_LAYER)

        def check_multi():
            for (w, s, o), (nm, _, _) in zip(copies[0][0], LLAMA_LAYER):
                exact(o, O.bfp_cast(O.sparsify(w.cpu(), s.cpu(), 2, 4), 8, 64).to(torch.bfloat16), f"oracle sparsify -> bfp_cast ({nm})")
            return f"7 weights ({nel} elements) bit-exact vs oracle sparsify -> bfp_cast"

        record("c4.weight_hypernet_multi_llama_layer", "Llama-3-8B decoder layer, 7 weights (218.1 M elements, bf16 w + bf16 score -> bf16) in ONE launch "
               "(dmxq_weight_hypernet_multi)", nel * 6,
               lambda i: _ok(L.dmxq_weight_hypernet_multi(copies[i][1], 7, BF16, BF16, 2, 4, BF16, 64, 8, 1, sp), "dmxq_weight_hypernet_multi"),
               2, check_multi, iters=20)
        del copies
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ config 5: Whisper-small approximator-slot modules + SmoothQuant input path
    if want("c5"):
        sq = (torch.rand(C, device=dev) + 0.5)
        yf = [torch.empty(R, C, device=dev) for _ in range(max(2, k // 2))]
        kf = len(yf)

        def check_ih():
            xs0 = xs[0].cpu().float() / sq.cpu()[None, :]
            return exact(yf[0], O.bfp_cast(xs0, 8, 64), "x / s in float32 -> oracle.bfp_cast")

        record("c5.input_hypernet_smoothquant_bfp16_64", "SmoothQuant x / s + BFP[8|8]{64}(SN) input cast, 4096x4096 bf16 -> float32, one launch "
               "(dmxq_input_hypernet; 2 + 4 B/element)", n * 6,
               lambda i: _ok(L.dmxq_input_hypernet(vp(xs[i].data_ptr()), BF16, vp(sq.data_ptr()), vp(yf[i % kf].data_ptr()), F32, R, C, 64, 8, 1, sp), "dmxq_input_hypernet"),
               kf, check_ih)
        del yf
        # GELU module, bf16: the module's default path for 16-bit tensors is the correctly-rounded table (csrc/lut16.hip), built once, untimed
        lut = torch.empty(65536, dtype=torch.int16, device=dev)
        with torch.cuda.stream(T.stream):
            _ok(L.dmxq_unary_cast_table(BF16, 0, ctypes.c_float(0.0), pf, pf, vp(lut.data_ptr()), sp), "dmxq_unary_cast_table")
        torch.cuda.synchronize(dev)

        def check_gelu(y, x, dtype, n_ulp):
            xc = x.cpu()
            cin = cast16(xc)
            bad = outside_cast_bracket(y, torch.nn.functional.gelu(cin.double()), cast16, dtype, n_ulp, cin.double().abs() / 2)
            if bad:
                raise AssertionError(f"{bad} elements outside the cast bracket")
            return f"{y.numel()} elements within {n_ulp} ulp of FLOAT16(gelu_float64(FLOAT16(x))) (cast bracket, oracle casts)"

        record("c5.gelu_module_bf16_lut", "GELU module FLOAT16 -> gelu -> FLOAT16 on 4096x4096 bf16, the module's default table form (dmxq_lut16_apply)", n * 4,
               lambda i: _ok(L.dmxq_lut16_apply(vp(xs[i].data_ptr()), vp(ys[i].data_ptr()), n, vp(lut.data_ptr()), sp), "dmxq_lut16_apply"),
               k, lambda: check_gelu(ys[0], xs[0], torch.bfloat16, 1))
        # Whisper's own GELU operand: [1500, 3072] float32 per batch element -> here 8 of them (147 MB per launch)
        g_rows, g_cols = 8 * 1500, 3072
        kg = sets_for(g_rows * g_cols * 8)
        gx = [heavy((g_rows, g_cols), 500 + i, dev, torch.float32, spread=0.0) * 3.0 for i in range(kg)]
        gy = [torch.empty_like(t) for t in gx]
        record("c5.gelu_module_f32_8x1500x3072", "GELU module FLOAT16 -> gelu -> FLOAT16 on float32 [8, 1500, 3072] (Whisper fc1 output), one launch (dmxq_unary_cast)",
               g_rows * g_cols * 8,
               lambda i: _ok(L.dmxq_unary_cast(vp(gx[i].data_ptr()), vp(gy[i].data_ptr()), F32, g_rows * g_cols, 0, ctypes.c_float(0.0), pf, pf, sp), "dmxq_unary_cast"),
               kg, lambda: check_gelu(gy[0], gx[0], torch.float32, 64))
        del gx, gy
        # softmax over attention rows of 1500, float32 (what config 5 runs): 12 heads x 1500 rows
        rows, cols = 12 * 1500, 1500
        ks = sets_for(rows * cols * 8)
        xr = [heavy((rows, cols), 600 + i, dev, torch.float32, spread=0.0) * 3.0 for i in range(ks)]
        yr = [torch.empty_like(t) for t in xr]

        def check_sm():
            cin = cast16(xr[0].cpu())
            bad = outside_cast_bracket(yr[0], torch.softmax(cin.double(), -1), cast16, torch.float32, 64)
            if bad:
                raise AssertionError(f"{bad} elements outside the cast bracket")
            return f"{rows * cols} elements within 64 fp32 ulp (2^-17) of FLOAT16(softmax_float64(FLOAT16(x))) (cast bracket, oracle casts)"

        record("c5.softmax_module_f32_rows1500", "Softmax module FLOAT16 -> softmax -> FLOAT16 on float32 [12 x 1500, 1500], one launch (dmxq_softmax_cast)", rows * cols * 8,
               lambda i: _ok(L.dmxq_softmax_cast(vp(xr[i].data_ptr()), vp(yr[i].data_ptr()), F32, rows, cols, ctypes.c_float(-math.inf), pf, pf, sp), "dmxq_softmax_cast"),
               ks, check_sm)
        del xr, yr
        rows2, cols2 = 16 * 1500, 768
        for dt, code, tag, nulp in ((torch.float32, F32, "f32", 3), (torch.bfloat16, BF16, "bf16", 1)):
            esz = 4 if dt == torch.float32 else 2
            kl = sets_for(rows2 * cols2 * 2 * esz)
            xl = [(heavy((rows2, cols2), 700 + i, dev, torch.float32, spread=0.0) * 2.0 + 0.5).to(dt) for i in range(kl)]
            yl = [torch.empty_like(t) for t in xl]
            w = (torch.randn(cols2, device=dev) * 0.1 + 1.0).to(dt)
            b = (torch.randn(cols2, device=dev) * 0.1).to(dt)

            def check_ln(xl=xl, yl=yl, w=w, b=b, dt=dt, nulp=nulp):
                cin = cast16(xl[0].cpu()).double()
                wd, bd = w.cpu().double(), b.cpu().double()
                truth = torch.nn.functional.layer_norm(cin, (cols2,), wd, bd, 1e-5)
                mu, rstd = cin.mean(-1, keepdim=True), (cin.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
                floor = (cin.abs().amax(-1, keepdim=True) + mu.abs()) * rstd * wd.abs() + bd.abs()
                bad = outside_cast_bracket(yl[0], truth, cast16, dt, nulp, floor)
                if bad:
                    raise AssertionError(f"{bad} elements outside the cast bracket")
                return f"{rows2 * cols2} elements within {nulp} ulp of FLOAT16(layer_norm_float64(FLOAT16(x))) (cast bracket, oracle casts)"

            record(f"c5.layernorm_module_{tag}_rows768", f"LayerNorm module FLOAT16 -> layer_norm -> FLOAT16 on {tag} [16 x 1500, 768], one launch (dmxq_layernorm_cast)",
                   rows2 * cols2 * 2 * esz,
                   lambda i, xl=xl, yl=yl, w=w, b=b, code=code: _ok(L.dmxq_layernorm_cast(vp(xl[i].data_ptr()), vp(yl[i].data_ptr()), code, rows2, cols2, vp(w.data_ptr()), vp(b.data_ptr()),
                                                                                          ctypes.c_float(1e-5), pf, pf, sp), "dmxq_layernorm_cast"),
                   kl, check_ln)
            del xl, yl
    del xs, ys
    torch.cuda.empty_cache()
    return out


def run_layers(dev, models=("opt125m", "llama", "whisper"), steps=20, warmup=5, log=None):
    """one configured layer of each BASELINE config 3 / 4 / 5 end to end (tools/bench_layer.py): eager us and one-hipGraph us per forward,
    weights live (re-quantised every forward, per module and batched) and folded"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_layer", os.path.join(ROOT, "tools", "bench_layer.py"))
    bl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bl)
    res = {}
    for name in models:
        t0 = time.perf_counter()
        try:
            line = bl.run(name, steps, warmup, dev=dev, modes=("live", "folded"))
            res[name] = {"workload": line["config"]["workload"], "us_per_forward": line["layer_us"], "steps": steps}
        except Exception as e:   # noqa: BLE001
            res[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        res[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()
        if log:
            log(f"layer {name}: {json.dumps(res[name].get('us_per_forward', res[name]))}")
    return res


def run(dev, only=None, layers=True, log=None):
    t0 = time.perf_counter()
    ops = run_ops(dev, only, log=log)
    t1 = time.perf_counter()
    lay = run_layers(dev, log=log) if layers else None
    return {"ops": ops, "layers": lay,
            "tier2_method": "ops: tools/bench_tier2.py -- C-ABI launches on one stream, rotation over > 512 MiB of buffer sets, HIP events around 5 groups "
                            "of `iters` launches (median), frac = algorithmic bytes / us / 8 TB/s; every op checked on slot 0 outside the timed region "
                            "(oracle bit-exact, or the cast-bracket contract for softmax / LayerNorm / GELU modules).  layers: tools/bench_layer.py",
            "tier2_wall_s": {"ops": round(t1 - t0, 1), "layers": round(time.perf_counter() - t1, 1)}}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--no-layers", action="store_true")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    r = run(torch.device("cuda", 0), a.only.split(",") if a.only else None, layers=not a.no_layers, log=lambda s: print(s, file=sys.stderr, flush=True))
    if a.json:
        json.dump(r, open(a.json, "w"), indent=1)
    print(json.dumps(r))
