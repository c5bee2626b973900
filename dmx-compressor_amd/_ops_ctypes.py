"""ctypes binding of the tensor-level front ends (see ops.py): the C ABI of lib/libdmxq.so called directly, outputs
allocated with torch.empty, torch's current stream passed as a raw hipStream_t.  Same signatures and results as
`_ops_torch`; no autograd registration and no meta kernels (not traceable by torch.compile).

No CPU path: a non-GPU tensor or a missing libdmxq.so raises `DmxqError`.
"""
import functools
import math
from typing import Optional

import torch

from . import _lib
from ._lib import DmxqError, check, dtype_code, lib, ptr, require_gpu, split3, stream_of

__all__ = [
    "bfp_qdq", "block_quantize", "bfp_qdq_multi", "bfp_pack", "bfp_unpack", "weight_hypernet", "weight_hypernet_multi", "input_hypernet", "binary_cast", "rope_cast", "relu_cast", "unary_cast", "unary_cast_table", "lut16_apply", "softmax_cast", "layernorm_cast", "rmsnorm_cast", "sbfp_qdq", "mxfp_qdq", "float_qdq", "fixed_qdq", "fixed_qdq_multi", "nm_mask", "nm_sparsify", "topk_mask", "topk_sparsify", "bernoulli_mask", "group_minmax", "group_minmax_accumulate", "qparams", "channel_maxabs",
    "smoothquant_scale", "scale_channels", "gelu", "silu", "quick_gelu", "exp", "silu_experimental", "rope", "softmax", "layernorm",
    "rmsnorm", "histc",
]

_SEED_COUNTER = [0x5EED]


def _next_seed() -> int:
    """Stochastic rounding draws from a counter-based stream keyed by (seed, element index); a fresh seed per
    call, derived from `torch.initial_seed()` (the DEFAULT generator's seed: `torch.manual_seed` makes runs
    reproducible; a non-default `torch.Generator` is not consulted -- pass `seed=` explicitly for that).

    The seed is a kernel ARGUMENT, so a hipGraph capture would freeze it and every replay would repeat the same
    draws (accumulated rounding would no longer be unbiased): implicit seeding is refused while the current stream is
    capturing.  An explicit `seed=` is the caller's statement that frozen draws are intended."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise DmxqError("stochastic rounding with an implicit seed inside a hipGraph capture: the seed would be frozen "
                        "into the graph and every replay would reuse the same random draws; pass seed= explicitly or "
                        "keep stochastic casts outside the captured region")
    _SEED_COUNTER[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _SEED_COUNTER[0]) & 0xFFFFFFFFFFFFFFFF


def _prep(x: torch.Tensor, what: str) -> torch.Tensor:
    require_gpu(x, what)
    dtype_code(x.dtype)
    return x if x.is_contiguous() else x.contiguous()


def bfp_qdq(x, precision: int, block_size: int, block_dim: int = -1, symmetric: bool = True,
            rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """BlockFloatingPoint Q->DQ of a whole tensor in one launch (numerical/format.py:304-343 semantics).
    Blocks run along `block_dim`; output has x's shape, contiguous, dtype `out_dtype` (default x.dtype)."""
    xc = _prep(x, "bfp_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    if xc.dim() == 0:
        outer, L, inner = 1, 1, 1
    else:
        outer, L, inner = split3(xc.shape, block_dim)
    seed = _next_seed() if (seed is None and rounding == "stochastic") else (seed or 0)
    check(lib().dmxq_bfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, L, inner,
                             block_size, precision, _lib.ROUNDING_CODE[rounding], int(symmetric), seed, stream_of(xc)),
          "dmxq_bfp_qdq")
    return out


def block_quantize(a2d, wl: int, symmetric: bool = True, rounding: str = "nearest", seed: Optional[int] = None):
    """The pybind seam on a [rows, L] float32 view, one block per row (quant_cpu.cpp:299-311).  symmetric False is the
    NATIVE asymmetric branch (:247-253), which is not the asymmetric FORMAT of bfp_qdq(symmetric=False)."""
    xc = _prep(a2d, "block_quantize")
    out = torch.empty_like(xc)
    seed = _next_seed() if (seed is None and rounding == "stochastic") else (seed or 0)
    check(lib().dmxq_bfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.shape[0], xc.shape[1], 1,
                             max(xc.shape[1], 2), wl, _lib.ROUNDING_CODE[rounding], 1 if symmetric else 2, seed,
                             stream_of(xc)), "dmxq_bfp_qdq")
    return out


def bfp_qdq_multi(tensors, precision: int, block_size: int, block_dim: int = -1, symmetric: bool = True,
                  rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """BFP Q->DQ of MANY tensors of one dtype on one device in as few launches as possible (`dmxq_bfp_qdq_multi`): the
    same results as `[bfp_qdq(t, ...) for t in tensors]`.  Made for the many small weights of a model (each of them is
    launch-bound on its own)."""
    tensors = list(tensors)
    if not tensors:
        return []
    xs = [_prep(t, "bfp_qdq_multi") for t in tensors]
    dt, dev = xs[0].dtype, xs[0].device
    if any(x.dtype != dt or x.device != dev for x in xs):
        raise ValueError("bfp_qdq_multi: all tensors must share one dtype and one device")
    outs = [torch.empty(x.shape, dtype=out_dtype or dt, device=dev) for x in xs]
    descs = (_lib.TensorDesc * len(xs))()
    for d, x, o in zip(descs, xs, outs):
        outer, L, inner = split3(x.shape, block_dim) if x.dim() else (1, 1, 1)
        d.in_, d.out, d.outer, d.L, d.inner = x.data_ptr(), o.data_ptr(), outer, L, inner
    seed = _next_seed() if (seed is None and rounding == "stochastic") else (seed or 0)
    with torch.cuda.device(dev):
        check(lib().dmxq_bfp_qdq_multi(descs, len(xs), dtype_code(dt), dtype_code(outs[0].dtype), block_size, precision,
                                       _lib.ROUNDING_CODE[rounding], int(symmetric), seed, stream_of(xs[0])),
              "dmxq_bfp_qdq_multi")
    return outs


def bfp_pack(x, precision: int, block_size: int, symmetric: bool = True):
    """Packed on-wire BFP of a tensor blocked along its last dim: (int8 mantissa codes, same shape; uint8 shared
    exponents, [..., ceil(L / block_size)]).  bfp_unpack(*bfp_pack(x)) == bfp_qdq(x)."""
    xc = _prep(x, "bfp_pack")
    L = xc.shape[-1] if xc.dim() else 1
    rows = xc.numel() // max(L, 1)
    nblk = -(-L // block_size)
    mant = torch.empty(xc.shape, dtype=torch.int8, device=xc.device)
    exps = torch.empty(tuple(xc.shape[:-1]) + (nblk,), dtype=torch.uint8, device=xc.device)
    check(lib().dmxq_bfp_pack(ptr(xc), dtype_code(xc.dtype), ptr(mant), ptr(exps), rows, L, block_size, precision,
                              int(symmetric), stream_of(xc)), "dmxq_bfp_pack")
    return mant, exps


def bfp_unpack(mant, exps, precision: int, block_size: int, out_dtype: torch.dtype = torch.float32):
    require_gpu(mant, "bfp_unpack")
    m, e = mant.contiguous(), exps.contiguous()
    L = m.shape[-1] if m.dim() else 1
    rows = m.numel() // max(L, 1)
    out = torch.empty(m.shape, dtype=out_dtype, device=m.device)
    check(lib().dmxq_bfp_unpack(ptr(m), ptr(e), ptr(out), dtype_code(out_dtype), rows, L, block_size, precision,
                                stream_of(m)), "dmxq_bfp_unpack")
    return out


def weight_hypernet(w, precision: int, block_size: int, symmetric: bool = True, score=None, K: int = 0, M: int = 0,
                    sq_scale=None, out_dtype: Optional[torch.dtype] = None, block_dim: int = -1):
    """Fused N:M mask -> SmoothQuant scale -> BFP Q->DQ over a weight (one launch), everything along `block_dim`: the last dim
    (Linear) or any other (conv weights along in-channels: dmxq_weight_hypernet_strided).
    Returns None when the geometry / dtype combination is not fusable (the caller runs the unfused chain)."""
    wc = _prep(w, "weight_hypernet")
    outer, L, inner = split3(wc.shape, block_dim) if wc.dim() else (1, 1, 1)
    rows = wc.numel() // max(L, 1)
    sc = _prep(score, "weight_hypernet") if (score is not None and M) else None
    if sc is not None and sc.shape != wc.shape:
        return None
    t1 = torch.promote_types(wc.dtype, sc.dtype) if sc is not None else wc.dtype
    od = out_dtype or t1
    out = torch.empty(wc.shape, dtype=od, device=wc.device)
    sq = sq_scale.detach().to(device=wc.device, dtype=torch.float32).contiguous() if sq_scale is not None else None
    if sq is not None and sq.numel() != L:
        return None
    if inner != 1:
        rc = lib().dmxq_weight_hypernet_strided(ptr(wc), dtype_code(wc.dtype), ptr(sc), dtype_code(sc.dtype) if sc is not None else 0,
                                                K, M if sc is not None else 0, ptr(sq), ptr(out), dtype_code(od), outer, L, inner,
                                                block_size, precision, int(symmetric), stream_of(wc))
    else:
        rc = lib().dmxq_weight_hypernet(ptr(wc), dtype_code(wc.dtype), ptr(sc), dtype_code(sc.dtype) if sc is not None else 0,
                                        K, M if sc is not None else 0, ptr(sq), ptr(out), dtype_code(od), rows, L, block_size,
                                        precision, int(symmetric), stream_of(wc))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "dmxq_weight_hypernet")
    return out


def weight_hypernet_multi(ws, precision: int, block_size: int, symmetric: bool = True, scores=None, K: int = 0, M: int = 0,
                          sq_scales=None, out_dtype: Optional[torch.dtype] = None):
    """The weight chain of MANY Linear weights (one dtype, one device, one N:M pattern, one BFP format) in as few launches as
    possible (`dmxq_weight_hypernet_multi`): the same results as `[weight_hypernet(w, ...) for w in ws]`.  scores / sq_scales: one
    per weight, or None.  Returns None -- nothing launched -- when any of them is not fusable (the caller goes one by one)."""
    ws = list(ws)
    if not ws:
        return []
    wcs = [_prep(w, "weight_hypernet_multi") for w in ws]
    dt, dev = wcs[0].dtype, wcs[0].device
    if any(w.dtype != dt or w.device != dev for w in wcs):
        raise ValueError("weight_hypernet_multi: all weights must share one dtype and one device")
    masked = scores is not None and M != 0
    scs = [_prep(s, "weight_hypernet_multi") for s in scores] if masked else []
    sqs = [q.detach().to(device=dev, dtype=torch.float32).contiguous() for q in sq_scales] if sq_scales is not None else []
    if (masked and len(scs) != len(wcs)) or (sqs and len(sqs) != len(wcs)):
        raise ValueError("weight_hypernet_multi: one score / scale per weight")
    if masked and (any(s.dtype != scs[0].dtype for s in scs) or any(s.shape != w.shape for s, w in zip(scs, wcs))):
        return None
    od = out_dtype or (torch.promote_types(dt, scs[0].dtype) if masked else dt)
    outs = [torch.empty(w.shape, dtype=od, device=dev) for w in wcs]
    descs = (_lib.HypernetDesc * len(wcs))()
    for i, (d, w, o) in enumerate(zip(descs, wcs, outs)):
        L = w.shape[-1] if w.dim() else 1
        if sqs and sqs[i].numel() != L:
            return None
        d.w, d.score, d.sq_scale, d.out = w.data_ptr(), (scs[i].data_ptr() if masked else None), (sqs[i].data_ptr() if sqs else None), o.data_ptr()
        d.rows, d.L = w.numel() // max(L, 1), L
    with torch.cuda.device(dev):
        rc = lib().dmxq_weight_hypernet_multi(descs, len(wcs), dtype_code(dt), dtype_code(scs[0].dtype) if masked else 0, K,
                                              M if masked else 0, dtype_code(od), block_size, precision, int(symmetric), stream_of(wcs[0]))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "dmxq_weight_hypernet_multi")
    return outs


def input_hypernet(x, sq_scale, precision: int, block_size: int, symmetric: bool = True):
    """Fused SmoothQuant input scaling -> BFP input cast along the last dim (one launch): BFP_QDQ(x / sq_scale) in float32, the
    dtype `a / scale` has in the reference (smoothquant.py:255-268).  None when the geometry is not fusable."""
    xc = _prep(x, "input_hypernet")
    L = xc.shape[-1] if xc.dim() else 1
    rows = xc.numel() // max(L, 1)
    sq = sq_scale.detach().to(device=xc.device, dtype=torch.float32).contiguous()
    if sq.numel() != L:
        return None
    out = torch.empty(xc.shape, dtype=torch.float32, device=xc.device)
    rc = lib().dmxq_input_hypernet(ptr(xc), dtype_code(xc.dtype), ptr(sq), ptr(out), dtype_code(torch.float32), rows, L, block_size,
                                   precision, int(symmetric), stream_of(xc))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "dmxq_input_hypernet")
    return out


def binary_cast(a, b, op: str, cast_a=None, cast_b=None, cast_out=None, then_bfp=None):
    """A binary DmxModule in one launch: cast_out(cast_a(a) op cast_b(b)), op in {"add", "mul"}; each cast is a FloatingPoint
    format (nearest, signed) or None = SAME.  then_bfp = (precision, block_size): the consumer's BFP input cast along the last dim in the
    same launch (dmxq_binary_cast_bfp).  None when not fusable (the caller runs the casts and the op one by one)."""
    ac, bc = _prep(a, "binary_cast"), _prep(b, "binary_cast")
    if ac.shape != bc.shape or ac.dtype != bc.dtype or ac.device != bc.device:
        return None
    structs = []
    for f in (cast_a, cast_b, cast_out):
        if f is not None and (f.rounding != "nearest" or f.unsigned):
            return None
        structs.append(None if f is None else _lib.FloatFmt(int(f.mantissa), int(f.exponent), int(f.bias), int(bool(f.flush_subnormal))))
    out = torch.empty_like(ac)
    import ctypes
    ptrs = [ctypes.cast(ctypes.pointer(st), ctypes.c_void_p) if st is not None else None for st in structs]
    if then_bfp:
        if ac.dim() < 1 or ac.numel() == 0:
            return None
        rc = lib().dmxq_binary_cast_bfp(ptr(ac), ptr(bc), ptr(out), dtype_code(ac.dtype), ac.numel(), {"add": 0, "mul": 1}[op], *ptrs,
                                        ac.shape[-1], int(then_bfp[1]), int(then_bfp[0]), stream_of(ac))
    else:
        rc = lib().dmxq_binary_cast(ptr(ac), ptr(bc), ptr(out), dtype_code(ac.dtype), ac.numel(), {"add": 0, "mul": 1}[op], *ptrs, stream_of(ac))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "dmxq_binary_cast")
    return out


def sbfp_qdq(x, precision: int, block_size: int, scaler_man: int, scaler_exp: int, scaler_bias: int,
             scaler_flush: bool = True, clamp: bool = True, symmetric: bool = True, block_dim: int = -1,
             out_dtype: Optional[torch.dtype] = None):
    """ScaledBlockFloatingPoint Q->DQ (numerical/format.py:453-479), one launch."""
    xc = _prep(x, "sbfp_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    outer, L, inner = split3(xc.shape, block_dim) if xc.dim() else (1, 1, 1)
    check(lib().dmxq_sbfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, L, inner, block_size,
                              precision, int(clamp), int(symmetric), scaler_man, scaler_exp, scaler_bias,
                              int(scaler_flush), stream_of(xc)), "dmxq_sbfp_qdq")
    return out


def mxfp_qdq(x, man: int, exp: int, block_size: int, block_dim: int = -1, out_dtype: Optional[torch.dtype] = None):
    """MXFP Q->DQ (numerical/format.py:545-564), one launch."""
    xc = _prep(x, "mxfp_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    outer, L, inner = split3(xc.shape, block_dim) if xc.dim() else (1, 1, 1)
    check(lib().dmxq_mxfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, L, inner, block_size,
                              man, exp, stream_of(xc)), "dmxq_mxfp_qdq")
    return out


def float_qdq(x, man: int, exp: int, bias: int, flush_subnormal: bool, unsigned: bool = False,
              rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """Low-bit float Q->DQ (quant/quant_function.py:120-152 semantics), one launch."""
    xc = _prep(x, "float_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    seed = _next_seed() if (seed is None and rounding == "stochastic") else (seed or 0)
    check(lib().dmxq_float_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.numel(), man, exp,
                               bias, int(flush_subnormal), int(unsigned), _lib.ROUNDING_CODE[rounding], seed,
                               stream_of(xc)), "dmxq_float_qdq")
    return out


def fixed_qdq(x, precision: int, fraction: int, clamp: bool = True, symmetric: bool = True, rounding: str = "nearest",
              scale: Optional[torch.Tensor] = None, zero_point: Optional[torch.Tensor] = None,
              ch_axis: Optional[int] = None, group_size: Optional[int] = None,
              out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """Fixed-point Q->DQ with the affine wrapper of numerical/cast.py:278-296 fused in, one launch.
    scale None: bare FixedPoint.cast.  ch_axis None: per-tensor scale; else per-channel (group_size None) or
    per-group slabs of `group_size` channels."""
    xc = _prep(x, "fixed_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    sc = zp = None
    outer, C, inner, gs = 1, 1, xc.numel(), 1
    if scale is not None:
        sc = scale.detach().to(device=xc.device, dtype=torch.float32).contiguous()
        zp = zero_point.detach().to(device=xc.device, dtype=torch.int64).contiguous()
        if ch_axis is not None and xc.dim() > 0:
            outer, C, inner = split3(xc.shape, ch_axis)
            gs = group_size or 1
            need = -(-C // gs)
        else:
            need = 1
        if sc.numel() < need or zp.numel() < need:
            raise ValueError(f"fixed_qdq: need {need} scale/zero_point entries, got {sc.numel()}/{zp.numel()}")
    seed = _next_seed() if (seed is None and rounding == "stochastic") else (seed or 0)
    check(lib().dmxq_fixed_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, C, inner,
                               precision, fraction, int(clamp), int(symmetric), _lib.ROUNDING_CODE[rounding],
                               ptr(sc), ptr(zp), gs, seed, stream_of(xc)), "dmxq_fixed_qdq")
    return out


def fixed_qdq_multi(tensors, precision: int, fraction: int, clamp: bool, symmetric: bool, scales, zero_points,
                    group_size: Optional[int] = None, rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None,
                    seed: Optional[int] = None):
    """Affine integer Q->DQ (numerical/cast.py:278-296) of MANY weights of one dtype on one device in as few launches as
    possible (`dmxq_fixed_qdq_multi`): the same results as `[fixed_qdq(t, ..., scale=s, zero_point=z, ch_axis=0 (or None when
    s has one entry), group_size=group_size) for t, s, z in zip(tensors, scales, zero_points)]`."""
    xs = [_prep(t, "fixed_qdq_multi") for t in tensors]
    if not xs:
        return []
    dt, dev = xs[0].dtype, xs[0].device
    if any(x.dtype != dt or x.device != dev for x in xs):
        raise ValueError("fixed_qdq_multi: all tensors must share one dtype and one device")
    outs = [torch.empty(x.shape, dtype=out_dtype or dt, device=dev) for x in xs]
    scs = [s.detach().to(device=dev, dtype=torch.float32).contiguous() for s in scales]
    zps = [z.detach().to(device=dev, dtype=torch.int64).contiguous() for z in zero_points]
    gs = group_size or 1
    descs = (_lib.AffineDesc * len(xs))()
    for i, (d, x, o, s, z) in enumerate(zip(descs, xs, outs, scs, zps)):
        # a one-entry scale means per-tensor ONLY when no group_size was asked for; with group_size set a weight needs
        # ceil(C / group_size) entries like `fixed_qdq(ch_axis=0, group_size=...)` (an uncalibrated cast, scale = [1.0], must
        # raise here too, not be folded with a per-tensor scale of 1)
        outer, C, inner = split3(x.shape, 0) if x.dim() > 0 else (1, 1, 1)
        need = -(-C // gs) if (group_size or s.numel() != 1) else 1
        if s.numel() < need or z.numel() < need:
            raise ValueError(f"fixed_qdq_multi: tensor {i} needs {need} scale/zero_point entries, got {s.numel()}/{z.numel()}")
        if need == 1:
            outer, C, inner = 1, 1, x.numel()
        d.in_, d.out, d.scale, d.zero_point, d.outer, d.C, d.inner = x.data_ptr(), o.data_ptr(), s.data_ptr(), z.data_ptr(), outer, C, inner
    seed = _next_seed() if (seed is None and rounding == "stochastic") else (seed or 0)
    with torch.cuda.device(dev):
        check(lib().dmxq_fixed_qdq_multi(descs, len(xs), dtype_code(dt), dtype_code(outs[0].dtype), precision, fraction, int(clamp),
                                         int(symmetric), _lib.ROUNDING_CODE[rounding], gs, seed, stream_of(xs[0])), "dmxq_fixed_qdq_multi")
    return outs


def _nm(score, x, K, M, block_dim, want_mask, want_y, mask_dtype, y_dtype):
    sc = _prep(score, "nm_mask")
    if sc.dim() == 0 or sc.shape[block_dim] % M != 0:
        # sparse.py:166-168
        raise AssertionError(
            f"score has size {tuple(sc.shape)} at dimension {block_dim}, not a multiple of block size {M}")
    outer, L, inner = split3(sc.shape, block_dim)
    xc = None
    if want_y:
        xc = _prep(x, "nm_sparsify")
        if xc.shape != sc.shape:
            xc = xc.expand(sc.shape).contiguous()
    mask = torch.empty(sc.shape, dtype=mask_dtype or sc.dtype, device=sc.device) if want_mask else None
    y = torch.empty(sc.shape, dtype=y_dtype, device=sc.device) if want_y else None
    check(lib().dmxq_nm_mask(ptr(sc), dtype_code(sc.dtype), ptr(xc), dtype_code(xc.dtype) if want_y else 0,
                             ptr(mask), dtype_code(mask.dtype) if want_mask else 0,
                             ptr(y), dtype_code(y.dtype) if want_y else 0, outer, L, inner, K, M, stream_of(sc)),
          "dmxq_nm_mask")
    return mask, y


def _topk(score, x, density, want_mask, want_y, mask_dtype, y_dtype):
    sc = _prep(score, "topk_mask")
    n = sc.numel()
    n_zero = int(n * (1.0 - density))  # sparse.py:116
    xc = None
    if want_y:
        xc = _prep(x, "topk_sparsify")
        if xc.shape != sc.shape:
            xc = xc.expand(sc.shape).contiguous()
    mask = torch.empty(sc.shape, dtype=mask_dtype or sc.dtype, device=sc.device) if want_mask else None
    y = torch.empty(sc.shape, dtype=y_dtype, device=sc.device) if want_y else None
    ws = torch.empty(max(1, (lib().dmxq_topk_workspace_bytes(n) + 7) // 8), dtype=torch.int64, device=sc.device)
    check(lib().dmxq_topk_mask(ptr(sc), dtype_code(sc.dtype), ptr(xc), dtype_code(xc.dtype) if want_y else 0,
                               ptr(mask), dtype_code(mask.dtype) if want_mask else 0,
                               ptr(y), dtype_code(y.dtype) if want_y else 0, n, n_zero, ptr(ws), stream_of(sc)),
          "dmxq_topk_mask")
    return mask, y


def topk_mask(score, density: float, mask_dtype: Optional[torch.dtype] = None):
    """Global top-k mask (sparse.py:109-123): the int(n * (1 - density)) lowest scores are zeroed; float mask in the
    score's dtype.  No sort: a radix select + one masking pass (csrc/topk.hip)."""
    return _topk(score, None, density, True, False, mask_dtype, None)[0]


def topk_sparsify(x, score, density: float, return_mask: bool = False):
    """x * topk_mask(score) in the same final pass (sparse.py:300), torch's type promotion for the product."""
    mask, y = _topk(score, x, density, return_mask, True, None, torch.promote_types(x.dtype, score.dtype))
    return (y, mask) if return_mask else y


def bernoulli_mask(score, seed: Optional[int] = None, mask_dtype: Optional[torch.dtype] = None):
    """Bernoulli supermask (sparse.py:201-221): 1 with probability score."""
    sc = _prep(score, "bernoulli_mask")
    mask = torch.empty(sc.shape, dtype=mask_dtype or sc.dtype, device=sc.device)
    check(lib().dmxq_bernoulli_mask(ptr(sc), ptr(mask), dtype_code(sc.dtype), dtype_code(mask.dtype), sc.numel(),
                                    _next_seed() if seed is None else seed, stream_of(sc)), "dmxq_bernoulli_mask")
    return mask


def nm_mask(score, K: int, M: int, block_dim: int = -1, mask_dtype: Optional[torch.dtype] = None):
    """N:M mask (sparse.py:163-180): float mask in the score's dtype."""
    return _nm(score, None, K, M, block_dim, True, False, mask_dtype, None)[0]


def nm_sparsify(x, score, K: int, M: int, block_dim: int = -1, out_dtype: Optional[torch.dtype] = None,
                return_mask: bool = False):
    """Fused mask + apply: y = x * mask(score) (sparse.py:287-301); out dtype defaults to torch's promotion of
    (x.dtype, score.dtype), i.e. what `x * mask` yields in the reference."""
    yd = out_dtype or torch.promote_types(x.dtype, score.dtype)
    mask, y = _nm(score, x, K, M, block_dim, return_mask, True, None, yd)
    return (y, mask) if return_mask else y


def group_minmax(x, ch_axis: int, group_size: int):
    """Per-group (slabs of `group_size` channels along ch_axis) min and max: two float32 [G] tensors."""
    xc = _prep(x, "group_minmax")
    outer, C, inner = split3(xc.shape, ch_axis)
    G = -(-C // group_size)
    mn = torch.empty(G, dtype=torch.float32, device=xc.device)
    mx = torch.empty(G, dtype=torch.float32, device=xc.device)
    check(lib().dmxq_group_minmax(ptr(xc), dtype_code(xc.dtype), outer, C, inner, group_size, ptr(mn), ptr(mx),
                                  stream_of(xc)), "dmxq_group_minmax")
    return mn, mx


def group_minmax_accumulate(x, ch_axis: int, group_size: int, mn, mx):
    """running min / max of slabs of `group_size` channels updated IN PLACE by one launch (dmxq_group_minmax_accumulate)"""
    xc = _prep(x, "group_minmax_accumulate")
    outer, C, inner = split3(xc.shape, ch_axis)
    gs = max(int(group_size), 1)
    G = -(-C // gs)
    for t in (mn, mx):
        if not (t.is_cuda and t.device == xc.device and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == G):
            raise RuntimeError(f"group_minmax_accumulate: running min / max must be contiguous float32 tensors of {G} entries on the input's device")
    check(lib().dmxq_group_minmax_accumulate(ptr(xc), dtype_code(xc.dtype), outer, C, inner, gs, ptr(mn), ptr(mx), stream_of(xc)),
          "dmxq_group_minmax_accumulate")


def qparams(mn, mx, qmin: int, qmax: int, symmetric_qscheme: bool):
    """(min,max) -> (scale fp32, zero_point int64), numerical/observer.py:59-115."""
    require_gpu(mn, "qparams")
    mn = mn.to(torch.float32).contiguous()
    mx = mx.to(torch.float32).contiguous()
    scale = torch.empty_like(mn)
    zp = torch.empty(mn.shape, dtype=torch.int64, device=mn.device)
    check(lib().dmxq_qparams(ptr(mn), ptr(mx), mn.numel(), qmin, qmax, int(symmetric_qscheme), ptr(scale), ptr(zp),
                             stream_of(mn)), "dmxq_qparams")
    return scale, zp


def histc(x, bins: int, lo: float = 0.0, hi: float = 0.0):
    """torch.histc(x, bins, min=lo, max=hi) as the HistogramObserver uses it (numerical/observer.py:470-472,
    489-491): float32 [bins] counts.  lo == hi selects the data's own range, widened by one either side when the
    data is constant (torch.histc's convention)."""
    xc = _prep(x, "histc").reshape(-1)
    lo, hi = float(lo), float(hi)
    if lo == hi and xc.numel():
        mn, mx = group_minmax(xc.reshape(1, -1), 0, 1)
        lo, hi = float(mn), float(mx)
        if lo == hi:
            lo, hi = lo - 1.0, hi + 1.0
    out = torch.empty(int(bins), dtype=torch.float32, device=xc.device)
    if xc.numel() == 0 and not lo < hi:
        return out.zero_()
    check(lib().dmxq_histc(ptr(xc), dtype_code(xc.dtype), xc.numel(), int(bins), lo, hi, ptr(out), stream_of(xc)),
          "dmxq_histc")
    return out


def channel_maxabs(x, ch_axis: int):
    """max|x| per channel along ch_axis (numerical/smoothquant.py:285-299): float32 [C]."""
    xc = _prep(x, "channel_maxabs")
    outer, C, inner = split3(xc.shape, ch_axis)
    out = torch.empty(C, dtype=torch.float32, device=xc.device)
    check(lib().dmxq_channel_maxabs(ptr(xc), dtype_code(xc.dtype), outer, C, inner, ptr(out), stream_of(xc)),
          "dmxq_channel_maxabs")
    return out


def smoothquant_scale(a_maxabs, b_maxabs, alpha: float, scale_min: float = 1e-5):
    require_gpu(a_maxabs, "smoothquant_scale")
    a = a_maxabs.to(torch.float32).contiguous()
    b = b_maxabs.to(device=a.device, dtype=torch.float32).contiguous()
    out = torch.empty_like(a)
    check(lib().dmxq_smoothquant_scale(ptr(a), ptr(b), a.numel(), float(alpha), float(scale_min), ptr(out),
                                       stream_of(a)), "dmxq_smoothquant_scale")
    return out


def scale_channels(x, scale, ch_axis: int, divide: bool, out_dtype: Optional[torch.dtype] = None):
    xc = _prep(x, "scale_channels")
    outer, C, inner = split3(xc.shape, ch_axis)
    sc = scale.detach().to(device=xc.device, dtype=torch.float32).contiguous()
    if sc.numel() != C:
        raise ValueError(f"scale_channels: scale has {sc.numel()} entries for {C} channels")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_scale_channels(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, C, inner,
                                    ptr(sc), int(divide), stream_of(xc)), "dmxq_scale_channels")
    return out


def gelu(x, approximate: str = "none", out_dtype: Optional[torch.dtype] = None):
    xc = _prep(x, "gelu")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_gelu(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.numel(),
                          int(approximate == "tanh"), stream_of(xc)), "dmxq_gelu")
    return out


UNARY_GELU, UNARY_GELU_TANH, UNARY_SILU, UNARY_QUICK_GELU, UNARY_EXP, UNARY_SILU_EXPERIMENTAL = range(6)


def _unary(x, kind: int, param: float = 0.0, out_dtype: Optional[torch.dtype] = None):
    xc = _prep(x, "unary")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_unary(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.numel(), kind, float(param),
                           stream_of(xc)), "dmxq_unary")
    return out


def silu(x, out_dtype: Optional[torch.dtype] = None):
    """torch.nn.functional.silu (modeling/nn/torch_modules.py:1559-1576), exact function."""
    return _unary(x, UNARY_SILU, 0.0, out_dtype)


def quick_gelu(x, out_dtype: Optional[torch.dtype] = None):
    """transformers' QuickGELUActivation `x * sigmoid(1.702 * x)` in the input dtype (custom_modules.py:112-117)."""
    return _unary(x, UNARY_QUICK_GELU, 0.0, out_dtype)


def exp(x, out_dtype: Optional[torch.dtype] = None):
    """torch.exp (modeling/nn/torch_modules.py:236-242 Exp)."""
    return _unary(x, UNARY_EXP, 0.0, out_dtype)


def silu_experimental(x, scale: float):
    """the reference's `experimental.silu` (functional/functions.py:7-21): relu(x.to(float16)) * scale -> float16"""
    return _unary(x, UNARY_SILU_EXPERIMENTAL, scale, torch.float16)


def rope(x, cos, sin, unsqueeze_dim: int = 1):
    """APPLY_LLAMA_ROPE for ONE of q / k: (x * cos) + (rotate_half(x) * sin) in the tensor dtype (custom_modules.py:142-172);
    x [B, n1, n2, D], cos / sin [B, S, D].  Returns None when the HIP kernel does not take this shape / dtype mix (the
    caller keeps torch's own ops)."""
    xc = _prep(x, "rope")
    if xc.dim() != 4 or cos.dim() != 3 or sin.dim() != 3 or unsqueeze_dim not in (1, 2) or cos.dtype != xc.dtype or sin.dtype != xc.dtype:
        return None
    c, s = _prep(cos, "rope"), _prep(sin, "rope")
    B, n1, n2, D = xc.shape
    if tuple(c.shape) != (B, n2 if unsqueeze_dim == 1 else n1, D) or s.shape != c.shape:
        return None
    out = torch.empty_like(xc)
    rc = lib().dmxq_rope(ptr(xc), ptr(c), ptr(s), ptr(out), dtype_code(xc.dtype), B, n1, n2, D, int(unsqueeze_dim == 1), stream_of(xc))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "dmxq_rope")
    return out


def relu_cast(x, cast_in=None, cast_out=None, then_bfp=None):
    """A ReLU DmxModule in one launch: cast_out(relu(cast_in(x))); casts are FloatingPoint formats (nearest, signed) or None = SAME;
    then_bfp as binary_cast (dmxq_relu_cast_bfp).  None when not fusable."""
    import ctypes
    xc = _prep(x, "relu_cast")
    structs = []
    for f in (cast_in, cast_out):
        if f is not None and (f.rounding != "nearest" or f.unsigned):
            return None
        structs.append(None if f is None else _lib.FloatFmt(int(f.mantissa), int(f.exponent), int(f.bias), int(bool(f.flush_subnormal))))
    ptrs = [ctypes.cast(ctypes.pointer(st), ctypes.c_void_p) if st is not None else None for st in structs]
    out = torch.empty_like(xc)
    if then_bfp:
        if xc.dim() < 1 or xc.numel() == 0:
            return None
        rc = lib().dmxq_relu_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel(), *ptrs, xc.shape[-1], int(then_bfp[1]), int(then_bfp[0]),
                                      stream_of(xc))
    else:
        rc = lib().dmxq_relu_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel(), *ptrs, stream_of(xc))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "dmxq_relu_cast")
    return out


def rope_cast(x, cos, sin, unsqueeze_dim: int = 1, cast_x=None, cast_cos=None, cast_sin=None, cast_out=None):
    """One operand of an ApplyRotaryPosEmb module with its casts in one launch: cast_out(rope(cast_x(x), cast_cos(cos), cast_sin(sin)));
    casts are FloatingPoint formats (nearest, signed) or None = SAME.  None when not fusable."""
    import ctypes
    xc = _prep(x, "rope_cast")
    if xc.dim() != 4 or cos.dim() != 3 or sin.dim() != 3 or unsqueeze_dim not in (1, 2) or cos.dtype != xc.dtype or sin.dtype != xc.dtype:
        return None
    c, s = _prep(cos, "rope_cast"), _prep(sin, "rope_cast")
    B, n1, n2, D = xc.shape
    if tuple(c.shape) != (B, n2 if unsqueeze_dim == 1 else n1, D) or s.shape != c.shape:
        return None
    structs = []
    for f in (cast_x, cast_cos, cast_sin, cast_out):
        if f is not None and (f.rounding != "nearest" or f.unsigned):
            return None
        structs.append(None if f is None else _lib.FloatFmt(int(f.mantissa), int(f.exponent), int(f.bias), int(bool(f.flush_subnormal))))
    ptrs = [ctypes.cast(ctypes.pointer(st), ctypes.c_void_p) if st is not None else None for st in structs]
    out = torch.empty_like(xc)
    rc = lib().dmxq_rope_cast(ptr(xc), ptr(c), ptr(s), ptr(out), dtype_code(xc.dtype), B, n1, n2, D, int(unsqueeze_dim == 1), *ptrs, stream_of(xc))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "dmxq_rope_cast")
    return out


def softmax(x, dim: int = -1, input_clamp: Optional[float] = None, out_dtype: Optional[torch.dtype] = None):
    require_gpu(x, "softmax")
    d = dim % x.dim()
    xt = x if d == x.dim() - 1 else x.transpose(d, -1)
    xc = xt.contiguous()
    cols = xc.shape[-1]
    rows = xc.numel() // max(cols, 1)
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_softmax(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), rows, cols,
                             float(input_clamp) if input_clamp is not None else -math.inf, stream_of(xc)),
          "dmxq_softmax")
    return out if d == x.dim() - 1 else out.transpose(d, -1)


def layernorm(x, normalized_shape, weight=None, bias=None, eps: float = 1e-5,
              out_dtype: Optional[torch.dtype] = None):
    xc = _prep(x, "layernorm")
    cols = 1
    for s in (normalized_shape if not isinstance(normalized_shape, int) else (normalized_shape,)):
        cols *= s
    rows = xc.numel() // max(cols, 1)
    w = weight.detach().contiguous() if weight is not None else None
    b = bias.detach().contiguous() if bias is not None else None
    if w is not None and b is not None and w.dtype != b.dtype:
        b = b.to(w.dtype)
    wb_dtype = dtype_code((w if w is not None else b).dtype) if (w is not None or b is not None) else 0
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_layernorm(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), rows, cols, ptr(w),
                               ptr(b), wb_dtype, float(eps), stream_of(xc)), "dmxq_layernorm")
    return out


def rmsnorm(x, normalized_shape, weight=None, eps: Optional[float] = None, out_dtype: Optional[torch.dtype] = None):
    """torch.nn.functional.rms_norm over the trailing `normalized_shape` (modeling/nn/torch_modules.py:1144-1170);
    eps None = torch.finfo(x.dtype).eps, as torch."""
    xc = _prep(x, "rmsnorm")
    cols = 1
    for s_ in (normalized_shape if not isinstance(normalized_shape, int) else (normalized_shape,)):
        cols *= s_
    rows = xc.numel() // max(cols, 1)
    w = weight.detach().contiguous() if weight is not None else None
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    eps = torch.finfo(xc.dtype).eps if eps is None else eps
    check(lib().dmxq_rmsnorm(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), rows, cols, ptr(w),
                             dtype_code(w.dtype) if w is not None else 0, float(eps), stream_of(xc)), "dmxq_rmsnorm")
    return out


# ---- an activation / normalisation DmxModule in one launch (include/dmxq.h dmxq_unary_cast ...): cast_out(f(cast_in(x)))
_UNARY_KIND = {"gelu": 0, "gelu_tanh": 1, "silu": 2, "quick_gelu": 3, "exp": 4}


def _cast_ptrs(*fmts):
    """(pointers, keep-alive) for dmxq_float_fmt arguments, or None when a format is not one the fused kernels take"""
    import ctypes
    structs = []
    for f in fmts:
        if f is not None and (f.rounding != "nearest" or f.unsigned):
            return None
        structs.append(None if f is None else _lib.FloatFmt(int(f.mantissa), int(f.exponent), int(f.bias), int(bool(f.flush_subnormal))))
    return [ctypes.cast(ctypes.pointer(st), ctypes.c_void_p) if st is not None else None for st in structs], structs


def _fused_rc(rc, what, out):
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, what)
    return out


def unary_cast(x, func: str, cast_in=None, cast_out=None):
    """A GELU / SiLU / QuickGELU / Exp DmxModule in one launch; None when not fusable."""
    xc = _prep(x, "unary_cast")
    cp = _cast_ptrs(cast_in, cast_out)
    if cp is None:
        return None
    out = torch.empty_like(xc)
    rc = lib().dmxq_unary_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel(), _UNARY_KIND[func], 0.0, *cp[0], stream_of(xc))
    return _fused_rc(rc, "dmxq_unary_cast", out)


def unary_cast_table(like, func: str, cast_in=None, cast_out=None, param: float = 0.0):
    """The 65,536-entry table of a unary DmxModule on `like`'s 16-bit dtype and device (dmxq_unary_cast_table); None when a cast is
    not tabulable."""
    require_gpu(like, "unary_cast_table")
    cp = _cast_ptrs(cast_in, cast_out)
    if cp is None or like.dtype not in (torch.bfloat16, torch.float16):
        return None
    kind = 5 if func == "silu_experimental" else _UNARY_KIND[func]
    table = torch.empty(65536, dtype=torch.int16, device=like.device)
    rc = lib().dmxq_unary_cast_table(dtype_code(like.dtype), kind, float(param), *cp[0], ptr(table), stream_of(like))
    return _fused_rc(rc, "dmxq_unary_cast_table", table)


def lut16_apply(x, table):
    """out[i] = table[x[i] as a 16-bit pattern] (dmxq_lut16_apply); None when x is not a whole number of aligned 16-byte vectors."""
    xc = _prep(x, "lut16_apply")
    if xc.element_size() != 2 or table.numel() != 65536 or table.element_size() != 2 or table.device != xc.device or not table.is_contiguous():
        raise ValueError("lut16_apply: a 16-bit tensor and a 65536-entry 16-bit table on its device")
    out = torch.empty_like(xc)
    rc = lib().dmxq_lut16_apply(ptr(xc), ptr(out), xc.numel(), ptr(table), stream_of(xc))
    return _fused_rc(rc, "dmxq_lut16_apply", out)


def softmax_cast(x, dim: int = -1, cast_in=None, cast_out=None, input_clamp: Optional[float] = None, then_bfp=None):
    """A Softmax DmxModule in one launch (softmax over the LAST dim only); then_bfp = (precision, block_size): the consumer's BFP input
    cast applied to the result in the same launch; None when not fusable."""
    xc = _prep(x, "softmax_cast")
    cp = _cast_ptrs(cast_in, cast_out)
    if cp is None or xc.dim() == 0 or dim % xc.dim() != xc.dim() - 1:
        return None
    cols = xc.shape[-1]
    out = torch.empty_like(xc)
    clamp = float(input_clamp) if input_clamp is not None else -math.inf
    if then_bfp:
        rc = lib().dmxq_softmax_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel() // max(cols, 1), cols, clamp, *cp[0],
                                         int(then_bfp[1]), int(then_bfp[0]), stream_of(xc))
        return _fused_rc(rc, "dmxq_softmax_cast_bfp", out)
    rc = lib().dmxq_softmax_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel() // max(cols, 1), cols, clamp, *cp[0], stream_of(xc))
    return _fused_rc(rc, "dmxq_softmax_cast", out)


def _norm_cast(x, normalized_shape, weight, bias, eps, rms, cast_in, cast_out, what, then_bfp=None):
    xc = _prep(x, what)
    cp = _cast_ptrs(cast_in, cast_out)
    cols = 1
    for d in (normalized_shape if not isinstance(normalized_shape, int) else (normalized_shape,)):
        cols *= d
    w = weight.detach().contiguous() if weight is not None else None
    b = bias.detach().contiguous() if bias is not None else None
    if cp is None or any(t is not None and (t.dtype != xc.dtype or t.device != xc.device or t.numel() != cols) for t in (w, b)):
        return None
    out = torch.empty_like(xc)
    rows = xc.numel() // max(cols, 1)
    if then_bfp:
        bb, bp = int(then_bfp[1]), int(then_bfp[0])
        if rms:
            rc = lib().dmxq_rmsnorm_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), float(eps), *cp[0], bb, bp, stream_of(xc))
        else:
            rc = lib().dmxq_layernorm_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), ptr(b), float(eps), *cp[0], bb, bp, stream_of(xc))
        return _fused_rc(rc, "dmxq_rmsnorm_cast_bfp" if rms else "dmxq_layernorm_cast_bfp", out)
    if rms:
        rc = lib().dmxq_rmsnorm_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), float(eps), *cp[0], stream_of(xc))
    else:
        rc = lib().dmxq_layernorm_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), ptr(b), float(eps), *cp[0], stream_of(xc))
    return _fused_rc(rc, "dmxq_rmsnorm_cast" if rms else "dmxq_layernorm_cast", out)


def layernorm_cast(x, normalized_shape, weight=None, bias=None, eps: float = 1e-5, cast_in=None, cast_out=None, then_bfp=None):
    """A LayerNorm DmxModule in one launch (weight / bias in x's dtype); then_bfp = (precision, block_size): the consumers' BFP input
    cast in the same launch; None when not fusable."""
    return _norm_cast(x, normalized_shape, weight, bias, eps, False, cast_in, cast_out, "layernorm_cast", then_bfp)


def rmsnorm_cast(x, normalized_shape, weight=None, eps: Optional[float] = None, cast_in=None, cast_out=None, then_bfp=None):
    """An RMSNorm DmxModule in one launch (eps None = torch.finfo(x.dtype).eps, as torch); then_bfp as layernorm_cast; None when not fusable."""
    eps = torch.finfo(x.dtype).eps if eps is None else eps
    return _norm_cast(x, normalized_shape, weight, None, eps, True, cast_in, cast_out, "rmsnorm_cast", then_bfp)


def _on_tensor_device(fn):
    """HIP launches go to the CURRENT device: a tensor that lives on another GPU of this process (single-process
    multi-GPU, device_map pipeline splits) needs its device made current around the C-ABI call (include/dmxq.h
    conventions), exactly like torch's own kernels do with a device guard.  One integer compare when it already is."""

    @functools.wraps(fn)
    def guarded(x, *args, **kwargs):
        if isinstance(x, torch.Tensor) and x.is_cuda and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return fn(x, *args, **kwargs)
        return fn(x, *args, **kwargs)

    return guarded


for _name in __all__:
    if _name not in ("bfp_qdq_multi", "fixed_qdq_multi", "weight_hypernet_multi"):  # (take lists; switch device themselves)
        globals()[_name] = _on_tensor_device(globals()[_name])
del _name
