"""The ctypes BINDING of the C ABI (include/dmxq.h): the Python twin of csrc/torch_binding.cpp.

One function per dispatcher op of `torch.ops.dmxq.*`, with the SAME raw schema (integer codes for rounding modes and function
kinds, `int[4]` lists for FloatingPoint formats, optional tensors, `out_dtype`) and the same behaviour: make the tensors contiguous,
factor the shape as [outer, L, inner], allocate the outputs, make the tensor's device current, pass torch's current HIP stream,
call ONE extern "C" entry point, raise `NotImplementedError` where the library answers DMXQ_ERR_UNSUPPORTED (what
TORCH_CHECK_NOT_IMPLEMENTED raises in the C++ binding).  No argument spelling, no policy: that lives ONCE in `_front.py`, which
runs on either binding (`DMXQ_BINDING=torch|ctypes`, ops.py).  No autograd registration and no meta kernels here (the ctypes
binding is not traceable by torch.compile); needs no C++ compiler against the torch headers.
"""
import ctypes
import functools

import torch

from . import _lib
from ._lib import check, dtype_code, lib, ptr, require_gpu, split3, stream_of

_U64 = 0xFFFFFFFFFFFFFFFF


def _prep(x: torch.Tensor, what: str) -> torch.Tensor:
    require_gpu(x, what)
    dtype_code(x.dtype)
    return x if x.is_contiguous() else x.contiguous()


def _split(x, dim):
    return split3(x.shape, dim) if x.dim() else (1, 1, 1)


def _unsupported(what):
    raise NotImplementedError(what)


def _fmt_ptrs(*fmts):
    """int[4] lists (man_bits, exp_bits, exp_bias, flush_subnormal; empty = SAME) -> dmxq_float_fmt pointers (+ keep-alive)"""
    structs = [None if not f else _lib.FloatFmt(int(f[0]), int(f[1]), int(f[2]), int(f[3])) for f in fmts]
    return [ctypes.cast(ctypes.pointer(s), ctypes.c_void_p) if s is not None else None for s in structs], structs


def _guarded(fn):
    """HIP launches go to the CURRENT device: a tensor on another GPU of this process needs its device made current around the
    C-ABI call (the C++ binding's device guard).  One integer compare when it already is."""

    @functools.wraps(fn)
    def g(x, *a, **k):
        t = x[0] if isinstance(x, (list, tuple)) and x else x
        if isinstance(t, torch.Tensor) and t.is_cuda and t.device.index != torch.cuda.current_device():
            with torch.cuda.device(t.device):
                return fn(x, *a, **k)
        return fn(x, *a, **k)

    return g


# ------------------------------------------------------------------------------------------------ block formats
@_guarded
def bfp_qdq(x, precision, block_size, block_dim=-1, symmetric=True, rounding=2, out_dtype=None, seed=0):
    xc = _prep(x, "bfp_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    outer, L, inner = _split(xc, block_dim)
    check(lib().dmxq_bfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, L, inner, block_size, precision,
                             rounding, int(symmetric), seed & _U64, stream_of(xc)), "dmxq_bfp_qdq")
    return out


bfp_qdq_nograd = bfp_qdq


@_guarded
def block_quantize(a, wl, symmetric, rounding, seed=0):
    xc = _prep(a, "block_quantize")
    if xc.dim() != 2:
        raise RuntimeError("block_quantize: expects a [rows, L] view")
    out = torch.empty_like(xc)
    check(lib().dmxq_bfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.shape[0], xc.shape[1], 1,
                             max(xc.shape[1], 2), wl, rounding, 1 if symmetric else 2, seed & _U64, stream_of(xc)), "dmxq_bfp_qdq")
    return out


@_guarded
def bfp_qdq_multi(xs, precision, block_size, block_dim=-1, symmetric=True, rounding=2, out_dtype=None, seed=0):
    if not xs:
        return []
    ins = [_prep(t, "bfp_qdq_multi") for t in xs]
    dt, dev = ins[0].dtype, ins[0].device
    if any(x.dtype != dt or x.device != dev for x in ins):
        raise RuntimeError("bfp_qdq_multi: all tensors must share one dtype and one device")
    outs = [torch.empty(x.shape, dtype=out_dtype or dt, device=dev) for x in ins]
    descs = (_lib.TensorDesc * len(ins))()
    for d, x, o in zip(descs, ins, outs):
        d.in_, d.out = x.data_ptr(), o.data_ptr()
        d.outer, d.L, d.inner = _split(x, block_dim)
    check(lib().dmxq_bfp_qdq_multi(descs, len(ins), dtype_code(dt), dtype_code(outs[0].dtype), block_size, precision, rounding,
                                   int(symmetric), seed & _U64, stream_of(ins[0])), "dmxq_bfp_qdq_multi")
    return outs


@_guarded
def bfp_pack(x, precision, block_size, symmetric=True):
    xc = _prep(x, "bfp_pack")
    L = xc.shape[-1] if xc.dim() else 1
    rows = xc.numel() // max(L, 1)
    mant = torch.empty(xc.shape, dtype=torch.int8, device=xc.device)
    exps = torch.empty((tuple(xc.shape[:-1]) if xc.dim() else ()) + (-(-L // block_size),), dtype=torch.uint8, device=xc.device)
    check(lib().dmxq_bfp_pack(ptr(xc), dtype_code(xc.dtype), ptr(mant), ptr(exps), rows, L, block_size, precision, int(symmetric),
                              stream_of(xc)), "dmxq_bfp_pack")
    return mant, exps


@_guarded
def bfp_unpack(mant, exps, precision, block_size, out_dtype):
    require_gpu(mant, "bfp_unpack")
    m, e = mant.contiguous(), exps.contiguous()
    L = m.shape[-1] if m.dim() else 1
    out = torch.empty(m.shape, dtype=out_dtype, device=m.device)
    check(lib().dmxq_bfp_unpack(ptr(m), ptr(e), ptr(out), dtype_code(out_dtype), m.numel() // max(L, 1), L, block_size, precision,
                                stream_of(m)), "dmxq_bfp_unpack")
    return out


@_guarded
def weight_hypernet(w, precision, block_size, symmetric, score, K, M, sq_scale, out_dtype=None, block_dim=-1):
    wc = _prep(w, "weight_hypernet")
    masked = score is not None and M != 0
    sc = _prep(score, "weight_hypernet") if masked else None
    if masked and sc.shape != wc.shape:
        _unsupported("weight_hypernet: score and weight shapes differ")
    outer, L, inner = _split(wc, block_dim if wc.dim() else -1)
    rows = wc.numel() // max(L, 1) if L else 0
    sq = sq_scale.detach().to(device=wc.device, dtype=torch.float32).contiguous() if sq_scale is not None else None
    if sq is not None and sq.numel() != L:
        _unsupported("weight_hypernet: scale length differs from the channel count")
    od = out_dtype or (torch.promote_types(wc.dtype, sc.dtype) if masked else wc.dtype)
    out = torch.empty(wc.shape, dtype=od, device=wc.device)
    if inner != 1:
        check(lib().dmxq_weight_hypernet_strided(ptr(wc), dtype_code(wc.dtype), ptr(sc), dtype_code(sc.dtype) if masked else 0, K,
                                                 M if masked else 0, ptr(sq), ptr(out), dtype_code(od), outer, L, inner, block_size,
                                                 precision, int(symmetric), stream_of(wc)), "dmxq_weight_hypernet_strided")
        return out
    check(lib().dmxq_weight_hypernet(ptr(wc), dtype_code(wc.dtype), ptr(sc), dtype_code(sc.dtype) if masked else 0, K, M if masked else 0,
                                     ptr(sq), ptr(out), dtype_code(od), rows, L, block_size, precision, int(symmetric), stream_of(wc)),
          "dmxq_weight_hypernet")
    return out


@_guarded
def weight_hypernet_multi(ws, precision, block_size, symmetric, scores, K, M, sq_scales, out_dtype=None):
    if not ws:
        return []
    masked = bool(scores) and M != 0
    if (masked and len(scores) != len(ws)) or (sq_scales and len(sq_scales) != len(ws)):
        raise RuntimeError("weight_hypernet_multi: one score / scale per weight, or none")
    wcs = [_prep(w, "weight_hypernet_multi") for w in ws]
    dt, dev = wcs[0].dtype, wcs[0].device
    if any(w.dtype != dt or w.device != dev for w in wcs):
        raise RuntimeError("weight_hypernet_multi: all weights must share one dtype and one device")
    scs = [_prep(s, "weight_hypernet_multi") for s in scores] if masked else []
    if masked and any(s.dtype != scs[0].dtype for s in scs):
        raise RuntimeError("weight_hypernet_multi: all scores must share one dtype")
    if masked and any(s.shape != w.shape for s, w in zip(scs, wcs)):
        _unsupported("weight_hypernet_multi: score and weight shapes differ")
    sqs = [q.detach().to(device=dev, dtype=torch.float32).contiguous() for q in sq_scales] if sq_scales else []
    od = out_dtype or (torch.promote_types(dt, scs[0].dtype) if masked else dt)
    outs = [torch.empty(w.shape, dtype=od, device=dev) for w in wcs]
    descs = (_lib.HypernetDesc * len(wcs))()
    for i, (d, w, o) in enumerate(zip(descs, wcs, outs)):
        L = w.shape[-1] if w.dim() else 1
        if sqs and sqs[i].numel() != L:
            _unsupported("weight_hypernet_multi: scale length differs from the channel count")
        d.w, d.score, d.sq_scale, d.out = w.data_ptr(), (scs[i].data_ptr() if masked else None), (sqs[i].data_ptr() if sqs else None), o.data_ptr()
        d.rows, d.L = (w.numel() // L if L else 0), L
    check(lib().dmxq_weight_hypernet_multi(descs, len(wcs), dtype_code(dt), dtype_code(scs[0].dtype) if masked else 0, K, M if masked else 0,
                                           dtype_code(od), block_size, precision, int(symmetric), stream_of(wcs[0])), "dmxq_weight_hypernet_multi")
    return outs


@_guarded
def input_hypernet(x, sq_scale, precision, block_size, symmetric):
    xc = _prep(x, "input_hypernet")
    L = xc.shape[-1] if xc.dim() else 1
    sq = sq_scale.detach().to(device=xc.device, dtype=torch.float32).contiguous()
    if sq.numel() != L:
        _unsupported("input_hypernet: scale length differs from the channel count")
    out = torch.empty(xc.shape, dtype=torch.float32, device=xc.device)
    check(lib().dmxq_input_hypernet(ptr(xc), dtype_code(xc.dtype), ptr(sq), ptr(out), _lib.F32, xc.numel() // max(L, 1), L, block_size,
                                    precision, int(symmetric), stream_of(xc)), "dmxq_input_hypernet")
    return out


@_guarded
def binary_cast(a, b, op, cast_a, cast_b, cast_out, bfp_block=0, bfp_precision=0):
    ac, bc = _prep(a, "binary_cast"), _prep(b, "binary_cast")
    if ac.shape != bc.shape or ac.dtype != bc.dtype or ac.device != bc.device:
        _unsupported("binary_cast: operands of one shape, dtype and device")
    out = torch.empty_like(ac)
    ptrs, _keep = _fmt_ptrs(cast_a, cast_b, cast_out)
    if bfp_block > 0:
        if ac.dim() < 1 or ac.numel() == 0:
            _unsupported("binary_cast: the BFP epilogue needs a last dim")
        check(lib().dmxq_binary_cast_bfp(ptr(ac), ptr(bc), ptr(out), dtype_code(ac.dtype), ac.numel(), op, *ptrs, ac.shape[-1], bfp_block,
                                         bfp_precision, stream_of(ac)), "dmxq_binary_cast_bfp")
    else:
        check(lib().dmxq_binary_cast(ptr(ac), ptr(bc), ptr(out), dtype_code(ac.dtype), ac.numel(), op, *ptrs, stream_of(ac)), "dmxq_binary_cast")
    return out


@_guarded
def relu_cast(x, cast_in, cast_out, bfp_block=0, bfp_precision=0):
    xc = _prep(x, "relu_cast")
    out = torch.empty_like(xc)
    ptrs, _keep = _fmt_ptrs(cast_in, cast_out)
    if bfp_block > 0:
        if xc.dim() < 1 or xc.numel() == 0:
            _unsupported("relu_cast: the BFP epilogue needs a last dim")
        check(lib().dmxq_relu_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel(), *ptrs, xc.shape[-1], bfp_block, bfp_precision,
                                       stream_of(xc)), "dmxq_relu_cast_bfp")
    else:
        check(lib().dmxq_relu_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel(), *ptrs, stream_of(xc)), "dmxq_relu_cast")
    return out


@_guarded
def sbfp_qdq(x, precision, block_size, scaler_man, scaler_exp, scaler_bias, scaler_flush, clamp, symmetric, block_dim=-1, out_dtype=None):
    xc = _prep(x, "sbfp_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    outer, L, inner = _split(xc, block_dim)
    check(lib().dmxq_sbfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, L, inner, block_size, precision,
                              int(clamp), int(symmetric), scaler_man, scaler_exp, scaler_bias, int(scaler_flush), stream_of(xc)), "dmxq_sbfp_qdq")
    return out


sbfp_qdq_nograd = sbfp_qdq


@_guarded
def mxfp_qdq(x, man, exp, block_size, block_dim=-1, out_dtype=None):
    xc = _prep(x, "mxfp_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    outer, L, inner = _split(xc, block_dim)
    check(lib().dmxq_mxfp_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, L, inner, block_size, man, exp,
                              stream_of(xc)), "dmxq_mxfp_qdq")
    return out


mxfp_qdq_nograd = mxfp_qdq


# ------------------------------------------------------------------------------------------------ element formats
@_guarded
def float_qdq(x, man, exp, bias, flush_subnormal, unsigned_abs=False, rounding=2, out_dtype=None, seed=0):
    xc = _prep(x, "float_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_float_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.numel(), man, exp, bias,
                               int(flush_subnormal), int(unsigned_abs), rounding, seed & _U64, stream_of(xc)), "dmxq_float_qdq")
    return out


float_qdq_nograd = float_qdq


@_guarded
def float_qdq_multi(xs, man, exp, bias, flush_subnormal, unsigned_abs=False, rounding=2, out_dtype=None, seed=0):
    if not xs:
        return []
    ins = [_prep(t, "float_qdq_multi") for t in xs]
    dt, dev = ins[0].dtype, ins[0].device
    if any(x.dtype != dt or x.device != dev for x in ins):
        raise RuntimeError("float_qdq_multi: all tensors must share one dtype and one device")
    outs = [torch.empty(x.shape, dtype=out_dtype or dt, device=dev) for x in ins]
    descs = (_lib.TensorDesc * len(ins))()
    for d, x, o in zip(descs, ins, outs):
        d.in_, d.out, d.outer, d.L, d.inner = x.data_ptr(), o.data_ptr(), 1, x.numel(), 1
    check(lib().dmxq_float_qdq_multi(descs, len(ins), dtype_code(dt), dtype_code(outs[0].dtype), man, exp, bias, int(flush_subnormal),
                                     int(unsigned_abs), rounding, seed & _U64, stream_of(ins[0])), "dmxq_float_qdq_multi")
    return outs


def _affine_need(C, scale_numel, group_size, has_axis):
    return (-(-C // (group_size or 1))) if has_axis else 1


@_guarded
def fixed_qdq(x, precision, fraction, clamp, symmetric, rounding, scale, zero_point, ch_axis, group_size, out_dtype=None, seed=0):
    xc = _prep(x, "fixed_qdq")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    sc = zp = None
    outer, C, inner, gs = 1, 1, xc.numel(), 1
    if scale is not None:
        if zero_point is None:
            raise RuntimeError("fixed_qdq: scale without zero_point")
        sc = scale.detach().to(device=xc.device, dtype=torch.float32).contiguous()
        zp = zero_point.detach().to(device=xc.device, dtype=torch.int64).contiguous()
        need = 1
        if ch_axis is not None and xc.dim() > 0:
            outer, C, inner = split3(xc.shape, ch_axis)
            gs = group_size or 1
            need = -(-C // gs)
        if sc.numel() < need or zp.numel() < need:
            raise ValueError(f"fixed_qdq: need {need} scale/zero_point entries, got {sc.numel()}/{zp.numel()}")
    check(lib().dmxq_fixed_qdq(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, C, inner, precision, fraction,
                               int(clamp), int(symmetric), rounding, ptr(sc), ptr(zp), gs, seed & _U64, stream_of(xc)), "dmxq_fixed_qdq")
    return out


fixed_qdq_nograd = fixed_qdq


@_guarded
def fixed_qdq_multi(xs, precision, fraction, clamp, symmetric, rounding, scales, zero_points, group_size, out_dtype=None, seed=0):
    if not xs:
        return []
    ins = [_prep(t, "fixed_qdq_multi") for t in xs]
    dt, dev = ins[0].dtype, ins[0].device
    if any(x.dtype != dt or x.device != dev for x in ins):
        raise RuntimeError("fixed_qdq_multi: all tensors must share one dtype and one device")
    if len(scales) != len(ins) or len(zero_points) != len(ins):
        raise RuntimeError("fixed_qdq_multi: one scale and one zero_point tensor per weight")
    outs = [torch.empty(x.shape, dtype=out_dtype or dt, device=dev) for x in ins]
    scs = [s.detach().to(device=dev, dtype=torch.float32).contiguous() for s in scales]
    zps = [z.detach().to(device=dev, dtype=torch.int64).contiguous() for z in zero_points]
    gs = group_size or 1
    descs = (_lib.AffineDesc * len(ins))()
    for i, (d, x, o, s, z) in enumerate(zip(descs, ins, outs, scs, zps)):
        # a one-entry scale means per-tensor ONLY when no group_size was asked for; with group_size set a weight needs
        # ceil(C / group_size) entries like fixed_qdq(ch_axis=0, group_size=...) (an uncalibrated cast, scale = [1.0], must raise here
        # too, not be folded with a per-tensor scale of 1)
        outer, C, inner = split3(x.shape, 0) if x.dim() > 0 else (1, 1, 1)
        need = -(-C // gs) if (group_size or s.numel() != 1) else 1
        if s.numel() < need or z.numel() < need:
            raise ValueError(f"fixed_qdq_multi: tensor {i} needs {need} scale/zero_point entries, got {s.numel()}/{z.numel()}")
        if need == 1:
            outer, C, inner = 1, 1, x.numel()
        d.in_, d.out, d.scale, d.zero_point, d.outer, d.C, d.inner = x.data_ptr(), o.data_ptr(), s.data_ptr(), z.data_ptr(), outer, C, inner
    check(lib().dmxq_fixed_qdq_multi(descs, len(ins), dtype_code(dt), dtype_code(outs[0].dtype), precision, fraction, int(clamp),
                                     int(symmetric), rounding, gs, seed & _U64, stream_of(ins[0])), "dmxq_fixed_qdq_multi")
    return outs


@_guarded
def fixed_float_qdq_multi(xs, precision, fraction, clamp, symmetric, rounding, scales, zero_points, group_size,
                          fs, man, exp, bias, flush_subnormal, unsigned_abs, rounding_float, seed=0):
    """(fixed results, float results): `fixed_qdq_multi(xs, ...)` and `float_qdq_multi(fs, ...)` in one launch where the set allows it"""
    if not xs or not fs:
        return (fixed_qdq_multi(xs, precision, fraction, clamp, symmetric, rounding, scales, zero_points, group_size, None, seed),
                float_qdq_multi(fs, man, exp, bias, flush_subnormal, unsigned_abs, rounding_float, None, seed))
    ins = [_prep(t, "fixed_float_qdq_multi") for t in xs]
    fins = [_prep(t, "fixed_float_qdq_multi") for t in fs]
    dt, dev = ins[0].dtype, ins[0].device
    if any(x.dtype != dt or x.device != dev for x in ins + fins):
        raise RuntimeError("fixed_float_qdq_multi: all tensors must share one dtype and one device")
    if len(scales) != len(ins) or len(zero_points) != len(ins):
        raise RuntimeError("fixed_float_qdq_multi: one scale and one zero_point tensor per weight")
    outs = [torch.empty_like(x) for x in ins]
    fouts = [torch.empty_like(x) for x in fins]
    scs = [s.detach().to(device=dev, dtype=torch.float32).contiguous() for s in scales]
    zps = [z.detach().to(device=dev, dtype=torch.int64).contiguous() for z in zero_points]
    gs = group_size or 1
    descs = (_lib.AffineDesc * len(ins))()
    for i, (d, x, o, s, z) in enumerate(zip(descs, ins, outs, scs, zps)):
        outer, C, inner = split3(x.shape, 0) if x.dim() > 0 else (1, 1, 1)
        need = -(-C // gs) if (group_size or s.numel() != 1) else 1
        if s.numel() < need or z.numel() < need:
            raise ValueError(f"fixed_float_qdq_multi: tensor {i} needs {need} scale/zero_point entries, got {s.numel()}/{z.numel()}")
        if need == 1:
            outer, C, inner = 1, 1, x.numel()
        d.in_, d.out, d.scale, d.zero_point, d.outer, d.C, d.inner = x.data_ptr(), o.data_ptr(), s.data_ptr(), z.data_ptr(), outer, C, inner
    fdescs = (_lib.TensorDesc * len(fins))()
    for d, x, o in zip(fdescs, fins, fouts):
        d.in_, d.out, d.outer, d.L, d.inner = x.data_ptr(), o.data_ptr(), 1, x.numel(), 1
    check(lib().dmxq_fixed_float_qdq_multi(descs, len(ins), precision, fraction, int(clamp), int(symmetric), rounding, gs, fdescs, len(fins), man, exp,
                                           bias, int(flush_subnormal), int(unsigned_abs), rounding_float, dtype_code(dt), seed & _U64,
                                           stream_of(ins[0])), "dmxq_fixed_float_qdq_multi")
    return outs, fouts


# ------------------------------------------------------------------------------------------------ sparsity
def _maybe(t, want, shape, dtype, device):
    return torch.empty(shape, dtype=dtype, device=device) if want else None


@_guarded
def nm_mask(score, x, K, M, block_dim, want_mask, want_y, mask_dtype=None, y_dtype=None):
    sc = _prep(score, "nm_mask")
    outer, L, inner = _split(sc, block_dim)
    xc = None
    if want_y:
        xc = _prep(x, "nm_sparsify")
        if xc.shape != sc.shape:
            xc = xc.expand(sc.shape).contiguous()
        y_dtype = y_dtype or torch.promote_types(xc.dtype, sc.dtype)
    mask = _maybe(None, want_mask, sc.shape, mask_dtype or sc.dtype, sc.device)
    y = _maybe(None, want_y, sc.shape, y_dtype, sc.device)
    check(lib().dmxq_nm_mask(ptr(sc), dtype_code(sc.dtype), ptr(xc), dtype_code(xc.dtype) if want_y else 0, ptr(mask),
                             dtype_code(mask.dtype) if want_mask else 0, ptr(y), dtype_code(y.dtype) if want_y else 0, outer, L, inner, K, M,
                             stream_of(sc)), "dmxq_nm_mask")
    return mask, y


@_guarded
def topk_mask(score, x, n_zero, want_mask, want_y, mask_dtype=None, y_dtype=None):
    sc = _prep(score, "topk_mask")
    n = sc.numel()
    xc = None
    if want_y:
        xc = _prep(x, "topk_sparsify")
        if xc.shape != sc.shape:
            xc = xc.expand(sc.shape).contiguous()
        y_dtype = y_dtype or torch.promote_types(xc.dtype, sc.dtype)
    mask = _maybe(None, want_mask, sc.shape, mask_dtype or sc.dtype, sc.device)
    y = _maybe(None, want_y, sc.shape, y_dtype, sc.device)
    ws = torch.empty(max(1, (lib().dmxq_topk_workspace_bytes(n) + 7) // 8), dtype=torch.int64, device=sc.device)
    check(lib().dmxq_topk_mask(ptr(sc), dtype_code(sc.dtype), ptr(xc), dtype_code(xc.dtype) if want_y else 0, ptr(mask),
                               dtype_code(mask.dtype) if want_mask else 0, ptr(y), dtype_code(y.dtype) if want_y else 0, n, n_zero, ptr(ws),
                               stream_of(sc)), "dmxq_topk_mask")
    return mask, y


@_guarded
def bernoulli_mask(score, seed, mask_dtype=None):
    sc = _prep(score, "bernoulli_mask")
    mask = torch.empty(sc.shape, dtype=mask_dtype or sc.dtype, device=sc.device)
    check(lib().dmxq_bernoulli_mask(ptr(sc), ptr(mask), dtype_code(sc.dtype), dtype_code(mask.dtype), sc.numel(), seed & _U64,
                                    stream_of(sc)), "dmxq_bernoulli_mask")
    return mask


# ------------------------------------------------------------------------------------------------ calibration
@_guarded
def group_minmax(x, ch_axis, group_size):
    xc = _prep(x, "group_minmax")
    outer, C, inner = _split(xc, ch_axis)
    G = -(-C // group_size)
    mn = torch.empty(G, dtype=torch.float32, device=xc.device)
    mx = torch.empty(G, dtype=torch.float32, device=xc.device)
    check(lib().dmxq_group_minmax(ptr(xc), dtype_code(xc.dtype), outer, C, inner, group_size, ptr(mn), ptr(mx), stream_of(xc)),
          "dmxq_group_minmax")
    return mn, mx


@_guarded
def group_minmax_accumulate(x, ch_axis, group_size, mn, mx):
    xc = _prep(x, "group_minmax_accumulate")
    outer, C, inner = _split(xc, ch_axis)
    gs = max(int(group_size), 1)
    G = -(-C // gs)
    for t in (mn, mx):
        if not (t.is_cuda and t.device == xc.device and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == G):
            raise RuntimeError(f"group_minmax_accumulate: running min / max must be contiguous float32 tensors of {G} entries on the input's device")
    check(lib().dmxq_group_minmax_accumulate(ptr(xc), dtype_code(xc.dtype), outer, C, inner, gs, ptr(mn), ptr(mx), stream_of(xc)),
          "dmxq_group_minmax_accumulate")


@_guarded
def qparams(mn, mx, qmin, qmax, symmetric_qscheme):
    require_gpu(mn, "qparams")
    a = mn.to(torch.float32).contiguous()
    b = mx.to(device=a.device, dtype=torch.float32).contiguous()
    scale = torch.empty_like(a)
    zp = torch.empty(a.shape, dtype=torch.int64, device=a.device)
    check(lib().dmxq_qparams(ptr(a), ptr(b), a.numel(), qmin, qmax, int(symmetric_qscheme), ptr(scale), ptr(zp), stream_of(a)), "dmxq_qparams")
    return scale, zp


@_guarded
def histc(x, bins, lo, hi):
    xc = _prep(x, "histc").reshape(-1)
    out = torch.empty(int(bins), dtype=torch.float32, device=xc.device)
    check(lib().dmxq_histc(ptr(xc), dtype_code(xc.dtype), xc.numel(), int(bins), float(lo), float(hi), ptr(out), stream_of(xc)), "dmxq_histc")
    return out


@_guarded
def channel_maxabs(x, ch_axis):
    xc = _prep(x, "channel_maxabs")
    outer, C, inner = _split(xc, ch_axis)
    out = torch.empty(C, dtype=torch.float32, device=xc.device)
    check(lib().dmxq_channel_maxabs(ptr(xc), dtype_code(xc.dtype), outer, C, inner, ptr(out), stream_of(xc)), "dmxq_channel_maxabs")
    return out


@_guarded
def smoothquant_scale(a_maxabs, b_maxabs, alpha, scale_min):
    require_gpu(a_maxabs, "smoothquant_scale")
    a = a_maxabs.to(torch.float32).contiguous()
    b = b_maxabs.to(device=a.device, dtype=torch.float32).contiguous()
    if b.numel() != a.numel():
        raise RuntimeError("smoothquant_scale: the two maxima must have one entry per channel")
    out = torch.empty_like(a)
    check(lib().dmxq_smoothquant_scale(ptr(a), ptr(b), a.numel(), float(alpha), float(scale_min), ptr(out), stream_of(a)), "dmxq_smoothquant_scale")
    return out


@_guarded
def scale_channels(x, scale, ch_axis, divide, out_dtype=None):
    xc = _prep(x, "scale_channels")
    outer, C, inner = _split(xc, ch_axis)
    sc = scale.detach().to(device=xc.device, dtype=torch.float32).contiguous()
    if sc.numel() != C:
        raise ValueError(f"scale_channels: scale has {sc.numel()} entries for {C} channels")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_scale_channels(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), outer, C, inner, ptr(sc), int(divide),
                                    stream_of(xc)), "dmxq_scale_channels")
    return out


# ------------------------------------------------------------------------------------------------ approximator slot
@_guarded
def unary(x, kind, param=0.0, out_dtype=None):
    xc = _prep(x, "unary")
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_unary(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.numel(), kind, float(param), stream_of(xc)),
          "dmxq_unary")
    return out


def _rope_args(x, cos, sin, unsqueeze_dim, what):
    xc = _prep(x, what)
    if xc.dim() != 4 or cos.dim() != 3 or sin.dim() != 3 or unsqueeze_dim not in (1, 2) or cos.dtype != xc.dtype or sin.dtype != xc.dtype:
        _unsupported(f"{what}: x [B, n1, n2, D] and cos / sin [B, S, D] of one dtype")
    c, s = _prep(cos, what), _prep(sin, what)
    B, n1, n2, D = xc.shape
    if tuple(c.shape) != (B, n2 if unsqueeze_dim == 1 else n1, D) or s.shape != c.shape:
        _unsupported(f"{what}: cos / sin shape does not match x")
    return xc, c, s, (B, n1, n2, D)


@_guarded
def rope(x, cos, sin, unsqueeze_dim=1):
    xc, c, s, (B, n1, n2, D) = _rope_args(x, cos, sin, unsqueeze_dim, "rope")
    out = torch.empty_like(xc)
    check(lib().dmxq_rope(ptr(xc), ptr(c), ptr(s), ptr(out), dtype_code(xc.dtype), B, n1, n2, D, int(unsqueeze_dim == 1), stream_of(xc)), "dmxq_rope")
    return out


@_guarded
def rope_cast(x, cos, sin, unsqueeze_dim, cast_x, cast_cos, cast_sin, cast_out):
    xc, c, s, (B, n1, n2, D) = _rope_args(x, cos, sin, unsqueeze_dim, "rope_cast")
    ptrs, _keep = _fmt_ptrs(cast_x, cast_cos, cast_sin, cast_out)
    out = torch.empty_like(xc)
    check(lib().dmxq_rope_cast(ptr(xc), ptr(c), ptr(s), ptr(out), dtype_code(xc.dtype), B, n1, n2, D, int(unsqueeze_dim == 1), *ptrs,
                               stream_of(xc)), "dmxq_rope_cast")
    return out


@_guarded
def softmax(x, clamp_min, out_dtype=None):   # over the contiguous last dim
    xc = _prep(x, "softmax")
    cols = xc.shape[-1] if xc.dim() else 1
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    check(lib().dmxq_softmax(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), xc.numel() // max(cols, 1), cols, float(clamp_min),
                             stream_of(xc)), "dmxq_softmax")
    return out


def _wb(x, weight, bias, what, same_dtype):
    w = weight.detach().contiguous() if weight is not None else None
    b = bias.detach().contiguous() if bias is not None else None
    if same_dtype:
        for t in (w, b):
            if t is not None and (t.dtype != x.dtype or t.device != x.device):
                _unsupported(f"{what}: weight / bias must be in the row dtype, on the row's device")
    elif w is not None and b is not None and w.dtype != b.dtype:
        b = b.to(w.dtype)
    return w, b


@_guarded
def norm(x, cols, weight, bias, eps, kind, out_dtype=None):   # kind 0: layer_norm, 1: rms_norm (no bias)
    xc = _prep(x, "norm")
    rows = xc.numel() // max(cols, 1)
    w, b = _wb(xc, weight, bias, "norm", False)
    out = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
    if kind == 1:
        check(lib().dmxq_rmsnorm(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), rows, cols, ptr(w),
                                 dtype_code(w.dtype) if w is not None else 0, float(eps), stream_of(xc)), "dmxq_rmsnorm")
    else:
        wb = dtype_code((w if w is not None else b).dtype) if (w is not None or b is not None) else 0
        check(lib().dmxq_layernorm(ptr(xc), ptr(out), dtype_code(xc.dtype), dtype_code(out.dtype), rows, cols, ptr(w), ptr(b), wb, float(eps),
                                   stream_of(xc)), "dmxq_layernorm")
    return out


@_guarded
def unary_cast(x, kind, param, cast_in, cast_out):
    xc = _prep(x, "unary_cast")
    ptrs, _keep = _fmt_ptrs(cast_in, cast_out)
    out = torch.empty_like(xc)
    check(lib().dmxq_unary_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), xc.numel(), kind, float(param), *ptrs, stream_of(xc)), "dmxq_unary_cast")
    return out


@_guarded
def unary_cast_table(like, kind, param, cast_in, cast_out):
    require_gpu(like, "unary_cast_table")
    if like.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("unary_cast_table: a bfloat16 / float16 GPU tensor names the dtype and the device")
    ptrs, _keep = _fmt_ptrs(cast_in, cast_out)
    table = torch.empty(65536, dtype=torch.int16, device=like.device)
    check(lib().dmxq_unary_cast_table(dtype_code(like.dtype), kind, float(param), *ptrs, ptr(table), stream_of(like)), "dmxq_unary_cast_table")
    return table


@_guarded
def lut16_apply(x, table):
    xc = _prep(x, "lut16_apply")
    if xc.element_size() != 2 or table.numel() != 65536 or table.element_size() != 2 or table.device != xc.device or not table.is_contiguous():
        raise RuntimeError("lut16_apply: a 16-bit tensor and a 65536-entry 16-bit table on its device")
    out = torch.empty_like(xc)
    check(lib().dmxq_lut16_apply(ptr(xc), ptr(out), xc.numel(), ptr(table), stream_of(xc)), "dmxq_lut16_apply")
    return out


@_guarded
def softmax_cast(x, clamp_min, cast_in, cast_out, bfp_block=0, bfp_precision=0):
    xc = _prep(x, "softmax_cast")
    cols = xc.shape[-1] if xc.dim() else 1
    rows = xc.numel() // max(cols, 1)
    ptrs, _keep = _fmt_ptrs(cast_in, cast_out)
    out = torch.empty_like(xc)
    if bfp_block > 0:
        check(lib().dmxq_softmax_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, float(clamp_min), *ptrs, bfp_block, bfp_precision,
                                          stream_of(xc)), "dmxq_softmax_cast_bfp")
    else:
        check(lib().dmxq_softmax_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, float(clamp_min), *ptrs, stream_of(xc)), "dmxq_softmax_cast")
    return out


@_guarded
def norm_cast(x, cols, weight, bias, eps, kind, cast_in, cast_out, bfp_block=0, bfp_precision=0):
    xc = _prep(x, "norm_cast")
    rows = xc.numel() // max(cols, 1)
    w, b = _wb(xc, weight, bias, "norm_cast", True)
    for t in (w, b):
        if t is not None and t.numel() != cols:
            _unsupported("norm_cast: one weight / bias entry per column")
    ptrs, _keep = _fmt_ptrs(cast_in, cast_out)
    out = torch.empty_like(xc)
    L = lib()
    if kind == 1:
        if bfp_block > 0:
            check(L.dmxq_rmsnorm_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), float(eps), *ptrs, bfp_block, bfp_precision,
                                          stream_of(xc)), "dmxq_rmsnorm_cast_bfp")
        else:
            check(L.dmxq_rmsnorm_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), float(eps), *ptrs, stream_of(xc)), "dmxq_rmsnorm_cast")
    elif bfp_block > 0:
        check(L.dmxq_layernorm_cast_bfp(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), ptr(b), float(eps), *ptrs, bfp_block,
                                        bfp_precision, stream_of(xc)), "dmxq_layernorm_cast_bfp")
    else:
        check(L.dmxq_layernorm_cast(ptr(xc), ptr(out), dtype_code(xc.dtype), rows, cols, ptr(w), ptr(b), float(eps), *ptrs, stream_of(xc)),
              "dmxq_layernorm_cast")
    return out
