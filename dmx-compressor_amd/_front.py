"""The tensor-level front ends of the library, written ONCE for both bindings of the C ABI (include/dmxq.h).

Every function here is the Python spelling of one (or two) dispatcher ops -- rounding names, torch dtypes, Format objects, the
`DmxqError` for non-GPU tensors, seeds for stochastic rounding, "None when not fusable" -- over a RAW namespace `_ops` whose
functions carry the schema of `torch.ops.dmxq.*` (csrc/torch_binding.cpp):
  * `_backend_torch.RAW`   the TORCH_LIBRARY extension `lib/dmxq_torch.so` (meta kernels, registered straight-through backward);
  * `_backend_ctypes`      the same schema in Python over plain ctypes (no compiler against the torch headers needed).
`ops.py` binds one of them (`DMXQ_BINDING=torch|ctypes`).  A new entry point is therefore written in the C ABI, in the two bindings
(which only allocate and call) and HERE -- not, as until round 3, as two complete front ends.

No CPU path: a non-GPU tensor or a missing library raises `DmxqError`.
"""
import math
from typing import Optional

import torch

from ._lib import DmxqError, ROUNDING_CODE, require_gpu

__all__ = [
    "bfp_qdq", "block_quantize", "bfp_qdq_multi", "bfp_pack", "bfp_unpack", "weight_hypernet", "weight_hypernet_multi", "input_hypernet", "binary_cast", "rope_cast", "relu_cast", "unary_cast", "unary_cast_table", "lut16_apply", "softmax_cast", "layernorm_cast", "rmsnorm_cast", "sbfp_qdq", "mxfp_qdq", "float_qdq", "float_qdq_multi", "fixed_qdq", "fixed_qdq_multi", "fixed_float_qdq_multi", "nm_mask", "nm_sparsify", "topk_mask", "topk_sparsify", "bernoulli_mask", "group_minmax", "group_minmax_accumulate", "qparams", "channel_maxabs",
    "smoothquant_scale", "scale_channels", "gelu", "silu", "quick_gelu", "exp", "silu_experimental", "rope", "softmax", "layernorm",
    "rmsnorm", "histc",
]

_ops = None   # the raw namespace: set by bind()


def bind(raw) -> None:
    """ops.py hands over the binding's raw namespace (torch.ops.dmxq overloads, or the _backend_ctypes module)"""
    global _ops
    _ops = raw


_SEED_COUNTER = [0x5EED]


def _next_seed() -> int:
    """Stochastic rounding draws from a counter-based stream keyed by (seed, element index); a fresh seed per call,
    derived from `torch.initial_seed()` (the DEFAULT generator's seed: `torch.manual_seed` makes runs reproducible; a
    non-default `torch.Generator` is not consulted -- pass `seed=` explicitly for that).

    The seed is a kernel ARGUMENT, so a hipGraph capture would freeze it and every replay would repeat the same draws
    (accumulated rounding would no longer be unbiased): implicit seeding is refused while the current stream is
    capturing.  An explicit `seed=` is the caller's statement that frozen draws are intended."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise DmxqError("stochastic rounding with an implicit seed inside a hipGraph capture: the seed would be frozen "
                        "into the graph and every replay would reuse the same random draws; pass seed= explicitly or "
                        "keep stochastic casts outside the captured region")
    _SEED_COUNTER[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _SEED_COUNTER[0]) & 0xFFFFFFFFFFFFFFFF


def _seed_arg(seed, rounding):
    s = _next_seed() if (seed is None and rounding == "stochastic") else (seed or 0)
    return s - (1 << 64) if s >= (1 << 63) else s  # the schema's int is int64: same 64 bits


# ---------------------------------------------------------------------------------------------------- block formats
def bfp_qdq(x, precision: int, block_size: int, block_dim: int = -1, symmetric: bool = True,
            rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """BlockFloatingPoint Q->DQ of a whole tensor in one launch (numerical/format.py:304-343 semantics).
    Blocks run along `block_dim`; output has x's shape, contiguous, dtype `out_dtype` (default x.dtype)."""
    require_gpu(x, "bfp_qdq")
    op = _ops.bfp_qdq if (x.requires_grad and torch.is_grad_enabled()) else _ops.bfp_qdq_nograd   # (the STE wrapper costs ~2.5 us per call)
    return op(x, precision, block_size, block_dim, symmetric, ROUNDING_CODE[rounding], out_dtype, _seed_arg(seed, rounding))


def block_quantize(a2d, wl: int, symmetric: bool = True, rounding: str = "nearest", seed: Optional[int] = None):
    """The pybind seam on a [rows, L] float32 view, one block per row (quant_cpu.cpp:299-311).  symmetric False is the
    NATIVE asymmetric branch (:247-253), which is not the asymmetric FORMAT of bfp_qdq(symmetric=False)."""
    require_gpu(a2d, "block_quantize")
    return _ops.block_quantize(a2d, wl, symmetric, ROUNDING_CODE[rounding], _seed_arg(seed, rounding))


def bfp_qdq_multi(tensors, precision: int, block_size: int, block_dim: int = -1, symmetric: bool = True,
                  rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """BFP Q->DQ of MANY tensors of one dtype on one device in as few launches as possible (`dmxq_bfp_qdq_multi`): the
    same results as `[bfp_qdq(t, ...) for t in tensors]`."""
    tensors = list(tensors)
    if not tensors:
        return []   # (an empty Tensor[] carries no dispatch key)
    for t in tensors:
        require_gpu(t, "bfp_qdq_multi")
    if tensors and any(t.dtype != tensors[0].dtype or t.device != tensors[0].device for t in tensors):
        raise ValueError("bfp_qdq_multi: all tensors must share one dtype and one device")
    return list(_ops.bfp_qdq_multi(tensors, precision, block_size, block_dim, symmetric, ROUNDING_CODE[rounding], out_dtype,
                                   _seed_arg(seed, rounding)))


def bfp_pack(x, precision: int, block_size: int, symmetric: bool = True):
    """Packed on-wire BFP of a tensor blocked along its last dim: (int8 mantissa codes, same shape; uint8 shared
    exponents, [..., ceil(L / block_size)]).  bfp_unpack(*bfp_pack(x)) == bfp_qdq(x)."""
    require_gpu(x, "bfp_pack")
    return _ops.bfp_pack(x, precision, block_size, symmetric)


def bfp_unpack(mant, exps, precision: int, block_size: int, out_dtype: torch.dtype = torch.float32):
    require_gpu(mant, "bfp_unpack")
    return _ops.bfp_unpack(mant, exps, precision, block_size, out_dtype)


def weight_hypernet(w, precision: int, block_size: int, symmetric: bool = True, score=None, K: int = 0, M: int = 0,
                    sq_scale=None, out_dtype: Optional[torch.dtype] = None, block_dim: int = -1):
    """Fused N:M mask -> SmoothQuant scale -> BFP Q->DQ over a weight (one launch), everything along `block_dim`: the last dim
    (Linear) or any other (conv weights along in-channels: dmxq_weight_hypernet_strided).
    Returns None when the geometry / dtype combination is not fusable (the caller runs the unfused chain)."""
    require_gpu(w, "weight_hypernet")
    try:
        return _ops.weight_hypernet(w, precision, block_size, symmetric, score if M else None, K, M if score is not None else 0,
                                    sq_scale, out_dtype, block_dim)
    except NotImplementedError:
        return None


def weight_hypernet_multi(ws, precision: int, block_size: int, symmetric: bool = True, scores=None, K: int = 0, M: int = 0,
                          sq_scales=None, out_dtype: Optional[torch.dtype] = None):
    """The weight chain of MANY Linear weights (one dtype, one device, one N:M pattern, one BFP format) in as few launches as
    possible (`dmxq_weight_hypernet_multi`): the same results as `[weight_hypernet(w, ...) for w in ws]`.  scores / sq_scales: one
    per weight, or None.  Returns None -- nothing launched -- when any of them is not fusable (the caller goes one by one)."""
    ws = list(ws)
    if not ws:
        return []   # (an empty Tensor[] carries no dispatch key)
    for w in ws:
        require_gpu(w, "weight_hypernet_multi")
    try:
        return list(_ops.weight_hypernet_multi(ws, precision, block_size, symmetric, list(scores) if (scores is not None and M) else [],
                                               K, M if scores is not None else 0, list(sq_scales) if sq_scales is not None else [], out_dtype))
    except NotImplementedError:
        return None


def input_hypernet(x, sq_scale, precision: int, block_size: int, symmetric: bool = True):
    """Fused SmoothQuant input scaling -> BFP input cast along the last dim (one launch): BFP_QDQ(x / sq_scale) in float32, the
    dtype `a / scale` has in the reference (smoothquant.py:255-268).  None when the geometry is not fusable."""
    require_gpu(x, "input_hypernet")
    try:
        return _ops.input_hypernet(x, sq_scale, precision, block_size, symmetric)
    except NotImplementedError:
        return None


def _fmt4(f):
    return [] if f is None else [int(f.mantissa), int(f.exponent), int(f.bias), int(bool(f.flush_subnormal))]


def binary_cast(a, b, op: str, cast_a=None, cast_b=None, cast_out=None, then_bfp=None):
    """A binary DmxModule in one launch: cast_out(cast_a(a) op cast_b(b)), op in {"add", "mul"}; each cast is a FloatingPoint
    format (nearest, signed) or None = SAME.  then_bfp = (precision, block_size): the consumer's BFP input cast along the last dim in the
    same launch (dmxq_binary_cast_bfp).  None when not fusable (the caller runs the casts and the op one by one)."""
    require_gpu(a, "binary_cast")
    for f in (cast_a, cast_b, cast_out):
        if f is not None and (f.rounding != "nearest" or f.unsigned):
            return None
    try:
        return _ops.binary_cast(a, b, {"add": 0, "mul": 1}[op], _fmt4(cast_a), _fmt4(cast_b), _fmt4(cast_out),
                                int(then_bfp[1]) if then_bfp else 0, int(then_bfp[0]) if then_bfp else 0)
    except NotImplementedError:
        return None


def sbfp_qdq(x, precision: int, block_size: int, scaler_man: int, scaler_exp: int, scaler_bias: int,
             scaler_flush: bool = True, clamp: bool = True, symmetric: bool = True, block_dim: int = -1,
             out_dtype: Optional[torch.dtype] = None):
    """ScaledBlockFloatingPoint Q->DQ (numerical/format.py:453-479), one launch."""
    require_gpu(x, "sbfp_qdq")
    op = _ops.sbfp_qdq if (x.requires_grad and torch.is_grad_enabled()) else _ops.sbfp_qdq_nograd
    return op(x, precision, block_size, scaler_man, scaler_exp, scaler_bias, scaler_flush, clamp, symmetric, block_dim, out_dtype)


def mxfp_qdq(x, man: int, exp: int, block_size: int, block_dim: int = -1, out_dtype: Optional[torch.dtype] = None):
    """MXFP Q->DQ (numerical/format.py:545-564), one launch."""
    require_gpu(x, "mxfp_qdq")
    op = _ops.mxfp_qdq if (x.requires_grad and torch.is_grad_enabled()) else _ops.mxfp_qdq_nograd
    return op(x, man, exp, block_size, block_dim, out_dtype)


# ---------------------------------------------------------------------------------------------------- element formats
def float_qdq(x, man: int, exp: int, bias: int, flush_subnormal: bool, unsigned: bool = False,
              rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """Low-bit float Q->DQ (quant/quant_function.py:120-152 semantics), one launch."""
    require_gpu(x, "float_qdq")
    op = _ops.float_qdq if (x.requires_grad and torch.is_grad_enabled()) else _ops.float_qdq_nograd
    return op(x, man, exp, bias, flush_subnormal, unsigned, ROUNDING_CODE[rounding], out_dtype, _seed_arg(seed, rounding))


def fixed_qdq(x, precision: int, fraction: int, clamp: bool = True, symmetric: bool = True, rounding: str = "nearest",
              scale: Optional[torch.Tensor] = None, zero_point: Optional[torch.Tensor] = None,
              ch_axis: Optional[int] = None, group_size: Optional[int] = None,
              out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """Fixed-point Q->DQ with the affine wrapper of numerical/cast.py:278-296 fused in, one launch.
    scale None: bare FixedPoint.cast.  ch_axis None: per-tensor scale; else per-channel (group_size None) or
    per-group slabs of `group_size` channels."""
    require_gpu(x, "fixed_qdq")
    op = _ops.fixed_qdq if (x.requires_grad and torch.is_grad_enabled()) else _ops.fixed_qdq_nograd
    return op(x, precision, fraction, clamp, symmetric, ROUNDING_CODE[rounding], scale, zero_point, ch_axis,
                          group_size or None, out_dtype, _seed_arg(seed, rounding))


def float_qdq_multi(tensors, mantissa: int, exponent: int, bias: Optional[int] = None, flush_subnormal: bool = True,
                    unsigned: bool = False, rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None, seed: Optional[int] = None):
    """Low-bit floating point Q->DQ (format.py:208-233) of MANY tensors of one dtype on one device in as few launches as possible
    (`dmxq_float_qdq_multi`): the same results as `[float_qdq(t, mantissa, exponent, bias, ...) for t in tensors]` -- the bias casts
    of a layer's modules (modeling/nn/core.py:191-203), a few hundred elements each, as ONE launch."""
    tensors = list(tensors)
    for t in tensors:
        require_gpu(t, "float_qdq_multi")
    if bias is None:
        bias = (1 << (exponent - 1)) - 1
    return list(_ops.float_qdq_multi(tensors, mantissa, exponent, bias, bool(flush_subnormal), bool(unsigned), ROUNDING_CODE[rounding],
                                     out_dtype, _seed_arg(seed, rounding)))


def fixed_qdq_multi(tensors, precision: int, fraction: int, clamp: bool, symmetric: bool, scales, zero_points,
                    group_size: Optional[int] = None, rounding: str = "nearest", out_dtype: Optional[torch.dtype] = None,
                    seed: Optional[int] = None):
    """Affine integer Q->DQ (numerical/cast.py:278-296) of MANY weights of one dtype on one device in as few launches as
    possible (`dmxq_fixed_qdq_multi`): the same results as `[fixed_qdq(t, ..., scale=s, zero_point=z, ch_axis=0 (or None when
    s has one entry), group_size=group_size) for t, s, z in zip(tensors, scales, zero_points)]`."""
    tensors = list(tensors)
    for t in tensors:
        require_gpu(t, "fixed_qdq_multi")
    return list(_ops.fixed_qdq_multi(tensors, precision, fraction, clamp, symmetric, ROUNDING_CODE[rounding], list(scales),
                                     list(zero_points), group_size or 0, out_dtype, _seed_arg(seed, rounding)))


def fixed_float_qdq_multi(tensors, precision: int, fraction: int, clamp: bool, symmetric: bool, scales, zero_points, group_size: Optional[int],
                          float_tensors, mantissa: int, exponent: int, bias: Optional[int] = None, flush_subnormal: bool = True,
                          unsigned: bool = False):
    """`fixed_qdq_multi(tensors, ...)` AND `float_qdq_multi(float_tensors, ...)` (nearest rounding both) in ONE launch where the set allows
    it (`dmxq_fixed_float_qdq_multi`: a layer's INT8 weight casts and its bias casts); -> (fixed results, float results), bit-identical to
    the two calls."""
    tensors, float_tensors = list(tensors), list(float_tensors)
    for t in tensors + float_tensors:
        require_gpu(t, "fixed_float_qdq_multi")
    if bias is None:
        bias = (1 << (exponent - 1)) - 1
    if not tensors or not float_tensors:
        # one of the sets is empty: exactly the two calls (decided HERE so that both bindings behave alike -- ADVICE r5: the torch
        # binding refused an empty list where the ctypes one fell back)
        return (fixed_qdq_multi(tensors, precision, fraction, clamp, symmetric, scales, zero_points, group_size=group_size) if tensors else [],
                float_qdq_multi(float_tensors, mantissa, exponent, bias, flush_subnormal, unsigned) if float_tensors else [])
    a, b = _ops.fixed_float_qdq_multi(tensors, precision, fraction, clamp, symmetric, ROUNDING_CODE["nearest"], list(scales), list(zero_points),
                                      group_size or 0, float_tensors, mantissa, exponent, bias, bool(flush_subnormal), bool(unsigned),
                                      ROUNDING_CODE["nearest"], 0)
    return list(a), list(b)


# ---------------------------------------------------------------------------------------------------- sparsity
def _nm(score, x, K, M, block_dim, want_mask, want_y, mask_dtype, y_dtype):
    require_gpu(score, "nm_mask")
    if score.dim() == 0 or score.shape[block_dim] % M != 0:
        # sparse.py:166-168
        raise AssertionError(
            f"score has size {tuple(score.shape)} at dimension {block_dim}, not a multiple of block size {M}")
    mask, y = _ops.nm_mask(score, x, K, M, block_dim, want_mask, want_y, mask_dtype, y_dtype)
    return (mask if want_mask else None), (y if want_y else None)


def nm_mask(score, K: int, M: int, block_dim: int = -1, mask_dtype: Optional[torch.dtype] = None):
    """N:M mask (sparse.py:163-180): float mask in the score's dtype."""
    return _nm(score, None, K, M, block_dim, True, False, mask_dtype, None)[0]


def nm_sparsify(x, score, K: int, M: int, block_dim: int = -1, out_dtype: Optional[torch.dtype] = None,
                return_mask: bool = False):
    """Fused mask + apply: y = x * mask(score) (sparse.py:287-301); out dtype defaults to torch's promotion of
    (x.dtype, score.dtype), i.e. what `x * mask` yields in the reference."""
    require_gpu(x, "nm_sparsify")
    mask, y = _nm(score, x, K, M, block_dim, return_mask, True, None, out_dtype)
    return (y, mask) if return_mask else y


def _topk(score, x, density, want_mask, want_y, mask_dtype, y_dtype):
    require_gpu(score, "topk_mask")
    n_zero = int(score.numel() * (1.0 - density))  # sparse.py:116
    mask, y = _ops.topk_mask(score, x, n_zero, want_mask, want_y, mask_dtype, y_dtype)
    return (mask if want_mask else None), (y if want_y else None)


def topk_mask(score, density: float, mask_dtype: Optional[torch.dtype] = None):
    """Global top-k mask (sparse.py:109-123): the int(n * (1 - density)) lowest scores are zeroed; float mask in the
    score's dtype.  No sort: a radix select + one masking pass (csrc/topk.hip)."""
    return _topk(score, None, density, True, False, mask_dtype, None)[0]


def topk_sparsify(x, score, density: float, return_mask: bool = False):
    """x * topk_mask(score) in the same final pass (sparse.py:300), torch's type promotion for the product."""
    require_gpu(x, "topk_sparsify")
    mask, y = _topk(score, x, density, return_mask, True, None, None)
    return (y, mask) if return_mask else y


def bernoulli_mask(score, seed: Optional[int] = None, mask_dtype: Optional[torch.dtype] = None):
    """Bernoulli supermask (sparse.py:201-221): 1 with probability score."""
    require_gpu(score, "bernoulli_mask")
    return _ops.bernoulli_mask(score, _seed_arg(seed, "stochastic"), mask_dtype)


# ---------------------------------------------------------------------------------------------------- calibration
def group_minmax(x, ch_axis: int, group_size: int):
    """Per-group (slabs of `group_size` channels along ch_axis) min and max: two float32 [G] tensors."""
    require_gpu(x, "group_minmax")
    return _ops.group_minmax(x, ch_axis, group_size)


def group_minmax_accumulate(x, ch_axis: int, group_size: int, mn, mx):
    """running min / max of slabs of `group_size` channels updated IN PLACE by one launch: mn = min(mn, min(x)), mx = max(mx, max(x))
    (MinMaxObserver.forward as a whole, observer.py:173-193).  mn / mx: contiguous float32 GPU tensors of ceil(C / group_size) entries."""
    require_gpu(x, "group_minmax_accumulate")
    _ops.group_minmax_accumulate(x, ch_axis, group_size, mn, mx)


def qparams(mn, mx, qmin: int, qmax: int, symmetric_qscheme: bool):
    """(min,max) -> (scale fp32, zero_point int64), numerical/observer.py:59-115."""
    require_gpu(mn, "qparams")
    return _ops.qparams(mn, mx, qmin, qmax, symmetric_qscheme)


def histc(x, bins: int, lo: float = 0.0, hi: float = 0.0):
    """torch.histc(x, bins, min=lo, max=hi) as the HistogramObserver uses it (numerical/observer.py:470-472,
    489-491): float32 [bins] counts.  lo == hi selects the data's own range, widened by one either side when the
    data is constant (torch.histc's convention)."""
    require_gpu(x, "histc")
    lo, hi = float(lo), float(hi)
    if lo == hi and x.numel():
        mn, mx = group_minmax(x.reshape(1, -1), 0, 1)
        lo, hi = float(mn), float(mx)
        if lo == hi:
            lo, hi = lo - 1.0, hi + 1.0
    if x.numel() == 0 and not lo < hi:
        return torch.zeros(int(bins), dtype=torch.float32, device=x.device)
    if not (lo < hi and math.isfinite(lo) and math.isfinite(hi)):
        raise DmxqError(f"histc: needs a finite range with min < max, got [{lo}, {hi}]")
    return _ops.histc(x, int(bins), lo, hi)


def channel_maxabs(x, ch_axis: int):
    """max|x| per channel along ch_axis (numerical/smoothquant.py:285-299): float32 [C]."""
    require_gpu(x, "channel_maxabs")
    return _ops.channel_maxabs(x, ch_axis)


def smoothquant_scale(a_maxabs, b_maxabs, alpha: float, scale_min: float = 1e-5):
    require_gpu(a_maxabs, "smoothquant_scale")
    return _ops.smoothquant_scale(a_maxabs, b_maxabs, float(alpha), float(scale_min))


def scale_channels(x, scale, ch_axis: int, divide: bool, out_dtype: Optional[torch.dtype] = None):
    require_gpu(x, "scale_channels")
    return _ops.scale_channels(x, scale, ch_axis, divide, out_dtype)


# ---------------------------------------------------------------------------------------------------- approximator slot
UNARY_GELU, UNARY_GELU_TANH, UNARY_SILU, UNARY_QUICK_GELU, UNARY_EXP, UNARY_SILU_EXPERIMENTAL = range(6)


def gelu(x, approximate: str = "none", out_dtype: Optional[torch.dtype] = None):
    require_gpu(x, "gelu")
    return _ops.unary(x, UNARY_GELU_TANH if approximate == "tanh" else UNARY_GELU, 0.0, out_dtype)


def silu(x, out_dtype: Optional[torch.dtype] = None):
    """torch.nn.functional.silu (modeling/nn/torch_modules.py:1559-1576), exact function."""
    require_gpu(x, "silu")
    return _ops.unary(x, UNARY_SILU, 0.0, out_dtype)


def quick_gelu(x, out_dtype: Optional[torch.dtype] = None):
    """transformers' QuickGELUActivation `x * sigmoid(1.702 * x)` in the input dtype (custom_modules.py:112-117)."""
    require_gpu(x, "quick_gelu")
    return _ops.unary(x, UNARY_QUICK_GELU, 0.0, out_dtype)


def exp(x, out_dtype: Optional[torch.dtype] = None):
    """torch.exp (modeling/nn/torch_modules.py:236-242 Exp)."""
    require_gpu(x, "exp")
    return _ops.unary(x, UNARY_EXP, 0.0, out_dtype)


def silu_experimental(x, scale: float):
    """the reference's `experimental.silu` (functional/functions.py:7-21): relu(x.to(float16)) * scale -> float16"""
    require_gpu(x, "silu_experimental")
    return _ops.unary(x, UNARY_SILU_EXPERIMENTAL, float(scale), torch.float16)


def rope(x, cos, sin, unsqueeze_dim: int = 1):
    """APPLY_LLAMA_ROPE for ONE of q / k: (x * cos) + (rotate_half(x) * sin) in the tensor dtype (custom_modules.py:142-172);
    x [B, n1, n2, D], cos / sin [B, S, D].  Returns None when the HIP kernel does not take this shape / dtype mix (the
    caller keeps torch's own ops)."""
    require_gpu(x, "rope")
    try:
        return _ops.rope(x, cos, sin, unsqueeze_dim)
    except NotImplementedError:
        return None


def relu_cast(x, cast_in=None, cast_out=None, then_bfp=None):
    """A ReLU DmxModule in one launch: cast_out(relu(cast_in(x))); casts are FloatingPoint formats (nearest, signed) or None = SAME;
    then_bfp as binary_cast (dmxq_relu_cast_bfp).  None when not fusable."""
    require_gpu(x, "relu_cast")
    for f in (cast_in, cast_out):
        if f is not None and (f.rounding != "nearest" or f.unsigned):
            return None
    try:
        return _ops.relu_cast(x, _fmt4(cast_in), _fmt4(cast_out), int(then_bfp[1]) if then_bfp else 0, int(then_bfp[0]) if then_bfp else 0)
    except NotImplementedError:
        return None


def rope_cast(x, cos, sin, unsqueeze_dim: int = 1, cast_x=None, cast_cos=None, cast_sin=None, cast_out=None):
    """One operand of an ApplyRotaryPosEmb module with its casts in one launch: cast_out(rope(cast_x(x), cast_cos(cos), cast_sin(sin)));
    casts are FloatingPoint formats (nearest, signed) or None = SAME.  None when not fusable."""
    require_gpu(x, "rope_cast")
    for f in (cast_x, cast_cos, cast_sin, cast_out):
        if f is not None and (f.rounding != "nearest" or f.unsigned):
            return None
    try:
        return _ops.rope_cast(x, cos, sin, unsqueeze_dim, _fmt4(cast_x), _fmt4(cast_cos), _fmt4(cast_sin), _fmt4(cast_out))
    except NotImplementedError:
        return None


def softmax(x, dim: int = -1, input_clamp: Optional[float] = None, out_dtype: Optional[torch.dtype] = None):
    require_gpu(x, "softmax")
    d = dim % x.dim()
    xt = x if d == x.dim() - 1 else x.transpose(d, -1)
    out = _ops.softmax(xt, float(input_clamp) if input_clamp is not None else -math.inf, out_dtype)
    return out if d == x.dim() - 1 else out.transpose(d, -1)


def _cols(normalized_shape):
    cols = 1
    for s in (normalized_shape if not isinstance(normalized_shape, int) else (normalized_shape,)):
        cols *= s
    return cols


def layernorm(x, normalized_shape, weight=None, bias=None, eps: float = 1e-5,
              out_dtype: Optional[torch.dtype] = None):
    require_gpu(x, "layernorm")
    return _ops.norm(x, _cols(normalized_shape), weight, bias, float(eps), 0, out_dtype)


def rmsnorm(x, normalized_shape, weight=None, eps: Optional[float] = None, out_dtype: Optional[torch.dtype] = None):
    """torch.nn.functional.rms_norm over the trailing `normalized_shape` (modeling/nn/torch_modules.py:1144-1170);
    eps None = torch.finfo(x.dtype).eps, as torch."""
    require_gpu(x, "rmsnorm")
    eps = torch.finfo(x.dtype).eps if eps is None else eps
    return _ops.norm(x, _cols(normalized_shape), weight, None, float(eps), 1, out_dtype)


# ---- an activation / normalisation DmxModule in one launch (include/dmxq.h dmxq_unary_cast ...): cast_out(f(cast_in(x)))
_UNARY_KIND = {"gelu": UNARY_GELU, "gelu_tanh": UNARY_GELU_TANH, "silu": UNARY_SILU, "quick_gelu": UNARY_QUICK_GELU, "exp": UNARY_EXP}


def _casts_ok(*fmts):
    return all(f is None or (f.rounding == "nearest" and not f.unsigned) for f in fmts)


def unary_cast(x, func: str, cast_in=None, cast_out=None):
    """A GELU / SiLU / QuickGELU / Exp DmxModule in one launch; func in {"gelu", "gelu_tanh", "silu", "quick_gelu", "exp"};
    casts are FloatingPoint formats (nearest, signed) or None = SAME.  None when not fusable (the caller runs the three steps)."""
    require_gpu(x, "unary_cast")
    if not _casts_ok(cast_in, cast_out):
        return None
    try:
        return _ops.unary_cast(x, _UNARY_KIND[func], 0.0, _fmt4(cast_in), _fmt4(cast_out))
    except NotImplementedError:
        return None


def unary_cast_table(like, func: str, cast_in=None, cast_out=None, param: float = 0.0):
    """The 65,536-entry table of a unary DmxModule on `like`'s 16-bit dtype and device: table[p] = cast_out(f(cast_in(x_p))) for input
    pattern p, f in float64 rounded once (dmxq_unary_cast_table).  func as unary_cast, plus "silu_experimental" (param = scale).
    Casts: FloatingPoint formats (nearest, signed; rounding casts allowed) or None.  None when a cast is not tabulable."""
    require_gpu(like, "unary_cast_table")
    if like.dtype not in (torch.bfloat16, torch.float16) or not _casts_ok(cast_in, cast_out):
        return None
    kind = 5 if func == "silu_experimental" else _UNARY_KIND[func]
    try:
        return _ops.unary_cast_table(like, kind, float(param), _fmt4(cast_in), _fmt4(cast_out))
    except NotImplementedError:
        return None


def lut16_apply(x, table):
    """out[i] = table[x[i] as a 16-bit pattern] (dmxq_lut16_apply): the application of unary_cast_table's result.  None when x is not
    a whole number of aligned 16-byte vectors."""
    require_gpu(x, "lut16_apply")
    try:
        return _ops.lut16_apply(x, table)
    except NotImplementedError:
        return None


def softmax_cast(x, dim: int = -1, cast_in=None, cast_out=None, input_clamp: Optional[float] = None, then_bfp=None):
    """A Softmax DmxModule in one launch (softmax over the LAST dim only).  then_bfp = (precision, block_size): the consumer's BFP input
    cast (symmetric, nearest, along the same dim) applied to the result in the same launch.  None when not fusable."""
    require_gpu(x, "softmax_cast")
    if x.dim() == 0 or dim % x.dim() != x.dim() - 1 or not _casts_ok(cast_in, cast_out):
        return None
    try:
        return _ops.softmax_cast(x, float(input_clamp) if input_clamp is not None else -math.inf, _fmt4(cast_in), _fmt4(cast_out),
                                 int(then_bfp[1]) if then_bfp else 0, int(then_bfp[0]) if then_bfp else 0)
    except NotImplementedError:
        return None


def layernorm_cast(x, normalized_shape, weight=None, bias=None, eps: float = 1e-5, cast_in=None, cast_out=None, then_bfp=None):
    """A LayerNorm DmxModule in one launch (weight / bias in x's dtype); then_bfp = (precision, block_size): the consumers' BFP input
    cast (symmetric, nearest, along the rows) applied in the same launch.  None when not fusable."""
    require_gpu(x, "layernorm_cast")
    if not _casts_ok(cast_in, cast_out):
        return None
    try:
        return _ops.norm_cast(x, _cols(normalized_shape), weight, bias, float(eps), 0, _fmt4(cast_in), _fmt4(cast_out),
                              int(then_bfp[1]) if then_bfp else 0, int(then_bfp[0]) if then_bfp else 0)
    except NotImplementedError:
        return None


def rmsnorm_cast(x, normalized_shape, weight=None, eps: Optional[float] = None, cast_in=None, cast_out=None, then_bfp=None):
    """An RMSNorm DmxModule in one launch (eps None = torch.finfo(x.dtype).eps, as torch); then_bfp as layernorm_cast.  None when not
    fusable."""
    require_gpu(x, "rmsnorm_cast")
    if not _casts_ok(cast_in, cast_out):
        return None
    eps = torch.finfo(x.dtype).eps if eps is None else eps
    try:
        return _ops.norm_cast(x, _cols(normalized_shape), weight, None, float(eps), 1, _fmt4(cast_in), _fmt4(cast_out),
                              int(then_bfp[1]) if then_bfp else 0, int(then_bfp[0]) if then_bfp else 0)
    except NotImplementedError:
        return None
