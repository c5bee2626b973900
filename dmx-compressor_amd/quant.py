"""S1 seam: the reference's native-module surface, backed by libdmxq (HIP, gfx950).

`quant/quant_function.py:38-43` picks `quant_cuda` for GPU tensors; `quant_hip` below exposes the same pybind
function names and argument meaning (quant_cuda.cpp:116-139 / quant_cpu.cpp:424-440), and the three Python
wrappers keep the reference's signatures and assertions (quant_function.py:47-152).  Semantics follow the
reference's CPU path bit-for-bit (BASELINE.json: parity is against the CPU path; e.g. fixed-point nearest is
half-to-even here, where the CUDA twin rounds half away from zero).
  - `a` must be a contiguous float32 GPU tensor (the pybind functions TORCH_CHECK contiguity and use
    data_ptr<float>()); a new float32 tensor is returned.
  - block_quantize's `dim`: 0 = one shared exponent per row `a[i, ...]` (the only form the Python layer uses,
    format.py:328-336), -1 = one for the whole tensor, d > 0 = one per index of dimension d.
"""
from types import SimpleNamespace

import torch

from . import ops
from ._lib import require_gpu

__all__ = ["fixed_point_quantize", "block_quantize", "float_quantize", "quant_hip"]


def _check_native_input(a, what):
    if not isinstance(a, torch.Tensor):
        raise TypeError(f"{what}: expected a torch.Tensor")
    require_gpu(a, what)
    if not a.is_contiguous():
        raise RuntimeError("a must be contiguous")  # CHECK_CONTIGUOUS, quant_cuda.cpp:7-11
    if a.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {a.dtype}")  # data_ptr<float>()


def _block(a, wl, dim, symmetric, rounding):
    _check_native_input(a, "block_quantize")
    if a.numel() == 0:
        return torch.zeros_like(a)
    # one launch on a [rows, L] view with one block per row.  symmetric = False is the NATIVE branch of
    # block_quantize_helper (quant_cpu.cpp:247-253), which the Python layer never requests (format.py:332) but which is
    # part of this seam; it is NOT the asymmetric format "(_N)" (that is Format.cast's post-pass, ops.bfp_qdq).
    if dim == -1:  # one block = the whole tensor (get_max_entry dim == -1, quant_cpu.cpp:280-283)
        return ops.block_quantize(a.reshape(1, -1), wl, symmetric, rounding).reshape(a.shape)
    if dim == 0:   # one block per leading index
        return ops.block_quantize(a.reshape(a.shape[0], -1), wl, symmetric, rounding).reshape(a.shape)
    # one block per index of `dim`: the block runs over every other dimension (quant_cpu.cpp:289-295)
    t = a.transpose(0, dim).contiguous()
    y = ops.block_quantize(t.reshape(t.shape[0], -1), wl, symmetric, rounding).reshape(t.shape)
    return y.transpose(0, dim).contiguous()


def _float(a, man_bits, exp_bits, exp_bias, flush_subnormal, rounding):
    _check_native_input(a, "float_quantize")
    return ops.float_qdq(a, man_bits, exp_bits, exp_bias, flush_subnormal, False, rounding)


def _fixed(a, wl, fl, clamp, symmetric, rounding):
    _check_native_input(a, "fixed_point_quantize")
    return ops.fixed_qdq(a, wl, fl, clamp, symmetric, rounding)


def _mk(rounding):
    return dict(
        block=lambda a, wl, dim, symmetric: _block(a, wl, dim, symmetric, rounding),
        flt=lambda a, man_bits, exp_bits, exp_bias, flush_subnormal: _float(a, man_bits, exp_bits, exp_bias, flush_subnormal, rounding),
        fix=lambda a, wl, fl, clamp, symmetric: _fixed(a, wl, fl, clamp, symmetric, rounding),
    )


def _fixed_mask(a, wl, fl, symmetric, rounding):
    """fixed_point_quantize_{nearest,stochastic}_mask (quant_cpu.cpp:86-126, exported at :424-440): (clamped result, uint8 mask of the
    elements the clamp changed).  Unreachable from the reference's Python wrappers, part of the pybind seam: composed here from the
    clamped and the unclamped cast (the same random draws for both through one explicit seed) and two compares against
    sim_helper.cpp:5-12's limits -- two launches of dmxq_fixed_qdq plus torch's compare kernels, not a hot path."""
    _check_native_input(a, "fixed_point_quantize_mask")
    seed = ops._next_seed() if rounding == "stochastic" else 0
    o = ops.fixed_qdq(a, wl, fl, True, symmetric, rounding, seed=seed)
    u = ops.fixed_qdq(a, wl, fl, False, symmetric, rounding, seed=seed)
    t_min = -(2.0 ** (wl - fl - 1))
    t_max = -t_min - 2.0 ** (-fl)
    if symmetric:
        t_min += 2.0 ** (-fl)
    return o, ((u > t_max) | (u < t_min)).to(torch.uint8)


quant_hip = SimpleNamespace()
quant_hip.fixed_point_quantize_nearest_mask = lambda a, wl, fl, symmetric: _fixed_mask(a, wl, fl, symmetric, "nearest")
quant_hip.fixed_point_quantize_stochastic_mask = lambda a, wl, fl, symmetric: _fixed_mask(a, wl, fl, symmetric, "stochastic")
for _r in ("nearest", "stochastic", "down", "up"):
    _f = _mk(_r)
    setattr(quant_hip, f"block_quantize_{_r}", _f["block"])
    setattr(quant_hip, f"float_quantize_{_r}", _f["flt"])
    setattr(quant_hip, f"fixed_point_quantize_{_r}", _f["fix"])


def fixed_point_quantize(x, wl, fl, clamp=True, symmetric=False, rounding="stochastic"):
    """quant_function.py:47-84."""
    assert isinstance(x, torch.Tensor)
    assert rounding in ["stochastic", "nearest", "up", "down"]
    if wl == -1 and fl != -1:
        raise ValueError("fixed point {} wl {}, fl {}".format("", wl, fl))
    return getattr(quant_hip, f"fixed_point_quantize_{rounding}")(x.contiguous(), wl, fl, clamp, symmetric)


def block_quantize(x, wl, dim=-1, symmetric=True, rounding="stochastic"):
    """quant_function.py:87-117."""
    assert isinstance(x, torch.Tensor), "x is not a single precision Floating Point Tensor"
    assert rounding in ["stochastic", "nearest", "down", "up"], "invalid rounding mode, {}".format(rounding)
    return getattr(quant_hip, f"block_quantize_{rounding}")(x.contiguous(), wl, dim, symmetric)


def float_quantize(x, exp, man, bias=None, flush_subnormal=True, rounding="stochastic"):
    """quant_function.py:120-152."""
    assert isinstance(x, torch.Tensor), "x is not a single precision Floating Point Tensor"
    assert rounding in ["stochastic", "nearest"], "invalid rounding mode, {}".format(rounding)
    if bias is None:
        bias = 2 ** (exp - 1) - 1
    return getattr(quant_hip, f"float_quantize_{rounding}")(x.contiguous(), man, exp, bias, flush_subnormal)
