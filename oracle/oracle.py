"""oracle/oracle.py — ctypes front-end of the CPU oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the product
package (dmx-compressor_amd/) never does and fails loudly without its HIP library instead.

The C file restates the reference's per-element arithmetic; this module restates the reference's *Python*
orchestration around it (layout handling and dtype round-trips), citing numerical/format.py and
numerical/cast.py of /root/reference/src/dmx/compressor.  CPU torch tensors are used only as typed
containers (bf16/fp16 <-> fp32 conversion, transposes); all quantisation arithmetic happens in oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB_PATH: another build of oracle.c (`make -C oracle asan`: AddressSanitizer + UBSan, tools/sanitize/run_sanitizers.sh)
_LIB_PATH = os.environ.get("ORACLE_LIB_PATH") or os.path.join(_HERE, "liboracle.so")
R_UP, R_DOWN, R_NEAREST, R_STOCHASTIC = 0, 1, 2, 3
ROUNDING = {"up": R_UP, "down": R_DOWN, "nearest": R_NEAREST, "stochastic": R_STOCHASTIC}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if os.environ.get("ORACLE_LIB_PATH"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp, i64, i32, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64
        L.oracle_float_qdq.argtypes = [fp, fp, i64, i32, i32, i32, i32, i32, u64]
        L.oracle_bfp_qdq.argtypes = [fp, fp, i64, i64, i64, i32, i32, i32, u64]
        L.oracle_fixed_qdq.argtypes = [fp, fp, i64, i64, i64, i32, i32, i32, i32, i32, fp, fp, i64, u64]
        L.oracle_nm_mask.argtypes = [fp, fp, fp, fp, i64, i32, i32]
        L.oracle_sbfp_qdq.argtypes = [fp, fp, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32]
        L.oracle_mxfp_qdq.argtypes = [fp, fp, i64, i64, i64, i32, i32]
        L.oracle_group_minmax.argtypes = [fp, i64, i64, i64, i64, fp, fp]
        L.oracle_qparams.argtypes = [fp, fp, i64, i32, i32, i32, fp, fp]
        L.oracle_channel_maxabs.argtypes = [fp, i64, i64, i64, fp]
        L.oracle_histc.argtypes = [fp, i64, i64, ctypes.c_float, ctypes.c_float, fp]
        L.oracle_bernoulli_mask.argtypes = [fp, fp, i64, u64]
        for f in ("oracle_sbfp_qdq", "oracle_mxfp_qdq", "oracle_float_qdq", "oracle_bfp_qdq", "oracle_fixed_qdq", "oracle_nm_mask",
                  "oracle_group_minmax", "oracle_qparams", "oracle_channel_maxabs", "oracle_histc", "oracle_bernoulli_mask"):
            getattr(L, f).restype = ctypes.c_int
        _lib = L
    return _lib


def _f32c(x: torch.Tensor) -> torch.Tensor:
    assert not x.is_cuda, "the oracle is CPU-only"
    return x.detach().to(torch.float32).contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what}: bad argument (rc={rc})")


# ------------------------------------------------------------------------------------------------ formats
def float_quantize(x, man, exp, bias, flush_subnormal, rounding="nearest", seed=0):
    """quant/quant_function.py:120-152 -> quant_cpu.cpp:359-402.  fp32 in -> fp32 out."""
    xi = _f32c(x)
    out = torch.empty_like(xi)
    _check(lib().oracle_float_qdq(_ptr(xi), _ptr(out), xi.numel(), man, exp, bias, int(flush_subnormal),
                                  ROUNDING[rounding], seed), "float_qdq")
    return out


def floating_point_cast(x, man, exp, bias, flush_subnormal, unsigned=False, rounding="nearest", seed=0):
    """numerical/format.py:208-233 FloatingPoint.cast (returns fp32, or x itself for the two native bypasses)."""
    rep = f"FP[{'0' if unsigned else '1'}|{exp}|{man},{bias}]({'F' if flush_subnormal else '_'}{'N' if rounding=='nearest' else 'S'})"
    if (x.dtype == torch.float32 and rep == "FP[1|8|23,127](_N)") or (x.dtype == torch.float16 and rep == "FP[1|5|10,15](_N)"):
        y = x
    else:
        y = float_quantize(x, man, exp, bias, flush_subnormal, rounding, seed)
    if rep == "FP[1|5|10,15](FN)":  # format.py:222-232 extra fp16 subnormal flush
        y = torch.where(y.abs() < 2.0 ** -14, torch.zeros((), dtype=y.dtype), y)
    return y.abs() if unsigned else y


def bfp_cast(x, precision, block_size, block_dim=-1, symmetric=True, rounding="nearest", seed=0):
    """numerical/format.py:304-343 BlockFloatingPoint.cast: fp32 result, same shape (values are what the tests
    compare).  The reference brings the block dim last with transpose(block_dim, -1); here movedim(block_dim, -1) --
    the same blocks and values, but a row order in which the stochastic draws are numbered like the kernels number
    them, (outer index, inner index, position along the block dim), also for tensors of more than two dims."""
    xf = x.detach().to(torch.float32)
    if block_size == 1:
        return float_quantize(xf, precision - 2, 8, 127, False, rounding, seed)
    xt = xf.movedim(block_dim, -1)
    shp = xt.shape
    x2 = xt.reshape(-1, shp[-1]).contiguous()
    out = torch.empty_like(x2)
    _check(lib().oracle_bfp_qdq(_ptr(x2), _ptr(out), x2.shape[0], x2.shape[1], block_size, precision,
                                ROUNDING[rounding], int(symmetric), seed), "bfp_qdq")
    return out.reshape(shp).movedim(-1, block_dim)


def block_quantize_native(a, wl, dim=-1, symmetric=True, rounding="nearest", seed=0):
    """quant_cpu.cpp:299-311 block_quantize_<rounding>(a, wl, dim, symmetric) with get_max_entry's block layouts (:277-297):
    dim -1 = the whole tensor is one block, 0 = one block per leading index, d > 0 = one block per index of dim d.
    symmetric False = the NATIVE branch (:247-253), not the Python post-pass."""
    af = a.detach().to(torch.float32).contiguous()
    if af.numel() == 0:
        return af.clone()
    if dim == -1:
        x2 = af.reshape(1, -1)
    elif dim == 0:
        x2 = af.reshape(af.shape[0], -1)
    else:
        t = af.transpose(0, dim).contiguous()
        x2 = t.reshape(t.shape[0], -1)
    out = torch.empty_like(x2)
    _check(lib().oracle_bfp_qdq(_ptr(x2), _ptr(out), x2.shape[0], x2.shape[1], max(x2.shape[1], 2), wl, ROUNDING[rounding],
                                1 if symmetric else 2, seed), "bfp_qdq(native)")
    if dim in (-1, 0):
        return out.reshape(af.shape)
    return out.reshape(t.shape).transpose(0, dim).contiguous()


def bfp_pack(x, precision, block_size, symmetric=True):
    """Packed on-wire BFP (the QuantizeBFP / DequantizeBFP pair named by numerical/cast.py:34-55, ids numerical/onnx.py):
    int8 two's-complement mantissa codes + one uint8 shared exponent per block, stated INDEPENDENTLY of the kernel's
    code extraction: codes = Q->DQ(x) / 2^(e - (p-2)) with Q->DQ the oracle's BFP cast and e the exponent of the block
    maximum; exps = biased fp32 exponent field of the block maximum.  Blocks with a zero / denormal maximum pack to
    zero codes with exps = 0, Inf / NaN maxima to zero codes with exps = 255 (include/dmxq.h)."""
    xf = x.detach().to(torch.float32).contiguous()
    L = xf.shape[-1]
    x2 = xf.reshape(-1, L)
    q = bfp_cast(x2, precision, block_size, -1, symmetric).numpy().astype(np.float64)
    nblk = -(-L // block_size)
    pad = nblk * block_size - L
    a = np.abs(x2.numpy())
    a = np.pad(a, ((0, 0), (0, pad)))  # zeros do not change a maximum; NaN must win like torch.max: handled below
    blk = a.reshape(-1, nblk, block_size)
    m = np.where(np.isnan(blk).any(-1), np.float32(np.nan), blk.max(-1)).astype(np.float32)
    eb = ((m.view(np.uint32) >> 23) & 0xFF).astype(np.int64)
    quantum = np.ldexp(1.0, (eb - 127 - (precision - 2)))
    qpad = np.pad(q, ((0, 0), (0, pad))).reshape(-1, nblk, block_size)
    ok = ((eb > 0) & (eb < 255))[..., None]
    codes = np.where(ok, qpad / quantum[..., None], 0.0)
    codes = np.where(np.isfinite(codes), codes, 0.0)
    assert np.all(codes == np.round(codes)) and np.all(np.abs(codes) <= 2 ** (precision - 1)), "codes must be integers"
    mant = torch.from_numpy(codes.reshape(-1, nblk * block_size)[:, :L].astype(np.int8)).reshape(xf.shape)
    exps = torch.from_numpy(eb.astype(np.uint8)).reshape(tuple(xf.shape[:-1]) + (nblk,))
    return mant, exps


def fixed_point_cast(x, precision, fraction, clamp=True, symmetric=True, rounding="nearest", seed=0):
    """numerical/format.py:134-142 FixedPoint.cast -> quant_cpu.cpp:148-167."""
    xi = _f32c(x)
    out = torch.empty_like(xi)
    _check(lib().oracle_fixed_qdq(_ptr(xi), _ptr(out), 1, 1, xi.numel(), precision, fraction, int(clamp),
                                  int(symmetric), ROUNDING[rounding], None, None, 1, seed), "fixed_qdq")
    return out


def fixed_point_affine_cast(x, precision, fraction, clamp, symmetric, scale, zero_point, ch_axis=None,
                            group_size=None, rounding="nearest", seed=0):
    """numerical/cast.py:278-296: x/sc + zp -> FixedPoint.cast -> (x - zp)*sc, fp32 result.
    ch_axis None = per-tensor (scale has one element); group_size None with ch_axis = per-channel."""
    xi = _f32c(x)
    out = torch.empty_like(xi)
    sc = scale.detach().to(torch.float32).contiguous()
    zp = zero_point.detach().to(torch.int64).contiguous()
    if ch_axis is None:
        outer, C, inner, gs = 1, 1, xi.numel(), 1
    else:
        ax = ch_axis % xi.dim()
        C = xi.shape[ax]
        outer = int(np.prod(xi.shape[:ax], dtype=np.int64))
        inner = int(np.prod(xi.shape[ax + 1:], dtype=np.int64))
        gs = group_size or 1
    _check(lib().oracle_fixed_qdq(_ptr(xi), _ptr(out), outer, C, inner, precision, fraction, int(clamp),
                                  int(symmetric), ROUNDING[rounding], _ptr(sc), _ptr(zp), gs, seed), "fixed_qdq")
    return out


def _blocked(x, block_dim, fn):
    """transpose block_dim to the end, flatten to [rows, L], run fn(in2d, out2d), restore (format.py:455-479 pattern)"""
    xt = x.detach().to(torch.float32).transpose(block_dim, -1)
    shp = xt.shape
    x2 = xt.reshape(-1, shp[-1]).contiguous()
    out = torch.empty_like(x2)
    fn(x2, out)
    return out.reshape(shp).transpose_(block_dim, -1)


def sbfp_cast(x, precision, block_size, man, exp, bias, flush_subnormal=True, clamp=True, symmetric=True, block_dim=-1):
    """numerical/format.py:453-479 ScaledBlockFloatingPoint.cast (fp32 result)."""
    return _blocked(x, block_dim, lambda a, o: _check(lib().oracle_sbfp_qdq(
        _ptr(a), _ptr(o), a.shape[0], a.shape[1], block_size, precision, int(clamp), int(symmetric), man, exp, bias,
        int(flush_subnormal)), "sbfp_qdq"))


def mxfp_cast(x, man, exp, block_size, block_dim=-1):
    """numerical/format.py:545-564 MXFP.cast (fp32 result; intended block layout, zero blocks stay zero)."""
    return _blocked(x, block_dim, lambda a, o: _check(lib().oracle_mxfp_qdq(
        _ptr(a), _ptr(o), a.shape[0], a.shape[1], block_size, man, exp), "mxfp_qdq"))


def cast_to(x, cast_fn):
    """numerical/cast.py:261-306 CastTo.forward dtype contract: remember physical dtype, cast in fp32,
    `.to(physical_dtype)` (torch CPU RNE narrowing)."""
    return cast_fn(x).to(x.dtype)


# ------------------------------------------------------------------------------------------------ sparsity
def nm_mask(score, K, M, block_dim=-1):
    """sparse.py:163-180 BlockTopK.forward: float mask in score's dtype."""
    assert score.shape[block_dim] % M == 0
    st = score.detach().transpose(block_dim, -1)
    shp = st.shape
    s2 = st.reshape(-1, M).to(torch.float32).contiguous()
    mask = torch.empty_like(s2)
    _check(lib().oracle_nm_mask(_ptr(s2), None, _ptr(mask), None, s2.shape[0], M, K), "nm_mask")
    return mask.reshape(shp).transpose_(block_dim, -1).to(score.dtype)


def sparsify(x, score, K, M, block_dim=-1):
    """sparse.py:287-301 Sparsify.forward: x * mask (type promotion as torch does)."""
    return x * nm_mask(score, K, M, block_dim)


# ------------------------------------------------------------------------------------------------ calibration
def group_minmax(x, ch_axis, group_size):
    xi = _f32c(x)
    ax = ch_axis % xi.dim()
    C = xi.shape[ax]
    outer = int(np.prod(xi.shape[:ax], dtype=np.int64))
    inner = int(np.prod(xi.shape[ax + 1:], dtype=np.int64))
    G = (C + group_size - 1) // group_size
    mn, mx = torch.empty(G), torch.empty(G)
    _check(lib().oracle_group_minmax(_ptr(xi), outer, C, inner, group_size, _ptr(mn), _ptr(mx)), "group_minmax")
    return mn, mx


def qparams(mn, mx, precision, fmt_symmetric, qscheme_symmetric):
    qmin = -(2 ** (precision - 1)) + (1 if fmt_symmetric else 0)
    qmax = 2 ** (precision - 1) - 1
    mn, mx = _f32c(mn), _f32c(mx)
    sc = torch.empty(mn.numel())
    zp = torch.empty(mn.numel(), dtype=torch.int64)
    _check(lib().oracle_qparams(_ptr(mn), _ptr(mx), mn.numel(), qmin, qmax, int(qscheme_symmetric), _ptr(sc), _ptr(zp)), "qparams")
    return sc, zp


def channel_maxabs(x, ch_axis):
    xi = _f32c(x)
    ax = ch_axis % xi.dim()
    C = xi.shape[ax]
    outer = int(np.prod(xi.shape[:ax], dtype=np.int64))
    inner = int(np.prod(xi.shape[ax + 1:], dtype=np.int64))
    out = torch.empty(C)
    _check(lib().oracle_channel_maxabs(_ptr(xi), outer, C, inner, _ptr(out)), "channel_maxabs")
    return out


def histc(x, bins, lo, hi):
    xi = _f32c(x).reshape(-1)
    out = torch.empty(bins)
    _check(lib().oracle_histc(_ptr(xi), xi.numel(), bins, ctypes.c_float(lo), ctypes.c_float(hi), _ptr(out)), "histc")
    return out


def topk_mask(score, density):
    """sparse.py:109-123 TopK.forward: the int(n * (1 - density)) lowest scores are zeroed (float mask in the score's
    dtype).  The reference sorts with torch.argsort's default UNSTABLE algorithm, so which of several scores equal to the
    threshold value are dropped is implementation-defined there; this restatement (and the HIP kernel) use the stable
    order -- lowest index first, -0 == +0, NaN largest -- and are pinned against the reference on inputs without a tie
    at the threshold (oracle/gen_golden.py)."""
    s = score.detach().to(torch.float32).reshape(-1).numpy()
    n_zero = int(score.numel() * (1.0 - density))
    idx = np.argsort(s, kind="stable")[:n_zero]
    mask = np.ones(s.shape, dtype=np.float32)
    mask[idx] = 0.0
    return torch.from_numpy(mask).reshape(score.shape).to(score.dtype)


def bernoulli_mask(score, seed):
    sc = _f32c(score)
    out = torch.empty_like(sc)
    _check(lib().oracle_bernoulli_mask(_ptr(sc), _ptr(out), sc.numel(), seed), "bernoulli_mask")
    return out.to(score.dtype)


def silu_experimental(x, scale):
    """functional/functions.py:7-21 `experimental.silu`: relu(input.to(float16)) * scale -> float16 (torch's own CPU ops:
    the statement IS the algorithm).  Pinned against the reference's SiLU module in tests/golden/approx.npz."""
    return torch.nn.functional.relu(x.detach().to(torch.float16)) * scale
