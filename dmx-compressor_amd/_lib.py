"""ctypes binding of libdmxq.so (C ABI: include/dmxq.h).

This is the ONLY compute back-end of the package: there is no CPU or eager-PyTorch fallback.  If the HIP
library is missing or a tensor is not on a GPU, the ops raise instead of silently computing elsewhere.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMXQ_LIB_PATH") or os.path.join(_HERE, "lib", "libdmxq.so")  # (override: A/B runs of two builds; DMXQ_BINDING=ctypes only, ops.py refuses it otherwise)

F32, F16, BF16 = 0, 1, 2
ROUND_UP, ROUND_DOWN, ROUND_NEAREST, ROUND_STOCHASTIC = 0, 1, 2, 3
ROUNDING_CODE = {"up": ROUND_UP, "down": ROUND_DOWN, "nearest": ROUND_NEAREST, "stochastic": ROUND_STOCHASTIC}
OK, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_LAUNCH, ERR_PENDING = 0, 1, 2, 3, 4

_DTYPE_CODE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

_vp, _i64, _i32, _u64, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_float

# name -> argtypes; mirrors include/dmxq.h one to one (tests/test_abi.py checks header <-> table <-> .so)
SIGNATURES = {
    "dmxq_bfp_qdq": [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _u64, _vp],
    "dmxq_bfp_qdq_multi": [_vp, _i64, _i32, _i32, _i64, _i32, _i32, _i32, _u64, _vp],
    "dmxq_bfp_qdq_describe": [_i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, ctypes.c_char_p, _i64],
    "dmxq_sbfp_qdq": [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "dmxq_mxfp_qdq": [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "dmxq_bfp_pack": [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp],
    "dmxq_bfp_unpack": [_vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp],
    "dmxq_weight_hypernet": [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp],
    "dmxq_weight_hypernet_multi": [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _i32, _vp],
    "dmxq_weight_hypernet_strided": [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "dmxq_input_hypernet": [_vp, _i32, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp],
    "dmxq_binary_cast": [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp],
    "dmxq_relu_cast": [_vp, _vp, _i32, _i64, _vp, _vp, _vp],
    "dmxq_binary_cast_bfp": [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "dmxq_relu_cast_bfp": [_vp, _vp, _i32, _i64, _vp, _vp, _i64, _i64, _i32, _vp],
    "dmxq_unary_cast": [_vp, _vp, _i32, _i64, _i32, _f32, _vp, _vp, _vp],
    "dmxq_unary_cast_table": [_i32, _i32, _f32, _vp, _vp, _vp, _vp],
    "dmxq_lut16_apply": [_vp, _vp, _i64, _vp, _vp],
    "dmxq_softmax_cast": [_vp, _vp, _i32, _i64, _i64, _f32, _vp, _vp, _vp],
    "dmxq_softmax_cast_bfp": [_vp, _vp, _i32, _i64, _i64, _f32, _vp, _vp, _i64, _i32, _vp],
    "dmxq_layernorm_cast": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _f32, _vp, _vp, _vp],
    "dmxq_rmsnorm_cast": [_vp, _vp, _i32, _i64, _i64, _vp, _f32, _vp, _vp, _vp],
    "dmxq_layernorm_cast_bfp": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _f32, _vp, _vp, _i64, _i32, _vp],
    "dmxq_rmsnorm_cast_bfp": [_vp, _vp, _i32, _i64, _i64, _vp, _f32, _vp, _vp, _i64, _i32, _vp],
    "dmxq_float_qdq": [_vp, _vp, _i32, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _u64, _vp],
    "dmxq_fixed_qdq": [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _u64, _vp],
    "dmxq_fixed_qdq_multi": [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _u64, _vp],
    "dmxq_float_qdq_multi": [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _u64, _vp],
    "dmxq_fixed_float_qdq_multi": [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _u64, _vp],
    "dmxq_nm_mask": [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp],
    "dmxq_topk_workspace_bytes": [_i64],
    "dmxq_topk_mask": [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i64, _i64, _vp, _vp],
    "dmxq_bernoulli_mask": [_vp, _vp, _i32, _i32, _i64, _u64, _vp],
    "dmxq_group_minmax": [_vp, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "dmxq_group_minmax_accumulate": [_vp, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "dmxq_qparams": [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp],
    "dmxq_histc": [_vp, _i32, _i64, _i64, _f32, _f32, _vp, _vp],
    "dmxq_channel_maxabs": [_vp, _i32, _i64, _i64, _i64, _vp, _vp],
    "dmxq_smoothquant_scale": [_vp, _vp, _i64, _f32, _f32, _vp, _vp],
    "dmxq_scale_channels": [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _vp],
    "dmxq_gelu": [_vp, _vp, _i32, _i32, _i64, _i32, _vp],
    "dmxq_unary": [_vp, _vp, _i32, _i32, _i64, _i32, _f32, _vp],
    "dmxq_rmsnorm": [_vp, _vp, _i32, _i32, _i64, _i64, _vp, _i32, _f32, _vp],
    "dmxq_rope": [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _i32, _vp],
    "dmxq_rope_cast": [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    "dmxq_softmax": [_vp, _vp, _i32, _i32, _i64, _i64, _f32, _vp],
    "dmxq_layernorm": [_vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _i32, _f32, _vp],
}


class TensorDesc(ctypes.Structure):
    """dmxq_tensor_desc (include/dmxq.h)"""
    _fields_ = [("in_", _vp), ("out", _vp), ("outer", _i64), ("L", _i64), ("inner", _i64)]


class HypernetDesc(ctypes.Structure):
    """dmxq_hypernet_desc (include/dmxq.h)"""
    _fields_ = [("w", _vp), ("score", _vp), ("sq_scale", _vp), ("out", _vp), ("rows", _i64), ("L", _i64)]


class FloatFmt(ctypes.Structure):
    """dmxq_float_fmt (include/dmxq.h)"""
    _fields_ = [("man_bits", _i32), ("exp_bits", _i32), ("exp_bias", _i32), ("flush_subnormal", _i32)]


class AffineDesc(ctypes.Structure):
    """dmxq_affine_desc (include/dmxq.h)"""
    _fields_ = [("in_", _vp), ("out", _vp), ("scale", _vp), ("zero_point", _vp), ("outer", _i64), ("C", _i64), ("inner", _i64)]


class DmxqError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libdmxq.so; raises DmxqError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DmxqError(
                f"{LIB_PATH} not found: build the HIP library first (python dmx-compressor_amd/build.py or "
                "__graft_entry__.build()).  dmx_compressor_amd has no CPU/eager fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        L.dmxq_topk_workspace_bytes.restype = ctypes.c_int64
        L.dmxq_status_string.argtypes = [ctypes.c_int]
        L.dmxq_status_string.restype = ctypes.c_char_p
        L.dmxq_abi_version.restype = ctypes.c_int
        _lib = L
    return _lib


def check(status: int, what: str):
    if status != OK:
        msg = lib().dmxq_status_string(status).decode()
        if status == ERR_UNSUPPORTED:
            raise NotImplementedError(f"{what}: {msg}")
        raise DmxqError(f"{what}: {msg} (status {status})")


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise TypeError(f"dmxq kernels take float32/float16/bfloat16 tensors, got {dt}") from None


def require_gpu(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise DmxqError(
            f"{what}: tensor is on {x.device}; dmx_compressor_amd runs on MI355X (HIP) tensors only and has no "
            "CPU fallback"
        )


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream_of(x: torch.Tensor):
    """hipStream_t of torch's current stream on x's device (kernels are enqueued there, never on the null stream)."""
    idx = x.device.index
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx))


def split3(shape, dim):
    """[outer, L, inner] factorisation of a contiguous tensor around `dim`."""
    nd = len(shape)
    d = dim % nd
    outer = 1
    for s in shape[:d]:
        outer *= s
    inner = 1
    for s in shape[d + 1:]:
        inner *= s
    return outer, shape[d], inner
