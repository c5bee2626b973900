"""The torch-extension BINDING of the C ABI: `torch.ops.dmxq.*`, registered by `lib/dmxq_torch.so` (csrc/torch_binding.cpp: device guard,
torch's current HIP stream, output allocation, one C-ABI call per op, meta kernels).  What belongs to THIS binding alone lives here:
loading the library, resolving the overloads once, and the straight-through-estimator backward (`torch.library.register_autograd`)
of the fake-quantisation ops -- the reference's CastToFormat / STE (numerical/cast.py:19-55: `grad_output` passed through unchanged).
The front ends are `_front.py`."""
import os

import torch

from . import _lib
from ._lib import DmxqError

# DMXQ_TORCH_LIB_PATH: another build of the binding, together with DMXQ_LIB_PATH = the libdmxq it was linked against (an instrumented
# pair: tools/sanitize/run_sanitizers.sh)
TORCH_LIB_PATH = os.environ.get("DMXQ_TORCH_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "dmxq_torch.so")


def _load():
    _lib.lib()  # libdmxq.so first (raises DmxqError when it has not been built): dmxq_torch.so links against it
    if not os.path.exists(TORCH_LIB_PATH):
        raise DmxqError(f"{TORCH_LIB_PATH} not found: build the torch extension first (python dmx-compressor_amd/build.py or "
                        "__graft_entry__.build()), or set DMXQ_BINDING=ctypes for the compiler-free binding")
    torch.ops.load_library(TORCH_LIB_PATH)
    return torch.ops.dmxq


class _Overloads:
    """`torch.ops.dmxq.<name>.default` resolved once: calling an OpOverload skips the packet's per-call overload
    resolution (~1.5 us of the host cost of a call)."""

    def __init__(self, ns):
        self._ns = ns

    def __getattr__(self, name):
        op = getattr(self._ns, name).default
        setattr(self, name, op)
        return op


RAW = _Overloads(_load())


# ---------------------------------------------------------------------------------------------------- direct entry points (round 6)
# `dmxq_torch.so` is also a Python extension module (`PyInit_dmxq_fast`, csrc/torch_binding.cpp): the SAME C++ functions the dispatcher
# calls, without the dispatcher (~2 us of schema matching, boxing and dispatch per call; profiles/r06_host_overhead.txt).  Eager
# inference calls take them; while torch.compile / export traces, and for everything that needs autograd, the dispatcher op stays
# (`*_nograd` are the no-gradient variants the front end picks by itself; the other names have no autograd formula to lose).
# DMXQ_NO_FAST_CALLS=1 switches them off (A/B runs).
_FAST_NAMES = {"bfp_qdq_nograd": "bfp_qdq", "float_qdq_nograd": "float_qdq", "fixed_qdq_nograd": "fixed_qdq", "sbfp_qdq_nograd": "sbfp_qdq",
               "mxfp_qdq_nograd": "mxfp_qdq", "weight_hypernet": "weight_hypernet", "input_hypernet": "input_hypernet", "binary_cast": "binary_cast",
               "relu_cast": "relu_cast", "scale_channels": "scale_channels", "rope_cast": "rope_cast", "unary_cast": "unary_cast",
               "lut16_apply": "lut16_apply", "softmax_cast": "softmax_cast", "norm_cast": "norm_cast"}
FAST = None


def _load_fast():
    import importlib.machinery
    import importlib.util

    loader = importlib.machinery.ExtensionFileLoader("dmxq_fast", TORCH_LIB_PATH)
    mod = importlib.util.module_from_spec(importlib.util.spec_from_loader("dmxq_fast", loader))
    loader.exec_module(mod)
    return mod


def _direct(fast_fn, op):
    is_compiling = torch.compiler.is_compiling

    def call(*args):
        return op(*args) if is_compiling() else fast_fn(*args)

    call.__wrapped__ = op
    return call


if not os.environ.get("DMXQ_NO_FAST_CALLS"):
    try:
        FAST = _load_fast()
    except (ImportError, OSError, AttributeError):   # (a binding built without the entry points: the dispatcher serves everything)
        FAST = None
    if FAST is not None:
        for _raw_name, _fast_name in _FAST_NAMES.items():
            setattr(RAW, _raw_name, _direct(getattr(FAST, _fast_name), getattr(RAW, _raw_name)))
        del _raw_name, _fast_name


# ---------------------------------------------------------------------------------------------------- autograd (STE)
def _ste_setup(ctx, inputs, output):
    ctx.in_dtype = inputs[0].dtype


def _ste_backward(n_args):
    def backward(ctx, g):
        if g is not None and g.dtype != ctx.in_dtype:
            g = g.to(ctx.in_dtype)
        return (g,) + (None,) * (n_args - 1)

    return backward


for _name, _n in (("bfp_qdq", 8), ("sbfp_qdq", 11), ("mxfp_qdq", 6), ("float_qdq", 9), ("fixed_qdq", 12)):
    torch.library.register_autograd(f"dmxq::{_name}", _ste_backward(_n), setup_context=_ste_setup)
del _name, _n
