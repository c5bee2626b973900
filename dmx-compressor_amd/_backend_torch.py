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
