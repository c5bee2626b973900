"""Tensor-level front ends of the libdmxq C ABI (include/dmxq.h): each function is ONE kernel launch on torch's
current HIP stream, allocating its output like the reference's native functions do (`zeros_like` + return,
quant/quant_cuda/quant_cuda.cpp:116-139 — here `empty`, every element is written).

ONE front end (`_front.py`: argument spelling and policy) over two bindings of the same C ABI with the same raw schema:
  * `_backend_torch` (default)  `torch.ops.dmxq.*`: the TORCH_LIBRARY extension `lib/dmxq_torch.so` (csrc/torch_binding.cpp),
                                with meta kernels (torch.compile / torch.export trace through it) and a registered
                                straight-through backward;
  * `_backend_ctypes`           plain ctypes on `lib/libdmxq.so`: needs no C++ compiler against the torch headers.
`DMXQ_BINDING=ctypes` selects the second; a missing `dmxq_torch.so` does NOT silently select it (the GPU box must show
the extension loaded) unless that variable says so.

No CPU path in either: a non-GPU tensor or a missing library raises `DmxqError`.
"""
import os

from . import _front

_binding = os.environ.get("DMXQ_BINDING", "torch").lower()
if _binding == "ctypes":
    from . import _backend_ctypes as _raw
    BINDING = "ctypes"
elif _binding == "torch":
    if os.environ.get("DMXQ_LIB_PATH") and not os.environ.get("DMXQ_TORCH_LIB_PATH"):
        # dmxq_torch.so is linked against lib/libdmxq.so (rpath $ORIGIN): the override would only reach the ctypes handle, an A/B
        # run would silently measure the stock build and two copies of the library (each with its own thread_local launch
        # state) would be loaded
        raise ImportError("DMXQ_LIB_PATH (an alternative libdmxq.so) only works with DMXQ_BINDING=ctypes: the torch extension "
                          "resolves lib/libdmxq.so through its rpath (or name the binding that was linked against it: DMXQ_TORCH_LIB_PATH)")
    from ._backend_torch import RAW as _raw
    BINDING = "torch"
else:
    raise ImportError(f"DMXQ_BINDING={_binding!r}: expected 'torch' or 'ctypes'")
_front.bind(_raw)
from ._front import *  # noqa: E402,F401,F403
from ._front import __all__, _next_seed  # noqa: E402,F401


def front(binding: str):
    """A front-end namespace bound to the NAMED binding, whatever DMXQ_BINDING selected for `ops` itself: `_front.py` executed once
    more over the other raw namespace (tests run the same calls through both bindings in one process)."""
    import importlib.util
    if binding == BINDING:
        return _front
    if binding == "ctypes":
        from . import _backend_ctypes as raw
    elif binding == "torch":
        from ._backend_torch import RAW as raw
    else:
        raise ValueError(binding)
    spec = importlib.util.find_spec(_front.__name__)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.bind(raw)
    return m
