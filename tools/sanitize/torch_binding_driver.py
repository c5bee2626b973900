#!/usr/bin/env python3
"""tools/sanitize/torch_binding_driver.py — csrc/torch_binding.cpp under AddressSanitizer + UBSan without a GPU: every `torch.ops.dmxq.*`
op is called (1) on META tensors -- its shape / dtype propagation kernel, what torch.compile / torch.export run -- and (2) on CPU
tensors, where the real kernel's first check must raise ("no CPU fallback") before anything is allocated or packed.  The descriptor
packing of the multi-tensor ops behind that check needs device tensors and is covered on the library side (host_driver.py: the same
structs filled through ctypes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dmx_compressor_amd as d  # noqa: E402,F401
from dmx_compressor_amd import _backend_torch  # noqa: E402

assert "asan" in _backend_torch.TORCH_LIB_PATH or os.environ.get("DMXQ_DRIVER_ANY_LIB"), _backend_torch.TORCH_LIB_PATH
ops = torch.ops.dmxq
n_meta = n_cpu = 0


def both(name, make_args):
    """make_args(device) -> args"""
    global n_meta, n_cpu
    op = getattr(ops, name)
    out = op(*make_args("meta"))
    n_meta += 1
    try:
        op(*make_args("cpu"))
    except (RuntimeError, NotImplementedError, TypeError):
        n_cpu += 1
    else:
        raise AssertionError(f"{name} accepted CPU tensors")
    return out


def t(dev, *shape, dtype=torch.bfloat16):
    return torch.empty(*shape, dtype=dtype, device=dev)


for dt in (torch.bfloat16, torch.float16, torch.float32):
    for shape in ((4, 64), (2, 3, 128), (0, 16), (7, 1500)):
        y = both("bfp_qdq", lambda dev: (t(dev, *shape, dtype=dt), 8, 16, -1, True, 2, None, 0))
        assert y.shape == torch.Size(shape) and y.dtype == dt
        both("bfp_qdq_nograd", lambda dev: (t(dev, *shape, dtype=dt), 8, 64, -1, False, 2, torch.float32, 0))
        both("float_qdq", lambda dev: (t(dev, *shape, dtype=dt), 10, 5, 15, True, False, 2, None, 0))
        both("float_qdq_nograd", lambda dev: (t(dev, *shape, dtype=dt), 3, 4, 7, False, False, 2, torch.float32, 0))
        both("fixed_qdq", lambda dev: (t(dev, *shape, dtype=dt), 8, 0, True, True, 2, None, None, None, None, None, 0))
        both("fixed_qdq_nograd", lambda dev: (t(dev, *shape, dtype=dt), 8, 0, True, True, 2, t(dev, max(shape[0], 1), dtype=torch.float32),
                                               torch.zeros(max(shape[0], 1), dtype=torch.int64, device=dev), 0, None, None, 0))
        both("sbfp_qdq", lambda dev: (t(dev, *shape, dtype=dt), 4, 16, 4, 4, 7, True, True, True, -1, None))
        both("mxfp_qdq", lambda dev: (t(dev, *shape, dtype=dt), 3, 4, 32, -1, None))
    both("bfp_qdq_multi", lambda dev: ([t(dev, 8, 64, dtype=dt), t(dev, 16, 128, dtype=dt)], 8, 16, -1, True, 2, None, 0))
    both("float_qdq_multi", lambda dev: ([t(dev, 8, dtype=dt), t(dev, 16, dtype=dt)], 10, 5, 15, True, False, 2, None, 0))
    both("nm_mask", lambda dev: (t(dev, 8, 64, dtype=torch.float32), t(dev, 8, 64, dtype=dt), 2, 4, -1, True, True, None, None))
    both("group_minmax", lambda dev: (t(dev, 256, 64, dtype=dt), 0, 128))
    both("channel_maxabs", lambda dev: (t(dev, 256, 64, dtype=dt), -1))
    both("scale_channels", lambda dev: (t(dev, 256, 64, dtype=dt), t(dev, 64, dtype=torch.float32), -1, True, None))
# the dispatcher-free entry points (PyInit_dmxq_fast in the same shared object): argument conversion and the refusal of CPU tensors
n_fast = 0
F = _backend_torch.FAST
assert F is not None, "the instrumented binding has no direct entry points"
x = torch.randn(4, 64)
f16 = [10, 5, 15, 1]
for name, args in (("bfp_qdq", (x, 8, 16)), ("bfp_qdq", (x, 8, 16, -1, True, 2, torch.float32, 0)), ("float_qdq", (x, 10, 5, 15, True)),
                   ("fixed_qdq", (x, 8, 0, True, True, 2, None, None, None, None)), ("sbfp_qdq", (x, 4, 16, 4, 4, 7, True, True, True)),
                   ("mxfp_qdq", (x, 3, 4, 32)), ("weight_hypernet", (x, 8, 64, True, None, 0, 0, None)), ("input_hypernet", (x, torch.ones(64), 8, 64, True)),
                   ("binary_cast", (x, x, 0, f16, f16, f16)), ("relu_cast", (x, f16, f16)), ("scale_channels", (x, torch.ones(64), -1, True)),
                   ("unary_cast", (x, 0, 0.0, f16, f16)), ("lut16_apply", (x, x)), ("softmax_cast", (x, float("-inf"), f16, f16)),
                   ("norm_cast", (x, 64, None, None, 1e-5, 0, f16, f16))):
    try:
        getattr(F, name)(*args)
    except (RuntimeError, NotImplementedError, TypeError):
        n_fast += 1
    else:
        raise AssertionError(f"dmxq_fast.{name} accepted CPU tensors")
print(f"torch binding driver: {n_meta} meta-kernel calls, {n_cpu} CPU-tensor calls refused through the dispatcher, {n_fast} through the direct entry points, "
      f"on {_backend_torch.TORCH_LIB_PATH}")
