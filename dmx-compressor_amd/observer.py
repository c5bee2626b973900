"""Calibration observers — mirror of the parts of the reference's `numerical/observer.py` that sit on the hot
path: MinMaxObserver (per-tensor / per-channel / per-group slabs) and `_calculate_qparams`.

The reference creates one observer nn.Module per group in a Python loop (numerical/cast.py:185-213) and runs two
ATen reductions per group; here ONE `dmxq_group_minmax` launch produces every group's min/max and one
`dmxq_qparams` launch turns them into (scale, zero_point).  HistogramObserver's scalar search is host-side
work outside the hot path and is not provided.
"""
from typing import Optional, Tuple

import torch

from . import ops
from .format import FixedPoint, Format


def get_qmin_qmax(fmt: Format) -> Tuple[Optional[int], Optional[int]]:
    """observer.py:13-21: integer range of a clamped, fraction-free fixed point format."""
    if isinstance(fmt, FixedPoint) and fmt.fraction == 0 and fmt.clamp:
        qmin, qmax = -(2 ** (fmt.precision - 1)), 2 ** (fmt.precision - 1) - 1
        if fmt.symmetric:
            qmin += 1
        return qmin, qmax
    return None, None


_SYMMETRIC = (torch.per_tensor_symmetric, torch.per_channel_symmetric)
_PER_CHANNEL = (torch.per_channel_affine, torch.per_channel_symmetric, torch.per_channel_affine_float_qparams)


class ObserverBase(torch.nn.Module):
    def __init__(self, dtype: Format, qscheme=torch.per_tensor_affine, ch_axis: int = -1, **_):
        super().__init__()
        assert isinstance(dtype, Format), f"illegal format {dtype}"
        self.dtype, self.qscheme, self.ch_axis = dtype, qscheme, ch_axis
        self.quant_min, self.quant_max = get_qmin_qmax(dtype)

    def calculate_qparams(self):
        raise NotImplementedError


class DummyObserver(ObserverBase):
    """observer.py:118-136: observes nothing; qparams are the defaults (scale 1, zero point 0)."""

    def forward(self, x, group_size=None):
        return x

    def calculate_qparams(self):
        return torch.tensor([1.0]), torch.tensor([0])


class MinMaxObserver(ObserverBase):
    """Running min/max (observer.py:139-211); one object covers all groups of a group-quantised tensor."""

    def __init__(self, dtype: Format = None, qscheme=torch.per_tensor_affine, ch_axis: int = -1, **kw):
        super().__init__(dtype or Format.from_shorthand("XP[8,0](CSN)"), qscheme, ch_axis, **kw)
        if qscheme == torch.per_channel_affine_float_qparams:
            raise NotImplementedError("MinMaxObserver does not support qscheme: torch.per_channel_affine_float_qparams")
        self.register_buffer("min_val", torch.tensor(float("inf")))
        self.register_buffer("max_val", torch.tensor(float("-inf")))

    def forward(self, x, group_size: Optional[int] = None):
        if x.numel() == 0:
            return x
        xd = x.detach()
        if group_size:                       # slabs along ch_axis (cast.py:200-204 torch.split)
            mn, mx = ops.group_minmax(xd, self.ch_axis, group_size)
        elif self.qscheme in _PER_CHANNEL:   # one group per channel
            mn, mx = ops.group_minmax(xd, self.ch_axis, 1)
        else:                                # whole tensor: a single group
            mn, mx = ops.group_minmax(xd.reshape(1, -1), 0, 1)
            mn, mx = mn.reshape(()), mx.reshape(())
        if self.min_val.shape == mn.shape and self.min_val.device == mn.device:
            mn, mx = torch.minimum(mn, self.min_val), torch.maximum(mx, self.max_val)
        elif self.min_val.dim() == 0 and mn.dim() > 0:  # first group/channel observation
            pass
        else:
            mn = torch.minimum(mn, self.min_val.to(mn.device))
            mx = torch.maximum(mx, self.max_val.to(mx.device))
        self.min_val, self.max_val = mn, mx
        return x

    def calculate_qparams(self):
        if self.quant_min is None:
            raise ValueError(f"{self.dtype!r} has no integer range: qparams are defined for XP[p,0](C..) formats only")
        mn, mx = self.min_val.reshape(-1), self.max_val.reshape(-1)
        if not mn.is_cuda:  # nothing observed yet (observer.py:66-69)
            return torch.tensor([1.0]), torch.tensor([0])
        return ops.qparams(mn, mx, self.quant_min, self.quant_max, self.qscheme in _SYMMETRIC)

    def reset_min_max_vals(self):
        self.min_val = torch.tensor(float("inf"))
        self.max_val = torch.tensor(float("-inf"))

    def extra_repr(self):
        return f"quant_min = {self.quant_min}, quant_max = {self.quant_max}, min_val = {self.min_val}, max_val = {self.max_val}"


class HistogramObserver(ObserverBase):
    def __init__(self, *a, **k):
        raise NotImplementedError(
            "HistogramObserver: the histogram/scalar search is host-side calibration outside the accelerated hot "
            "path (SURVEY.md §2 row 6); use MinMaxObserver")
