"""Calibration observers — mirror of the parts of the reference's `numerical/observer.py` that sit on the hot
path: MinMaxObserver (per-tensor / per-channel / per-group slabs) and `_calculate_qparams`.

The reference creates one observer nn.Module per group in a Python loop (numerical/cast.py:185-213) and runs two
ATen reductions per group; here ONE `dmxq_group_minmax` launch produces every group's min/max and one
`dmxq_qparams` launch turns them into (scale, zero_point).  HistogramObserver keeps its two passes over the
tensor on the device (min/max + `dmxq_histc`) and its bins-long bookkeeping and scalar search on the host.
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
        self.process_group = None

    def set_process_group(self, process_group):
        """Sharded calibration: every rank observes a SHARD of the tensor and all ranks hold the same set of groups (per tensor; per
        channel along an axis that is not the sharded one) -- after each step the running minima / maxima are completed over the
        group with one all_reduce(MIN) and one all_reduce(MAX) of `[n_groups]` floats (`parallel.WORLD` = the default group; None =
        off).  Groups that live inside one rank's shard (slabs along the sharded axis) need no exchange: leave it off."""
        self.process_group = process_group

    def _exchange(self):
        if self.process_group is not None and self.min_val.dim() == self.max_val.dim() and self.min_val.is_floating_point():
            from . import parallel

            g = parallel.resolve_group(self.process_group)
            parallel.allreduce_min_(self.min_val, g)
            parallel.allreduce_max_(self.max_val, g)

    def forward(self, x, group_size: Optional[int] = None):
        if x.numel() == 0:
            if self.process_group is None:
                return x
            # an empty shard still takes part in the exchange, with the identities in the shape the other ranks hold
            if group_size or self.qscheme in _PER_CHANNEL:
                n_groups = -(-x.shape[self.ch_axis] // (group_size or 1)) if x.dim() else 1
                if self.min_val.dim() == 0 and n_groups > 0:
                    self.min_val = torch.full((n_groups,), float("inf"), device=x.device)
                    self.max_val = torch.full((n_groups,), float("-inf"), device=x.device)
            else:
                self.min_val, self.max_val = self.min_val.to(x.device), self.max_val.to(x.device)
        else:
            self._observe(x, group_size)
        self._exchange()
        return x

    def _observe(self, x, group_size: Optional[int] = None):
        xd = x.detach()
        # the running state has this observation's shape already (every call after the first): ONE launch folds the tensor's extrema
        # into it (dmxq_group_minmax_accumulate) -- reduction and `min_val = torch.min(x_min, min_val)` together, no fill launch
        if group_size or self.qscheme in _PER_CHANNEL:
            view, axis, gs = xd, self.ch_axis, (group_size or 1)
            n_groups = -(-xd.shape[self.ch_axis] // gs) if xd.dim() else 1
            shape = (n_groups,)
        else:
            view, axis, gs, shape = xd.reshape(1, -1), 0, 1, ()
        if (self.min_val.shape == shape and self.min_val.is_cuda and self.min_val.device == xd.device and self.min_val.dtype == torch.float32
                and self.max_val.shape == shape and self.max_val.device == xd.device and self.max_val.dtype == torch.float32
                and self.min_val.is_contiguous() and self.max_val.is_contiguous() and not torch.compiler.is_compiling()):
            ops.group_minmax_accumulate(view, axis, gs, self.min_val.view(-1), self.max_val.view(-1))
            return x
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
    """Running histogram + L2-error range search (observer.py:213-583, itself adapted from torch.ao's observer).

    Split by cost: the two passes over the observed tensor -- its min/max and `torch.histc` -- are device launches
    (`dmxq_group_minmax`, `dmxq_histc`); everything that only touches the `bins`-long histogram (re-binning onto a
    wider range, the quantile walk that picks the clipping range) is fp32 work on a host copy, with the same
    operand types and order as the reference's tensor expressions so that the same bins win.  Per-tensor only.
    Quirk kept: the histogram range is the observed min/max truncated toward zero by `int()`
    (observer.py:470-472, 489-491), so the fractional tails fall outside and are not counted."""

    def __init__(self, bins: int = 2048, upsample_rate: int = 128, dtype: Format = None,
                 qscheme=torch.per_tensor_affine, ch_axis: int = -1, **kw):
        if qscheme in _PER_CHANNEL:
            raise NotImplementedError(
                "HistogramObserver's qscheme only support torch.per_tensor_symmetric or torch.per_tensor_affine.")
        super().__init__(dtype or Format.from_shorthand("XP[8,0](CSN)"), qscheme, ch_axis, **kw)
        self.bins, self.upsample_rate = bins, upsample_rate
        self.register_buffer("histogram", torch.zeros(bins))
        self.register_buffer("min_val", torch.tensor(float("inf")))
        self.register_buffer("max_val", torch.tensor(float("-inf")))
        self._device = None

    # ------------------------------------------------------------------ observation (device passes + host merge)
    def _uninitialised(self):
        return float(self.min_val) == float("inf") and float(self.max_val) == float("-inf")

    def forward(self, x, group_size=None):
        if x.numel() == 0:
            return x
        xd = x.detach()
        self._device = xd.device
        mn, mx = ops.group_minmax(xd.reshape(1, -1), 0, 1)
        new_min, new_max = mn.reshape(()).cpu(), mx.reshape(()).cpu()
        old_min, old_max = self.min_val.cpu(), self.max_val.cpu()
        if self._uninitialised() or float(old_min) == float(old_max):
            hist = ops.histc(xd, self.bins, int(new_min), int(new_max)).cpu()
            lo, hi = new_min, new_max
        else:
            lo, hi = torch.min(new_min, old_min), torch.max(new_max, old_max)
            # common fine grid (observer.py:400-423): the old bins are cut into upsample_rate pieces, the merged range
            # is a whole number (`down`) of old bin widths per new bin; only the upper end is relaxed to make it fit
            fine = (old_max - old_min) / (self.bins * self.upsample_rate)
            down = int(torch.ceil((hi - lo) / (self.bins * fine)).item())
            hi = hi + (down * (self.bins * fine) - (hi - lo))
            start = int(torch.round((old_min - lo) / fine).item())
            hist = ops.histc(xd, self.bins, int(lo), int(hi)).cpu()
            old_hist = self.histogram.cpu()
            if lo == old_min and hi == old_max:
                hist = hist + old_hist
            else:
                hist = self._rebin_onto(hist, old_hist, down, start)
        self.histogram = hist
        self.min_val, self.max_val = lo.clone(), hi.clone()
        return x

    def _rebin_onto(self, hist, old_hist, down, start):
        """observer.py:425-458: spread each old bin uniformly over its fine cells, place them at `start` on the
        merged fine grid, and integrate `down` cells per new bin (running sum in fp64, differences back to fp32)."""
        n, up = self.bins, self.upsample_rate
        cells = torch.zeros(n * down)
        cells[start:n * up + start] = old_hist.repeat_interleave(up)
        upto = torch.cumsum(cells, 0, dtype=torch.double)[down - 1::down]
        before = torch.zeros(n)
        before[1:n] = upto[0:-1]
        return hist + ((upto - before) / up).to(torch.float)

    # ------------------------------------------------------------------ range search (host, bins-long vectors)
    def _clip_error(self, hist, lo: float, hi: float, first: int, last: int) -> float:
        """L2 error of quantising the histogram's mass with 2^precision uniform levels spanning bins
        [first, last], each source bin treated as a uniform density (observer.py:277-329).  Three pieces per source
        bin: from its start to the end of the level it starts in, the whole levels it spans, and from the start of
        the level it ends in to its end; each piece is density * integral of x^2 = density * (b^3 - a^3) / 3."""
        levels = 2 ** self.dtype.precision
        width = (hi - lo) / self.bins
        step = width * (last - first + 1) / levels
        if step == 0.0:
            return 0.0
        cube3 = lambda a, b: (b * b * b - a * a * a) / 3
        begin = (torch.arange(self.bins) - first) * width
        end = begin + width
        lvl_b = torch.clamp(torch.div(begin, step, rounding_mode="floor"), 0, levels - 1)
        lvl_e = torch.clamp(torch.div(end, step, rounding_mode="floor"), 0, levels - 1)
        density = hist / width
        err = torch.zeros(self.bins)
        err += density * cube3(begin - (lvl_b + 0.5) * step, torch.ones(self.bins) * (step / 2))
        err += (lvl_e - lvl_b - 1) * (density * cube3(torch.tensor(-step / 2), torch.tensor(step / 2)))
        err += density * cube3(torch.tensor(-step / 2), end - (lvl_e * step + step / 2))
        return err.sum().item()

    def _search_range(self):
        """observer.py:331-397: shave 1e-5 quantile steps off whichever tail frees more bins while the L2 error
        keeps falling.  The reference walks l / r bin by bin; the cumulative histogram is monotone, so the same
        bins come out of a binary search."""
        hist, min_val, max_val = self.histogram.cpu(), self.min_val.cpu(), self.max_val.cpu()
        assert hist.numel() == self.bins, "bins mistmatch"
        width = (max_val - min_val) / self.bins
        total = torch.sum(hist).item()
        csum = torch.cumsum(hist, dim=0)
        step, lo_q, hi_q = 1e-5, 0.0, 1.0
        first, last, best = 0, self.bins - 1, float("inf")
        while lo_q < hi_q:
            nlo, nhi = lo_q + step, hi_q - step
            # first bin >= `first` whose cumulative count reaches the lower quantile / last bin <= `last` still
            # within the upper quantile (thresholds compared in fp32, like tensor-vs-scalar comparisons)
            l = int(torch.searchsorted(csum, torch.tensor(nlo * total, dtype=csum.dtype), right=False))
            l = min(last, max(first, l))
            r = int(torch.searchsorted(csum, torch.tensor(nhi * total, dtype=csum.dtype), right=True)) - 1
            r = max(first, min(last, r))
            nfirst, nlast = first, last
            if (l - first) > (last - r):
                nfirst, lo_q = l, nlo
            else:
                nlast, hi_q = r, nhi
            if nfirst == first and nlast == last:
                continue
            err = self._clip_error(hist, min_val.item(), max_val.item(), nfirst, nlast)
            if err > best:
                break
            best, first, last = err, nfirst, nlast
        return min_val + width * first, min_val + width * (last + 1)

    def calculate_qparams(self):
        if self.quant_min is None:
            raise ValueError(f"{self.dtype!r} has no integer range: qparams are defined for XP[p,0](C..) formats only")
        if self._uninitialised() or self._device is None:
            return torch.tensor([1.0]), torch.tensor([0])
        assert self.bins == len(self.histogram), \
            "The number of bins in histogram should be equal to the number of bins supplied while making this observer"
        lo, hi = self._search_range()
        return ops.qparams(lo.reshape(1).to(self._device), hi.reshape(1).to(self._device), self.quant_min,
                           self.quant_max, self.qscheme in _SYMMETRIC)

    def extra_repr(self):
        return f"quant_min = {self.quant_min}, quant_max = {self.quant_max}, min_val = {self.min_val}, max_val = {self.max_val}"


class PercentileObserver(HistogramObserver):
    """observer.py:585-634: carries a `percentile` but computes its qparams exactly like HistogramObserver (the
    percentile estimate is unimplemented upstream)."""

    def __init__(self, percentile: float = 99.99, **kw):
        if percentile < 0 or percentile > 100:
            raise ValueError("Invalid percentile. Must be in range 0 <= percentile <= 100.")
        self.percentile = percentile
        super().__init__(**kw)
