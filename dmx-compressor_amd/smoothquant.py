"""SmoothQuant — mirror of the reference's `numerical/smoothquant.py` ActivationWeightSmoothQuant.

scale[c] = clamp( amax|inp[.., c]|^alpha / clamp(amax|w[:, c]|, scale_min)^(1-alpha), scale_min ), cast through the
scale format; the input is divided and the weight multiplied along their channel axes
(smoothquant.py:255-321).  The two per-channel reductions, the scale and the scaling are one kernel launch each
(dmxq_channel_maxabs / dmxq_smoothquant_scale / dmxq_scale_channels) instead of abs + amax + pow + div chains.

Reference quirk kept: the derived class stores `weight_maxabs` / `input_maxabs` but its
"exists" checks read the base class's `b_maxabs` / `a_maxabs` (smoothquant.py:455-473, 527-535), so BOTH maxima
are recomputed on every calibration call and the scale reflects the LAST batch only, not a running maximum.

Sharded calibration (the one real exchange on the path): both maxima are reductions over everything BUT the
channel axis -- the weight's over its rows (smoothquant.py:285-299 with win_ch_axis = -1), the input's over its tokens.
A rank that holds a row shard of the weight (or a token shard of the activations) has the maxima of its shard only;
`set_process_group(group, weight=..., input=...)` makes `forward` finish each flagged maximum with ONE
`all_reduce(MAX)` of the `[C_in]` fp32 vector (RCCL on GPUs; 16 KiB for 4096 channels) before the scale is computed,
so every rank ends with the whole tensor's scale, bit for bit (a maximum does not depend on the order of its operands).
"""
from typing import Union

import torch

from . import ops, parallel
from ._flags import HostFlags
from .cast import CastTo
from .format import Format

__all__ = ["ActivationWeightSmoothQuant"]


class _ScaleChannels(torch.autograd.Function):
    """x / s[c] or x * s[c] along a channel axis as one launch; the scale is a constant (a buffer computed under
    no_grad in the reference too), so the gradient is the incoming one scaled the same way."""

    @staticmethod
    def forward(ctx, x, scale, ch_axis, divide, out_dtype):
        ctx.save_for_backward(scale)
        ctx.ch_axis, ctx.divide, ctx.in_dtype = ch_axis, divide, x.dtype
        return ops.scale_channels(x, scale, ch_axis, divide=divide, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return ops.scale_channels(g.contiguous(), scale, ctx.ch_axis, divide=ctx.divide, out_dtype=ctx.in_dtype), None, None, None, None


def _scale(x, scale, ch_axis, divide, out_dtype):
    if torch.is_grad_enabled() and x.requires_grad:
        return _ScaleChannels.apply(x, scale, ch_axis, divide, out_dtype)
    return ops.scale_channels(x, scale, ch_axis, divide=divide, out_dtype=out_dtype)


class ActivationWeightSmoothQuant(HostFlags, torch.nn.Module):
    _flag_names = ("enabled", "dynamic", "fused_to_weight")
    _scalar_names = ("migration_strength", "scale_min")

    def __init__(self, ch_axis: int, win_ch_axis: int, migration_strength: float = 0.5,
                 scale_format: Union[str, Format] = "SAME", dynamic: bool = False, scale_min: float = 1e-5):
        super().__init__()
        assert 0 <= migration_strength <= 1, "migration strength should be between 0 and 1"
        self.ch_axis, self.win_ch_axis = ch_axis, win_ch_axis
        self.register_buffer("migration_strength", torch.tensor(float(migration_strength)))
        self.register_buffer("scale_min", torch.tensor(float(scale_min)))
        self.register_buffer("scale", torch.ones(1))
        self.register_buffer("enabled", torch.tensor([0], dtype=torch.long))
        self.register_buffer("dynamic", torch.tensor([1 if dynamic else 0], dtype=torch.long))
        self.register_buffer("fused_to_weight", torch.tensor([0], dtype=torch.long))
        self.refresh_flags()
        self.scale_cast = CastTo(format=scale_format)
        self.calibrating = False
        self.input_maxabs = self.weight_maxabs = None
        self.process_group, self._reduce_weight, self._reduce_input = None, False, False

    # -------------------------------------------------------------- sharded calibration
    def set_process_group(self, process_group, weight: bool = True, input: bool = False):
        """`process_group`: a torch.distributed group, `parallel.WORLD` for the default group, or None to switch the exchange off.
        `weight`: this rank's weight is a shard along a NON-channel axis (row-sharded Linear: rows = output features) -- its
        per-input-channel maxima are completed over the group; `input`: likewise for activations sharded over tokens."""
        self.process_group = process_group
        self._reduce_weight, self._reduce_input = bool(weight and process_group is not None), bool(input and process_group is not None)

    # -------------------------------------------------------------- switches
    def enable(self, enabled: bool = True):
        self._set_flag("enabled", enabled)

    def disable(self):
        self.enable(False)

    def set_dynamic(self, dynamic: bool = True):
        if dynamic and self._flag("fused_to_weight"):
            raise RuntimeError("SmoothQuant cannot be dynamic as scale has been fused to weight already")
        self._set_flag("dynamic", dynamic)

    def set_scale_format(self, format: Union[str, Format]):
        self.scale_cast.set_format(format)

    def set_migration_strength(self, migration_strength: float):
        assert 0 <= migration_strength <= 1, "migration strength should be between 0 and 1"
        self._set_scalar("migration_strength", migration_strength)

    # -------------------------------------------------------------- calibration
    def compute_scale(self, inp_maxabs: torch.Tensor, wgt_maxabs: torch.Tensor) -> None:
        # (host mirrors of the two scalar buffers: `float(buffer)` would copy from the device and drain the stream, per Linear per forward)
        s = ops.smoothquant_scale(inp_maxabs, wgt_maxabs, self._scalar("migration_strength"), self._scalar("scale_min"))
        self.scale = self.scale_cast(s)

    def forward(self, inp: torch.Tensor, wgt: torch.Tensor) -> None:
        """smoothquant.py:518-535 (with the quirk above: both maxima from this call only)."""
        with torch.no_grad():
            self.weight_maxabs = ops.channel_maxabs(wgt.detach(), self.win_ch_axis)
            self.input_maxabs = ops.channel_maxabs(inp.detach(), self.ch_axis)
            if self._reduce_weight:   # max over ALL rows per input channel (smoothquant.py:285-299), not this shard's
                parallel.allreduce_max_(self.weight_maxabs, parallel.resolve_group(self.process_group))
            if self._reduce_input:
                parallel.allreduce_max_(self.input_maxabs, parallel.resolve_group(self.process_group))
            self.compute_scale(self.input_maxabs, self.weight_maxabs)

    # -------------------------------------------------------------- application
    def scale_input(self, inp: torch.Tensor) -> torch.Tensor:
        if self._flag("enabled"):
            # reference: a / scale.view(...) -> torch promotion of (input dtype, fp32 scale)
            return _scale(inp, self.scale, self.ch_axis, True, torch.promote_types(inp.dtype, torch.float32))
        return inp

    def scale_weight(self, wgt: torch.Tensor) -> torch.Tensor:
        if self._flag("enabled"):
            return _scale(wgt, self.scale, self.win_ch_axis, False, wgt.dtype)  # .to(wgt.dtype)
        return wgt

    def fuse_to_weight(self, wgt: torch.Tensor) -> None:
        wgt.data = self.scale_weight(wgt.data)
        self._set_flag("fused_to_weight", True)

    def extra_repr(self) -> str:
        return (f"migration_strength = {self._scalar('migration_strength')}, ch_axis = {self.ch_axis}, win_ch_axis = "
                f"{self.win_ch_axis}, scale_format = {self.scale_cast.format}, dynamic = {bool(self.dynamic.item())}")
