"""CastTo / CastToDict / CastToFormat — mirror of the reference's `numerical/cast.py` op containers.

`CastTo.forward` keeps the reference's order of operations (cast.py:261-306: remember physical dtype ->
pre_transform {shaping, noquant_shortcut, format} -> observer step -> [affine] -> format cast -> [inverse
affine] -> shortcut restore -> inverse shaping -> `.to(physical_dtype)`), but the arithmetic part
(`.float()`, `x/sc + zp`, the format cast, `(x - zp)*sc`, the final narrowing) is ONE kernel launch with the
input and output in the tensor's own dtype, instead of 5-7 elementwise ATen passes around a chunked native call.
"""
import warnings
from typing import Dict, Optional, Union

import torch
from torch.autograd import Function

from . import ops
from ._flags import HostFlags
from .format import FixedPoint, Format, Same
from .observer import _PER_CHANNEL, DummyObserver, HistogramObserver, MinMaxObserver, ObserverBase

__all__ = ["CastToFormat", "CastTo", "CastToDict"]


class CastToFormat(Function):
    """Numerical cast with a straight-through-estimator backward (cast.py:20-32)."""

    @staticmethod
    def forward(ctx, x, fmt, block_dim, out_dtype=torch.float32):
        ctx.set_materialize_grads(False)
        ctx.in_dtype = x.dtype
        return fmt.cast(x, block_dim, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, g):
        if g is not None and g.dtype != ctx.in_dtype:
            g = g.to(ctx.in_dtype)
        return g, None, None, None


class _FixedAffineCast(Function):
    """x/sc + zp -> FixedPoint cast -> (x - zp)*sc as one launch (cast.py:278-296), STE backward."""

    @staticmethod
    def forward(ctx, x, fmt, scale, zero_point, ch_axis, group_size, out_dtype):
        ctx.set_materialize_grads(False)
        ctx.in_dtype = x.dtype
        return ops.fixed_qdq(x, fmt.precision, fmt.fraction, fmt.clamp, fmt.symmetric, fmt.rounding, scale=scale,
                             zero_point=zero_point, ch_axis=ch_axis, group_size=group_size, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, g):
        if g is not None and g.dtype != ctx.in_dtype:
            g = g.to(ctx.in_dtype)
        return g, None, None, None, None, None, None


class CastTo(HostFlags, torch.nn.Module):
    """Simulated numerical cast to a target format (cast.py:136-162).  Buffers and switches follow
    torch.ao's FakeQuantize, which the reference subclasses: scale, zero_point, fake_quant_enabled,
    observer_enabled, qscheme, ch_axis.  The two switches are read from host mirrors (_flags.py): forward never
    waits for the device."""
    _flag_names = ("fake_quant_enabled", "observer_enabled")
    #: a SAME-format cast returns a COPY, like the reference's Same.cast (`x.clone()`, numerical/format.py:89-90): the
    #: caller may mutate the result in place.  DmxModule switches this off on the casts it owns (their results are
    #: consumed inside the module) and re-establishes the no-alias guarantee once, at the module boundary (nn.py).
    copy_on_same = True
    #: cast a transposed / permuted view IN ITS OWN LAYOUT: the kernels take contiguous tensors, so a dense permuted view is cast through
    #: the permutation that makes it contiguous (the block dimension follows it) and the result is handed back with the input's strides,
    #: instead of being copied to a contiguous tensor first.  Same values; set by modules whose `_forward` takes any strides
    #: (ActActMatMul: q / k^T / v of an attention arrive as transposed views -- k^T blocked along -2 is k blocked along its LAST dim).
    keep_layout = False

    def __init__(self, format="SAME", observer=DummyObserver, group_size=None, block_dim=-1,
                 qscheme=torch.per_tensor_affine, ch_axis=-1, **observer_kwargs):
        super().__init__()
        self.set_format(format)
        self.qscheme, self.ch_axis = qscheme, ch_axis
        self.is_per_channel = qscheme in _PER_CHANNEL
        if group_size:
            assert not self.is_per_channel, "group_size must be used with per tensor quantization scheme"
        self.group_size = group_size if group_size else None
        self.activation_post_process = observer(dtype=self.format, qscheme=qscheme, ch_axis=ch_axis, **observer_kwargs)
        self.register_buffer("scale", torch.tensor([1.0], dtype=torch.float))
        self.register_buffer("zero_point", torch.tensor([0], dtype=torch.int64))
        self.register_buffer("fake_quant_enabled", torch.tensor([1], dtype=torch.uint8))
        self.register_buffer("observer_enabled", torch.tensor([0], dtype=torch.uint8))
        self.refresh_flags()
        self.physical_dtype = None
        self.block_dim = block_dim
        self.pre_transform = {}

    # ------------------------------------------------------------------ switches (FakeQuantize API)
    def enable_fake_quant(self, enabled: bool = True):
        self._set_flag("fake_quant_enabled", enabled)

    def disable_fake_quant(self):
        self.enable_fake_quant(False)

    def enable_observer(self, enabled: bool = True):
        self._set_flag("observer_enabled", enabled)

    def disable_observer(self):
        self.enable_observer(False)

    def calculate_qparams(self):
        return self.activation_post_process.calculate_qparams()

    # ------------------------------------------------------------------ configuration
    def set_format(self, format: Union[str, torch.dtype, Format]):
        if isinstance(format, str):
            format = Format.from_shorthand(format)
        self.format = format
        if hasattr(self, "activation_post_process"):
            self.activation_post_process.dtype = format
            from .observer import get_qmin_qmax
            self.activation_post_process.quant_min, self.activation_post_process.quant_max = get_qmin_qmax(format)

    @property
    def dtype(self):
        return self.format

    def set_pre_transform(self, pre_transform: Dict):
        self.pre_transform = dict(pre_transform)
        if isinstance(self.pre_transform.get("format"), str):
            self.pre_transform["format"] = Format.from_shorthand(self.pre_transform["format"])

    def enable_calibration(self, state: bool = True, observer_cls: ObserverBase = HistogramObserver,
                           qscheme_to_overload: Optional[torch.qscheme] = None, group_size: int = None,
                           ch_axis: int = None) -> None:
        """cast.py:308-340: install an observer and switch to observe-only, or back to fake-quant."""
        if state:
            if ch_axis is not None:
                self.ch_axis = ch_axis
            if qscheme_to_overload is not None:
                self.qscheme = qscheme_to_overload
                self.is_per_channel = qscheme_to_overload in _PER_CHANNEL
            self.group_size = group_size if group_size else None
            if self.group_size:
                assert not self.is_per_channel, "group quantization is to be used with per tensor quantization"
            self.activation_post_process = observer_cls(dtype=self.format, qscheme=self.qscheme, ch_axis=self.ch_axis)
            self._group_observers = []
            self.disable_fake_quant()
            self.enable_observer()
        else:
            self.enable_fake_quant()
            self.disable_observer()

    # ------------------------------------------------------------------ forward pieces
    def _observer_step(self, x):
        """cast.py:179-226, all groups in one reduction launch."""
        obs = self.activation_post_process
        obs.ch_axis = self.ch_axis
        obs.qscheme = self.qscheme
        if self.group_size and not isinstance(obs, (MinMaxObserver, DummyObserver)):
            # per-tensor-only observers (histogram): one instance per slab, as cast.py:185-213 does for every class
            slabs = torch.split(x.detach(), self.group_size, dim=self.ch_axis)
            if len(getattr(self, "_group_observers", ())) != len(slabs):
                self._group_observers = [obs.__class__(dtype=self.format, qscheme=self.qscheme, ch_axis=self.ch_axis)
                                         for _ in slabs]
            qp = []
            for o, slab in zip(self._group_observers, slabs):
                o(slab)
                qp.append(o.calculate_qparams())
            _scale = torch.cat([s.reshape(1) for s, _ in qp])
            _zero_point = torch.cat([z.reshape(1) for _, z in qp])
            obs.min_val = torch.stack([o.min_val.reshape(()) for o in self._group_observers])
            obs.max_val = torch.stack([o.max_val.reshape(()) for o in self._group_observers])
        else:
            obs(x.detach(), self.group_size) if isinstance(obs, MinMaxObserver) else obs(x.detach())
            _scale, _zero_point = obs.calculate_qparams()
        _scale, _zero_point = _scale.to(x.device), _zero_point.to(x.device)
        if self.scale.shape != _scale.shape or self.scale.device != _scale.device:
            self.scale = torch.zeros_like(_scale)
            self.zero_point = torch.zeros_like(_zero_point)
        self.scale.copy_(_scale)
        self.zero_point.copy_(_zero_point)

    @staticmethod
    def apply_shaping_seq(x, shaping_list):
        """cast.py:239-259: view / permute / flatten sequence and its inverse."""
        inverse = []
        for op, args in shaping_list:
            orig = x.size()
            if op == "view":
                x = x.reshape(*args)
                inverse.append(("view", orig))
            elif op == "permute":
                x = x.permute(*args)
                inverse.append(("permute", torch.LongTensor(list(args)).argsort().tolist()))
            elif op == "flatten":
                x = x.flatten(*args)
                inverse.append(("view", orig))
            else:
                raise Exception(f"unknown shape op {op}")
        return x, inverse[::-1]

    def _quantize(self, x, out_dtype):
        fmt = self.format
        # the autograd wrappers only matter when a gradient will flow (straight-through estimator); inference skips them
        ste = torch.is_grad_enabled() and x.requires_grad
        if isinstance(fmt, FixedPoint):
            # per-tensor: one scale; per-channel: scale[c]; per-group: scale[c // group_size] (cast.py:279-293)
            if self.group_size:
                ch_axis, gs = self.ch_axis, self.group_size
            elif self.is_per_channel:
                ch_axis, gs = self.ch_axis, None
            else:
                ch_axis, gs = None, None
            if ste:
                return _FixedAffineCast.apply(x, fmt, self.scale, self.zero_point, ch_axis, gs, out_dtype)
            return ops.fixed_qdq(x, fmt.precision, fmt.fraction, fmt.clamp, fmt.symmetric, fmt.rounding, scale=self.scale,
                                 zero_point=self.zero_point, ch_axis=ch_axis, group_size=gs, out_dtype=out_dtype)
        if ste:
            return CastToFormat.apply(x, fmt, self.block_dim, out_dtype)
        if self.keep_layout and x.dim() > 1 and not x.is_contiguous() and not torch.compiler.is_compiling():
            order = sorted(range(x.dim()), key=lambda d: (-x.stride(d), d))
            xp = x.permute(order)
            if xp.is_contiguous():
                y = fmt.cast(xp, order.index(self.block_dim % x.dim()), out_dtype=out_dtype)
                inv = [0] * len(order)
                for i, d in enumerate(order):
                    inv[d] = i
                return y.permute(inv)
        return fmt.cast(x, self.block_dim, out_dtype=out_dtype)

    def forward(self, x):
        d = self.__dict__  # (plain attributes: nn.Module.__setattr__ costs several microseconds per assignment)
        d["physical_dtype"] = x.dtype
        fmt, pt = self.format, self.pre_transform
        if not pt:  # the common case: no shaping / shortcut / pre-format
            same = isinstance(fmt, Same)
            if d["_h_observer_enabled"] and not same:
                self._observer_step(x)
            if not d["_h_fake_quant_enabled"]:
                return x
            if same:  # the reference's Same.cast is x.clone() (format.py:89-90): one full copy per no-op cast
                return x.clone() if self.copy_on_same else x
            if not isinstance(fmt, Format):
                raise TypeError("CastTo with a torch.dtype format is torch.ao's stock FakeQuantize path, "
                                "not part of the accelerated hot path")
            return self._quantize(x, x.dtype)
        inverse_shaping = None
        shortcut = None
        if "shaping" in pt:
            x, inverse_shaping = self.apply_shaping_seq(x, pt["shaping"])
        sc_idx = pt.get("noquant_shortcut")
        if isinstance(sc_idx, list):  # the reference indexes with the list itself (a multi-dim index); spelled as a tuple
            sc_idx = tuple(sc_idx)
        if sc_idx is not None:
            shortcut = x[sc_idx].clone()
        if "format" in pt:
            x = CastToFormat.apply(x, pt["format"], self.block_dim, torch.float32)
        if d["_h_observer_enabled"] and x is not None and not isinstance(fmt, Same):
            self._observer_step(x)
        if d["_h_fake_quant_enabled"]:
            if isinstance(fmt, Format):
                x = self._quantize(x, self.physical_dtype)
            else:
                raise TypeError("CastTo with a torch.dtype format is torch.ao's stock FakeQuantize path, "
                                "not part of the accelerated hot path")
        if shortcut is not None:
            x = x.clone() if x.dtype == self.physical_dtype else x.to(self.physical_dtype)
            x[sc_idx] = shortcut.to(x.dtype)
        if inverse_shaping is not None:
            x, _ = self.apply_shaping_seq(x, inverse_shaping)
        return x.to(self.physical_dtype)

    # ------------------------------------------------------------------ introspection
    def get_precision(self) -> Optional[float]:
        if isinstance(self.format, Same):
            if self.physical_dtype is not None:
                return float(torch.finfo(self.physical_dtype).bits)
            return None
        return self.format.bit_precision

    def extra_repr(self):
        return f"format = dtype = {self.format!r}, qscheme = {self.qscheme}, ch_axis = {self.ch_axis}, " \
               f"group_size = {self.group_size}, block_dim = {self.block_dim}"


class CastToDict(torch.nn.ModuleDict):
    """Keyed collection of CastTo's applied to a module's positional / keyword tensors (cast.py:58-134)."""

    def forward(self, x, *args, output=False, first_done=False, **kwargs):
        """first_done: `x` already went through its cast (the fused input path of DmxModule.forward)."""
        keys = list(self.keys())
        if output:
            if isinstance(x, (tuple, list)):
                return type(x)(self[keys[i]](a) for i, a in enumerate(x))
            return self[keys[0]](x)
        i = 1
        new_args, new_kwargs = [], {}
        for a in args:
            if isinstance(a, torch.Tensor):
                new_args.append(self[keys[i]](a))
                i += 1
            else:
                new_args.append(a)
        for k, v in kwargs.items():
            new_kwargs[k] = self[k + "_cast"](v) if isinstance(v, torch.Tensor) else v
        return (x if first_done else self[keys[0]](x)), new_args, new_kwargs

    def pack_to_dict(self, param):
        keys = list(self.keys())
        if isinstance(param, (tuple, list)):
            param = {keys[i]: (p if p is not None else "SAME") for i, p in enumerate(param)}
        elif not isinstance(param, dict):
            raise ValueError("format needs to be a dict, tuple or list!")
        if len(param) != len(self):
            warnings.warn(f"length of format to set is not equal to length of input_casts, some CastTos might not "
                          f"be set properly!\nlen({param}!={len(self)})")
        return param

    def set_pre_transform(self, pre_transforms):
        for k, t in self.pack_to_dict(pre_transforms).items():
            self[k].set_pre_transform(t)

    def set_format(self, format):
        for k, f in self.pack_to_dict(format).items():
            if k not in self.keys():
                raise RuntimeError(f"No CastTo with key {k}!")
            self[k].set_format(f)

    def disable_fake_quant(self):
        for c in self.values():
            c.disable_fake_quant()

    def enable_fake_quant(self):
        for c in self.values():
            c.enable_fake_quant()

    def enable_observer(self):
        for c in self.values():
            c.enable_observer()

    def disable_observer(self):
        for c in self.values():
            c.disable_observer()
