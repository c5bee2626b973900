"""DmxModule mirror — the Python CALLER of the hot path (reference: modeling/nn/core.py, torch_modules.py).

Only what the parity harness of SURVEY.md §8 (rows a11 / a13) needs: the base class with the reference's order
of operations, and the module types BASELINE.json's configs touch (Linear, Conv1d/2d, ResAdd, ActActMatMul,
Softmax, LayerNorm, GELU, ReLU, MaxPool2d, Embedding).  The dense math stays `torch.nn.functional` (rocBLAS /
MIOpen — not ours); every cast / mask / scale around it is a libdmxq kernel.

Order of operations reproduced (core.py:178-264):
  weight hypernet : weight_sparsifier -> smoothquant.scale_weight -> weight_storage_cast -> weight_cast   (:184-196)
  forward         : smoothquant.scale_input -> input_casts -> _forward(_weight, _bias) -> output_casts -> .to(input dtype)
  fold            : bias cast, effective weight, smoothquant fuse, storage cast, weight cast become permanent (:146-176)
"""
import re
from collections import OrderedDict
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import torch.nn.functional as F

from .approximate import Approximate, NoApproximation
from .cast import CastTo, CastToDict
from ._flags import FastAttr
from .format import Same
from .smoothquant import ActivationWeightSmoothQuant
from .sparse import Dense, Sparsify

__all__ = ["DmxModule", "DmxQuantizerCalibrationHyperparams", "DmxModuleQuantizerCalibrationHyperparams",
           "DmxModuleSmoothQuantHyperparams", "Linear", "Conv1d", "Conv2d", "ResAdd", "ActActMatMul", "Softmax", "LayerNorm", "GELU", "ReLU", "SiLU", "QuickGELU", "Exp", "Mul", "RMSNorm", "ApplyRotaryPosEmb",
           "MaxPool2d", "AvgPool2d", "Embedding", "ReLU6", "Tanh", "Dropout", "NewGELU", "FastGELU", "BloomGELU", "ClippedGELU", "AdaptiveAvgPool2d",
           "BatchNorm2d", "GroupNorm", "ConvTranspose2d", "BAddBMM", "ScaledDotProductAttention", "DmxConfigRule", "configure_model",
           "fold_weights_and_biases", "GraphedForward", "LiveWeightBatch", "link_consumer", "link_consumers_from_fx", "DmxTracer"]


class _LazySparsify(Sparsify):
    """score materialised at first use with the weight's shape (reference: sparse.py:323-344 LazySparsify)"""

    def __init__(self, sparseness="DENSE", backward_mode="STE", score_func=None):
        super().__init__(torch.Size([0]), sparseness, backward_mode, score_func)

    def forward(self, x):
        if not isinstance(self.sparseness, Dense) and self.score.shape != x.shape:
            self.score = torch.nn.Parameter(torch.rand(x.shape, device=x.device), requires_grad=True)
        return super().forward(x)


@dataclass
class DmxQuantizerCalibrationHyperparams:
    """advanced_recipe.py:42-51: how one CastTo is calibrated."""
    observer_cls: Any = None
    qscheme_to_overload: Any = torch.per_tensor_symmetric
    group_size: Optional[int] = None
    ch_axis: Optional[int] = None

    def __post_init__(self):
        if self.observer_cls is None:
            from .observer import HistogramObserver
            self.observer_cls = HistogramObserver


@dataclass
class DmxModuleQuantizerCalibrationHyperparams:
    """advanced_recipe.py:54-63"""
    inputs: Optional[Dict[str, DmxQuantizerCalibrationHyperparams]] = None
    outputs: Optional[Dict[str, DmxQuantizerCalibrationHyperparams]] = None
    weight: Optional[DmxQuantizerCalibrationHyperparams] = None
    weight_storage: Optional[DmxQuantizerCalibrationHyperparams] = None


@dataclass
class DmxModuleSmoothQuantHyperparams:
    """advanced_recipe.py:66-73"""
    migration_strength: float = 0.5
    fuse_to_weight: bool = False


def _shares_storage(a, b) -> bool:
    if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)):
        return False
    if torch.compiler.is_compiling():
        # storages have no addresses while tracing with fake tensors: identity, and the view relation (`_base`) -- an all-SAME module
        # whose `_forward` hands back a VIEW of its input or parameter (x.view(...), x[..., :k], .t()) must still be cloned at the
        # boundary like the reference's Same.cast does
        if a is b:
            return True
        ba, bb = getattr(a, "_base", None), getattr(b, "_base", None)
        return (ba is not None and (ba is b or ba is bb)) or (bb is not None and bb is a)
    return a.device == b.device and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.numel() > 0


def _aliases_any(out, inputs, module) -> bool:
    if not isinstance(out, torch.Tensor):
        return False
    for t in inputs:
        if out is t or _shares_storage(out, t):
            return True
    for p in module._parameters.values():
        if p is not None and _shares_storage(out, p):
            return True
    return False


class DmxModule(FastAttr, torch.nn.Module):
    """Mixin base: call `_dmx_init()` after the torch module's own __init__ (see the concrete classes)."""

    functional_forward = None
    ch_axis = win_ch_axis = wout_ch_axis = None
    has_accum = False

    def _dmx_init(self, n_inputs: int = 1, input_names=("input_cast",), sparsifiable: bool = False):
        pnames = [n for n, _ in self.named_parameters(recurse=False)]
        self.align_boundary_dtype = True
        self.input_casts = CastToDict(OrderedDict((n, CastTo(ch_axis=self.ch_axis if self.ch_axis is not None else -1))
                                                  for n in input_names))
        self.output_casts = CastToDict(OrderedDict({"output_cast": CastTo()}))
        self.accum_cast = CastTo() if self.has_accum else None
        wax = self.wout_ch_axis if self.wout_ch_axis is not None else -1
        self.weight_storage_cast = CastTo(ch_axis=wax) if "weight" in pnames else None
        self.weight_cast = CastTo(ch_axis=wax) if "weight" in pnames else None
        self.bias_cast = CastTo() if "bias" in pnames else None
        self.smoothquant = (ActivationWeightSmoothQuant(self.ch_axis, self.win_ch_axis)
                            if self.ch_axis is not None and self.win_ch_axis is not None else None)
        self.weight_sparsifier = _LazySparsify() if sparsifiable else None
        self.approximator = Approximate()
        self.approximation_error = None
        self._mark_internal_casts()

    def _mark_internal_casts(self):
        """The casts a DmxModule owns are internal: what they return is consumed by `_forward` (read-only) or handed to
        the next internal stage, so a SAME-format cast does not need the reference's defensive `x.clone()`
        (numerical/format.py:89-90) at every site.  The guarantee that clone gives the CALLER -- the module's result
        never aliases its inputs or parameters -- is re-established once in `forward` / `_weight` / `_bias`."""
        for c in self.modules():
            if isinstance(c, CastTo):
                c.copy_on_same = False

    # ------------------------------------------------------------------ configuration (core.py:65-108)
    def configure(self, config) -> None:
        if "input_formats" in config:
            self.input_casts.set_format(config["input_formats"])
        if "pre_input_transform" in config:
            self.input_casts.set_pre_transform(config["pre_input_transform"])
        if "output_formats" in config:
            self.output_casts.set_format(config["output_formats"])
        if "pre_output_transform" in config:
            self.output_casts.set_pre_transform(config["pre_output_transform"])
        if self.accum_cast is not None and "accum_format" in config:
            self.accum_cast.set_format(config["accum_format"])
        if self.weight_storage_cast is not None and "weight_storage_format" in config:
            self.weight_storage_cast.set_format(config["weight_storage_format"])
        if self.weight_cast is not None and "weight_format" in config:
            self.weight_cast.set_format(config["weight_format"])
        if self.weight_cast is not None and "pre_weight_transform" in config:
            self.weight_cast.set_pre_transform(config["pre_weight_transform"])
        if self.bias_cast is not None and "bias_format" in config:
            self.bias_cast.set_format(config["bias_format"])
        if self.smoothquant is not None and "smoothquant_scale_format" in config:
            self.smoothquant.set_scale_format(config["smoothquant_scale_format"])
        if self.weight_sparsifier is not None and "weight_sparseness" in config:
            self.weight_sparsifier.configure(sparseness=config["weight_sparseness"])
        if "approximation_function" in config:
            self.approximator.set_function(config["approximation_function"])

    transform = configure

    # ------------------------------------------------------------------ views used by the harness
    @property
    def input_formats(self):
        return [c.format for c in self.input_casts.values()]

    @property
    def output_formats(self):
        return [c.format for c in self.output_casts.values()]

    @property
    def accum_format(self):
        return self.accum_cast.format if self.accum_cast is not None else None

    @property
    def weight_format(self):
        return self.weight_cast.format if self.weight_cast is not None else None

    @property
    def bias_format(self):
        return self.bias_cast.format if self.bias_cast is not None else None

    @property
    def weight_sparseness(self):
        return self.weight_sparsifier.sparseness if self.weight_sparsifier is not None else None

    # ------------------------------------------------------------------ weight path (core.py:178-213)
    #: use the single-kernel weight path (csrc/hypernet.hip) when the configuration allows it; results are bit-identical
    fuse_weight_hypernet = True

    def _fused_weight_plan(self, _w):
        """The arguments of the single-kernel weight path for this module's configuration, or None when it must take the chain:
        inference only (no autograd through the fused op), mask groups, SmoothQuant channels and BFP blocks all along ONE dim -- the
        last (Linear layout: the tiled kernel) or any other (Conv1d / Conv2d weights along in-channels, torch_modules.py:582-585,
        674-677: dmxq_weight_hypernet_strided) --, BlockTopK or Dense sparseness, SAME storage format, plain BFP weight format with
        nearest rounding.  -> (fmt, block_dim, last, score, K, M, sq_scale)"""
        from .format import BlockFloatingPoint
        from .sparse import BlockTopK
        if not self.fuse_weight_hypernet or self.weight_cast is None or _w.dim() == 0 or torch.is_grad_enabled() and _w.requires_grad:
            return None
        wc, st = self.weight_cast, self.weight_storage_cast
        fmt = wc.format
        nd = _w.dim()
        if (not isinstance(fmt, BlockFloatingPoint) or fmt.rounding != "nearest" or wc.pre_transform or not -nd <= wc.block_dim < nd
                or not wc._flag("fake_quant_enabled") or wc._flag("observer_enabled")):
            return None
        bd = wc.block_dim % nd
        last = bd == nd - 1
        if fmt.block_size < (8 if last else 2):
            return None
        if st is not None and not (isinstance(st.format, Same) and not st.pre_transform):
            return None
        sp, score, K, M = self.weight_sparsifier, None, 0, 0
        if sp is not None and not isinstance(sp.sparseness, Dense):
            if not isinstance(sp.sparseness, BlockTopK) or not -nd <= sp.sparseness.block_dim < nd or sp.sparseness.block_dim % nd != bd \
                    or sp.plastic or sp.score.shape != _w.shape or self.training:
                return None
            score, K, M = sp.score.detach(), sp.sparseness.K, sp.sparseness.block_size
        sq = None
        if self.smoothquant is not None and not self.smoothquant._flag("fused_to_weight") and self.smoothquant._flag("enabled"):
            if not -nd <= self.smoothquant.win_ch_axis < nd or self.smoothquant.win_ch_axis % nd != bd:
                return None
            sq = self.smoothquant.scale
        if not last and score is None and sq is None:
            return None  # a plain cast along another dim: the chain IS one launch (dmxq_bfp_qdq: the column / sub-slab kernels)
        return fmt, bd, last, score, K, M, sq

    def _fused_weight(self, _w, out_dtype=None):
        """mask -> SmoothQuant scale -> BFP in ONE launch, or None when this configuration must take the chain (_fused_weight_plan)"""
        plan = self._fused_weight_plan(_w)
        if plan is None:
            return None
        fmt, bd, last, score, K, M, sq = plan
        from . import ops
        if out_dtype is not None and last:
            # the tiled kernel is instantiated for these (weight, score, output) dtypes (csrc/hypernet.hip); anything else keeps the chain's
            # own dtype and the consumer converts.  Decided HERE, statically: under torch.compile the op is traced through its meta
            # kernel, which cannot refuse a combination.
            sd = score.dtype if score is not None else _w.dtype
            ok = {(torch.bfloat16, torch.float32): (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16): (torch.bfloat16,),
                  (torch.float16, torch.float32): (torch.float16, torch.float32), (torch.float16, torch.float16): (torch.float16,),
                  (torch.float32, torch.float32): (torch.float32,)}.get((_w.dtype, sd), ())
            if out_dtype not in ok:
                out_dtype = None
        y = ops.weight_hypernet(_w.detach(), fmt.precision, fmt.block_size, fmt.symmetric, score, K, M, sq, out_dtype=out_dtype, block_dim=bd)
        if y is not None and self.weight_sparsifier is not None and M:
            self.weight_sparsifier.mask = None  # not materialised on the fused path
        return y

    @property
    def weight_hypernet(self):
        def _weight_hypernet(_w, out_dtype=None):
            """out_dtype: the dtype the CONSUMER converts the result to anyway (Linear: `_weight.to(_input.dtype)`, torch_modules.py:
            347-350) -- the fused kernel then rounds to it directly (the chain's dtype is torch's promotion of weight and score dtypes:
            float32 for a bf16 weight with an N:M score Parameter, i.e. a 4 B/element store and a separate conversion pass)"""
            if _w is self.weight:
                lw = self._batched_live_weight(out_dtype)
                if lw is not None:
                    return lw
            fused = self._fused_weight(_w, out_dtype)
            if fused is not None:
                return fused
            if self.weight_sparsifier is not None:
                _w = self.weight_sparsifier(_w)
            if self.smoothquant is not None and not self.smoothquant._flag("fused_to_weight"):
                _w = self.smoothquant.scale_weight(_w)
            if self.weight_storage_cast is not None:
                _w = self.weight_storage_cast(_w)
            if self.weight_cast is not None:
                _w = self.weight_cast(_w)
            return _w

        return _weight_hypernet

    def _batched_live_weight(self, out_dtype=None):
        """this forward's quantised weight when a `LiveWeightBatch` computed it together with its siblings' (one launch for the set), else
        None.  Valid only for the Parameter state it was computed from (same storage, same version) and in inference."""
        lw = self.__dict__.get("_live_weight")
        if lw is None:
            return None
        w, (o, ver, ptr, natural) = self.weight, lw
        if ver != w._version or ptr != w.data_ptr() or (torch.is_grad_enabled() and w.requires_grad) or torch.compiler.is_compiling():
            self.__dict__.pop("_live_weight", None)
            return None
        # `natural`: the dtype the module's own chain returns when nobody asks for one (torch's promotion of weight and score dtypes for
        # a masked weight); the batch rounds straight to the weight's dtype -- what Linear asks for -- and serves only that request
        return o if o.dtype == (out_dtype if out_dtype is not None else natural) else None

    @property
    def _weight_ro(self):
        """the weight as `_forward` consumes it: READ-ONLY, may be the Parameter itself when every stage is a no-op"""
        lw = self._batched_live_weight()
        return lw if lw is not None else self.weight_hypernet(self.weight)

    @property
    def _bias_ro(self):
        if self.bias_cast is None or self.bias is None:
            return None
        lb = self.__dict__.get("_live_bias")   # (LiveWeightBatch: this forward's bias cast, done together with the siblings')
        if lb is not None:
            b, (o, ver, ptr) = self.bias, lb
            if ver == b._version and ptr == b.data_ptr() and not (torch.is_grad_enabled() and b.requires_grad) and not torch.compiler.is_compiling():
                return o
            self.__dict__.pop("_live_bias", None)
        return self.bias_cast(self.bias)

    @property
    def _weight(self):
        """core.py:200-203; never aliases the Parameter (the reference's chain ends in at least one `clone()`)"""
        w = self._weight_ro
        return w.clone() if _shares_storage(w, self.weight) else w

    @property
    def _bias(self):
        b = self._bias_ro
        return b.clone() if b is not None and _shares_storage(b, self.bias) else b

    @property
    def effective_weight(self):
        return self.weight_sparsifier(self.weight) if self.weight_sparsifier is not None else self.weight

    def fold_weight_and_bias(self) -> None:
        with torch.no_grad():
            if self.bias_cast is not None and not isinstance(self.bias_format, Same) and self.bias is not None:
                self.bias.data = self.bias_cast(self.bias.data)
                self.bias_cast = CastTo(format=Same())
            if self.weight_sparsifier is not None and not isinstance(self.weight_sparseness, Dense):
                self.weight.data = self.effective_weight
                self.weight_sparsifier = _LazySparsify(sparseness=Dense())
            if self.smoothquant is not None and not self.smoothquant._flag("fused_to_weight"):
                self.smoothquant.fuse_to_weight(self.weight)
            if self.weight_storage_cast is not None and not isinstance(self.weight_storage_cast.format, Same):
                self.weight.data = self.weight_storage_cast(self.weight.data)
                self.weight_storage_cast = CastTo(format=Same())
            if self.weight_cast is not None and not isinstance(self.weight_cast.format, Same):
                self.weight.data = self.weight_cast(self.weight.data)
                self.weight_cast = CastTo(format=Same())
        self._mark_internal_casts()

    # ------------------------------------------------------------------ forward (core.py:215-264)
    def update_smoothquant_scale(self, input):
        if self.smoothquant is not None:  # layer_reconstruction.py:32-34: calibrates against the MASKED weight
            self.smoothquant(input, self.effective_weight)

    def enable_smoothquant_calib(self, state: bool, hyperparams=None, migration_strength: float = 0.5,
                                 fuse_to_weight: bool = False):
        """layer_reconstruction.py:57-68; `hyperparams` is a DmxModuleSmoothQuantHyperparams (or pass the two fields)."""
        if hyperparams is not None:
            migration_strength, fuse_to_weight = hyperparams.migration_strength, hyperparams.fuse_to_weight
        if self.smoothquant is not None:
            if self.smoothquant._flag("fused_to_weight"):
                raise RuntimeError("SmoothQuant cannot be calibrated because it has been fused to weight already")
            self.smoothquant.set_migration_strength(migration_strength)
            self.smoothquant.set_dynamic(False)
            self.smoothquant.enable(not state)
            self.smoothquant.calibrating = state
            if not state and fuse_to_weight:
                self.smoothquant.fuse_to_weight(self.weight)

    def enable_quantizer_calib(self, state: bool, hyperparams) -> None:
        """layer_reconstruction.py:36-55: switch the boundary / weight casts between observing and fake-quantising."""
        if hyperparams.inputs is not None:
            for k in self.input_casts.keys():
                self.input_casts[k].enable_calibration(state, **vars(hyperparams.inputs[k]))
        if hyperparams.outputs is not None:
            for k in self.output_casts.keys():
                self.output_casts[k].enable_calibration(state, **vars(hyperparams.outputs[k]))
        if getattr(self, "weight", None) is not None and self.weight_cast is not None:
            if hyperparams.weight is not None:
                self.weight_cast.enable_calibration(state, **vars(hyperparams.weight))
            if hyperparams.weight_storage is not None and self.weight_storage_cast is not None:
                self.weight_storage_cast.enable_calibration(state, **vars(hyperparams.weight_storage))

    @contextmanager
    def calibrating_quantizers(self, hyperparams):
        self.enable_quantizer_calib(True, hyperparams)
        yield self
        self.enable_quantizer_calib(False, hyperparams)

    @contextmanager
    def calibrating_smoothquant(self, hyperparams):
        self.enable_smoothquant_calib(True, hyperparams)
        yield self
        self.enable_smoothquant_calib(False, hyperparams)

    #: use the single-kernel activation path (dmxq_input_hypernet) when the configuration allows it; results are bit-identical
    fuse_input_hypernet = True

    def _fused_input(self, x):
        """SmoothQuant `x / scale` -> BFP input cast in ONE launch (6 B/element instead of 14 for a bf16 input), or None when this
        configuration must take the two steps: inference only, channels and blocks along the last dim, plain BFP input format
        with nearest rounding, no observer / pre-transform on the input cast."""
        from .format import BlockFloatingPoint
        sq = self.smoothquant
        if (not self.fuse_input_hypernet or not sq._flag("enabled") or not isinstance(x, torch.Tensor) or not x.is_floating_point()
                or x.dim() == 0 or sq.ch_axis not in (-1, x.dim() - 1) or torch.is_grad_enabled() and x.requires_grad
                or torch.compiler.is_compiling()):
            return None
        ic = self.input_casts[next(iter(self.input_casts.keys()))] if len(self.input_casts) else None
        if ic is None:
            return None
        fmt = ic.format
        if (not isinstance(fmt, BlockFloatingPoint) or fmt.rounding != "nearest" or fmt.block_size < 8 or ic.pre_transform
                or ic.block_dim not in (-1, x.dim() - 1) or not ic._flag("fake_quant_enabled") or ic._flag("observer_enabled")):
            return None
        from . import ops
        return ops.input_hypernet(x.detach(), sq.scale, fmt.precision, fmt.block_size, fmt.symmetric)

    def _first_input_cast(self):
        ic = self.input_casts
        return ic[next(iter(ic.keys()))] if len(ic) else None

    def _input_cast_at(self, idx: int):
        """the cast CastToDict.forward applies to the idx-th positional tensor argument"""
        keys = list(self.input_casts.keys())
        return self.input_casts[keys[idx]] if 0 <= idx < len(keys) else None

    #: apply the linked consumers' BFP input cast in this module's launch (dmxq_softmax_cast_bfp, dmxq_layernorm_cast_bfp,
    #: dmxq_rmsnorm_cast_bfp); see `link_consumer`
    fuse_next_cast = True

    def _output_cast_absorbed(self, y) -> bool:
        """True when this module's (single) output cast may be skipped because every linked consumer (nn.link_consumer) applies the same
        nearest-rounding FloatingPoint format to the value as its first input cast: F(F(y)) == F(y) for such a cast (round to a grid,
        saturate, flush -- also through CastTo's `.to(dtype)` after each), so the consumer's result is unchanged.  Not while either cast
        observes (calibration must see the data it would see), has a pre-transform, or is switched off."""
        from .format import FloatingPoint
        consumers = self.__dict__.get("_next_consumers") if self.fuse_next_cast else None
        if not consumers or not isinstance(y, torch.Tensor) or len(self.output_casts) != 1 or torch.compiler.is_compiling() \
                or torch.is_grad_enabled() and y.requires_grad:
            return False
        oc = self.output_casts[next(iter(self.output_casts.keys()))]
        f = oc.format
        if (not isinstance(f, FloatingPoint) or f.rounding != "nearest" or f.unsigned or oc.pre_transform
                or not oc._flag("fake_quant_enabled") or oc._flag("observer_enabled")):
            return False
        for c, idx in consumers:
            nc = c._input_cast_at(idx)
            sq = getattr(c, "smoothquant", None) if idx == 0 else None
            if nc is None or (sq is not None and (sq._flag("enabled") or sq._flag("dynamic") or sq.calibrating)):
                return False
            g = nc.format
            if (not isinstance(g, FloatingPoint) or g.rounding != "nearest" or g.unsigned or nc.pre_transform
                    or not nc._flag("fake_quant_enabled") or nc._flag("observer_enabled")
                    or (g.mantissa, g.exponent, g.bias, bool(g.flush_subnormal)) != (f.mantissa, f.exponent, f.bias, bool(f.flush_subnormal))):
                return False
        return True

    def _linked_bfp_cast(self, x):
        """the consumers' (common) first input cast when it can ride in this module's launch: a live link (nn.link_consumer), inference,
        every consumer with the SAME plain BFP format (symmetric, nearest, along this module's last dim), fake-quantising and not
        observing, no pre-transform, and no SmoothQuant scaling in front of it.  Returns (casts, format) or None."""
        from .format import BlockFloatingPoint
        consumers = self.__dict__.get("_next_consumers") if self.fuse_next_cast else None
        if not consumers or torch.compiler.is_compiling() or torch.is_grad_enabled() and x.requires_grad:
            return None
        casts, fmt0 = [], None
        for c, idx in consumers:
            if idx != 0:   # (the consumer-side skip exists for the first input only)
                return None
            nc = c._first_input_cast()
            sq = getattr(c, "smoothquant", None)
            if nc is None or (sq is not None and (sq._flag("enabled") or sq._flag("dynamic") or sq.calibrating)):
                return None
            fmt = nc.format
            if (not isinstance(fmt, BlockFloatingPoint) or fmt.rounding != "nearest" or not fmt.symmetric or nc.pre_transform
                    or nc.block_dim not in (-1, x.dim() - 1) or not nc._flag("fake_quant_enabled") or nc._flag("observer_enabled")):
                return None
            if fmt0 is not None and (fmt.precision != fmt0.precision or fmt.block_size != fmt0.block_size):
                return None
            fmt0 = fmt
            casts.append(nc)
        return tuple(casts), fmt0

    def _fused_forward(self, input, *args, **kwargs):
        """A module whose whole forward (input casts -> op -> output cast) exists as ONE kernel returns its result here, or None to
        take the general path.  Results must be bit-identical to the general path."""
        return None

    #: run an activation / normalisation module (input cast -> function -> output cast) as ONE launch (dmxq_unary_cast,
    #: dmxq_softmax_cast, dmxq_layernorm_cast, dmxq_rmsnorm_cast) when the configuration allows; see `_act_casts`
    fuse_activation = True

    def _act_casts(self, x, func_id):
        """(input format, output format, approximator wrapper params) when this module's forward may run as ONE launch of this
        library's kernel for `func_id`, else None: inference only, one float tensor on the GPU, no SmoothQuant, one input and one
        output cast that are SAME or nearest-rounding FloatingPoint formats, and an approximator that is NONE or
        `func_id[dmxq]` (the exact function on these kernels -- then the input clamp of the Softmax wrapper applies)."""
        if (not self.fuse_activation or not isinstance(x, torch.Tensor) or x.dtype not in (torch.bfloat16, torch.float16, torch.float32)
                or not x.is_cuda or x.numel() == 0 or self.smoothquant is not None and self.smoothquant._flag("enabled")
                or torch.is_grad_enabled() and x.requires_grad or torch.compiler.is_compiling()):
            return None
        fn, wp = self.approximator.function, {}
        if not isinstance(fn, NoApproximation):
            if getattr(fn, "algorithm", None) != "dmxq" or fn.func_id != func_id or fn.extra_params:
                return None
            wp = fn.wrapper_params
        ics, ocs = list(self.input_casts.values()), list(self.output_casts.values())
        if len(ics) != 1 or len(ocs) != 1:
            return None
        (ok_i, fi), (ok_o, fo) = _range_only_format(ics[0], x.dtype), _range_only_format(ocs[0], x.dtype)
        return (fi, fo, wp) if ok_i and ok_o else None

    def _plain_param(self, name, dtype):
        """the parameter `name` when the module hands it to the function UNCHANGED (every stage of its hypernet a no-op) and it is
        of dtype `dtype`; False when a cast / sparsifier / SmoothQuant applies (the fused norm kernels take raw parameters)"""
        p = getattr(self, name, None)
        if p is None:
            return None
        if p.dtype != dtype or not p.is_cuda or torch.is_grad_enabled() and p.requires_grad:  # (autograd needs torch's own function)
            return False
        casts = [self.bias_cast] if name == "bias" else [self.weight_cast, self.weight_storage_cast]
        for c in casts:
            if c is not None and not (isinstance(c.format, Same) and not c.pre_transform):
                return False
        if name == "weight":
            if self.weight_sparsifier is not None and not isinstance(self.weight_sparsifier.sparseness, Dense):
                return False
            if self.smoothquant is not None and not self.smoothquant._flag("fused_to_weight") and self.smoothquant._flag("enabled"):
                return False
        return p.detach()

    #: 16-bit tensors: run a unary module (input cast, function, output cast) as a 65,536-entry TABLE lookup (csrc/lut16.hip): the
    #: table is built once per (function, casts, dtype, device) with the function in float64 rounded ONCE -- the correctly rounded
    #: result, bit-identical to the reference's CPU evaluation wherever that is itself exact -- and kept on the module.
    #:   True (default)  every function, every 16-bit tensor of a whole number of 16-byte vectors (numel % 8 == 0, 16-byte aligned):
    #:                   ONE rounding policy at every such size.  What it does NOT cover takes the direct kernel -- float32 evaluation
    #:                   rounded once, within one ulp of the same value -- and says so here rather than silently: tensors that are not
    #:                   whole aligned vectors; and a stream CAPTURE that finds no table yet RAISES (a capture must not allocate
    #:                   long-lived state, and falling back would make the captured module return other last bits than the eager
    #:                   one: run one forward first, or capture through `GraphedForward`, whose warm-up forwards build the table).
    #:                   Costs ~1 us per launch over the direct kernel on small tensors (the 128 KiB table copy per workgroup: 4.2 vs
    #:                   3.1 us on Llama's [128, 14336] SiLU input, 3.0 vs 1.9 us on 128 KB), nothing from ~16 MiB up, and is the
    #:                   FASTER kernel for the GELU family from ~8 MiB (4096 x 4096 bf16: 12.7 vs 15.3-16.1 us) -- profiles/r04_small_tensor_ops.txt
    #:   "auto"          the table only where it is not slower than the direct kernel (dmxq_unary_cast: within one ulp of the same
    #:                   value): GELU (erf / tanh) and QuickGELU from 4 M elements up
    #:   False           never
    lut_activation = True
    lut_min_elems = 8
    _LUT_AUTO_MIN = {"gelu": 4 << 20, "gelu_tanh": 4 << 20, "quick_gelu": 4 << 20}

    def _lut_wanted(self, x, func) -> bool:
        if not self.lut_activation or x.element_size() != 2 or x.numel() % 8 != 0 or x.numel() < self.lut_min_elems:
            return False
        if self.lut_activation == "auto":
            return x.numel() >= self._LUT_AUTO_MIN.get(func, 1 << 62)
        return True

    def _unary_table(self, x, func, cast_in, cast_out):
        from . import ops
        key = (func, repr(cast_in), repr(cast_out), x.dtype, x.device)
        cache = self.__dict__.setdefault("_lut_cache", {})   # (a plain attribute: not a buffer, not in the state_dict; rebuilt on demand)
        t = cache.get(key)
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                # no allocation of long-lived state inside a capture.  With the table as the module's POLICY (lut_activation = True)
                # a silent fall-back to the direct kernel would make the captured forward differ from the eager one in last bits
                # (ADVICE r4): refuse; "auto" is a speed choice and may fall back
                if self.lut_activation is True:
                    raise RuntimeError(f"{type(self).__name__}: the {func} table for {x.dtype} is not built yet and a stream capture is in progress; "
                                       "run one forward before capturing (GraphedForward's warm-up does), or set lut_activation = 'auto' / False")
                return None
            t = ops.unary_cast_table(x, func, cast_in, cast_out)
            if t is None:
                return None
            if len(cache) >= 8:
                cache.clear()
            cache[key] = t
        return t

    def _fused_unary(self, x, func_id, func, args, kwargs):
        """input cast -> per-element function -> output cast as ONE launch (a table lookup for 16-bit tensors: dmxq_lut16_apply, else
        dmxq_unary_cast), or None"""
        c = None if (args or kwargs) else self._act_casts(x, func_id)
        if c is None or c[2]:
            return None
        from . import ops
        out = None
        if self._lut_wanted(x, func):
            table = self._unary_table(x, func, c[0], c[1])
            if table is not None:
                out = ops.lut16_apply(x.detach(), table)
        if out is None:
            out = ops.unary_cast(x.detach(), func, c[0], c[1])
        if out is not None:
            self.approximation_error = None
        return out

    def forward(self, input, *args, **kwargs):
        whole = self._fused_forward(input, *args, **kwargs)
        if whole is not None:
            return whole
        _dtype = input.dtype
        fused = None
        if self.smoothquant is not None:
            if self.smoothquant._flag("dynamic") or self.smoothquant.calibrating:
                self.update_smoothquant_scale(input)
            fused = self._fused_input(input)
            if fused is None:
                input = self.smoothquant.scale_input(input)
        if fused is not None:
            _input, args, kwargs = self.input_casts(fused, *args, first_done=True, **kwargs)
        elif not torch.compiler.is_compiling() and any(c is self._first_input_cast() for c in _precast_of(input)):
            # the producer already applied THIS module's first input cast in its own launch (nn.link_consumer)
            _input, args, kwargs = self.input_casts(input, *args, first_done=True, **kwargs)
        else:
            _input, args, kwargs = self.input_casts(input, *args, **kwargs)
        _output = self._forward(_input, *args, **kwargs)
        # (a GEMM-bearing module's output cast is a launch of its own; when the linked consumer applies the SAME cast to this value as
        #  its first input cast -- inside its fused kernel -- the cast here is redundant: FloatingPoint casts are projections)
        output = _output if self._output_cast_absorbed(_output) else self.output_casts(_output, output=True)
        if self.align_boundary_dtype:
            output = (type(output)(a.to(_dtype) for a in output) if isinstance(output, (tuple, list)) else output.to(_dtype))
        # module boundary: the reference's SAME casts clone, so its result never aliases an input (a caller may run
        # `out.add_()`); the internal casts here do not copy, so check once and copy only when an alias got through
        # (all-SAME pass-through modules such as a BASELINE-mode Dropout / ResAdd-free identity)
        ins = [t for t in (input, *args, *kwargs.values()) if isinstance(t, torch.Tensor)]
        if isinstance(output, (tuple, list)):
            output = type(output)(o.clone() if _aliases_any(o, ins, self) else o for o in output)
        elif _aliases_any(output, ins, self):
            output = output.clone()
        return output

    # approximator slot (functional/approximate.py:300-327): exact function first, then overwritten by the approximation
    def approximator_wrapper(self, inputs, approx_args, approx_kwargs, **wrapper_kwargs):
        return self.approximator(*inputs, *approx_args, **approx_kwargs)

    def approx_forward(self, inputs, *args, **kwargs):
        fn = self.approximator.function
        if (getattr(fn, "algorithm", None) == "dmxq" and not torch.compiler.is_compiling()
                and not (torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in (*inputs, *args)))):
            # `[dmxq]` IS the exact function, evaluated by this library's kernels: in inference there is nothing to overwrite, so
            # torch's own evaluation (whose only other use is the autograd graph) is skipped; `approximation_error` would be the
            # rounding difference between two evaluations of the same function and is not formed
            self.approximation_error = None
            with torch.no_grad():
                return self.approximator_wrapper(inputs, args, kwargs, **fn.wrapper_params)
        _output = self.functional_forward(*inputs, *args, **kwargs)
        if not isinstance(self.approximator.function, NoApproximation):
            with torch.no_grad():
                _approx = self.approximator_wrapper(inputs, args, kwargs, **self.approximator.function.wrapper_params)
                if isinstance(_approx, tuple):  # modules that return several values (approximate.py:311-319)
                    assert isinstance(_output, tuple), "module and its approximation should both return a tuple"
                    self.approximation_error = [x - y.data for x, y in zip(_approx, _output)]
                    for x, y in zip(_approx, _output):
                        y.data = x
                else:
                    self.approximation_error = _approx - _output.data
                    _output.data = _approx  # (no dtype alignment here, as in the reference: experimental.silu hands back float16)
        return _output


# ---------------------------------------------------------------------------------------------------- modules
class Linear(DmxModule, torch.nn.Linear):
    ch_axis, win_ch_axis, wout_ch_axis, has_accum = -1, -1, 0, True

    def __init__(self, in_features, out_features, bias=True, **kw):
        torch.nn.Linear.__init__(self, in_features, out_features, bias=bias, **kw)
        self._dmx_init(sparsifiable=True)
        self.input_casts.input_cast.block_dim = -1
        self.weight_cast.block_dim = -1
        if self.bias_cast is not None:
            self.bias_cast.block_dim = -1

    def _forward(self, _input):
        if isinstance(self.accum_format, Same):  # torch_modules.py:346-350
            _weight = self.weight_hypernet(self.weight, _input.dtype).to(_input.dtype)
            _bias = self._bias_ro  # (a property: one cast launch per evaluation)
            return F.linear(_input, _weight, None if _bias is None else _bias.to(_input.dtype))
        _weight = self._weight_ro
        _product = self.accum_cast(torch.matmul(_input.to(_weight.dtype), _weight.t()))
        return torch.add(_product, self._bias_ro) if self.bias is not None else _product

    @classmethod
    def from_raw(cls, raw):
        m = cls(raw.in_features, raw.out_features, bias=raw.bias is not None)
        m.weight, m.bias = raw.weight, raw.bias
        return m.to(raw.weight.device, raw.weight.dtype)


class _ConvNd(DmxModule):
    ch_axis, win_ch_axis, wout_ch_axis, has_accum = 1, 1, 0, True

    def _conv_init(self):
        self._dmx_init(sparsifiable=True)
        self.input_casts.input_cast.block_dim = 1   # torch_modules.py:582-585, 674-677
        self.weight_cast.block_dim = 1
        if self.bias_cast is not None:
            self.bias_cast.block_dim = -1

    def _forward(self, _input):  # torch_modules.py:677-686 (Conv2d), 585-594 (Conv1d)
        _weight = self._weight_ro
        _convolution = self.accum_cast(self._conv_forward(_input.to(_weight.dtype), _weight, None))
        if self.bias is not None:
            _b = self._bias_ro
            for _ in range(_convolution.dim() - 2):
                _b = _b.unsqueeze(-1)
            return torch.add(_convolution, _b)
        return _convolution


class Conv2d(_ConvNd, torch.nn.Conv2d):
    def __init__(self, *a, **kw):
        torch.nn.Conv2d.__init__(self, *a, **kw)
        self._conv_init()


class Conv1d(_ConvNd, torch.nn.Conv1d):
    def __init__(self, *a, **kw):
        torch.nn.Conv1d.__init__(self, *a, **kw)
        self._conv_init()


def _range_only_format(cast, dtype):
    """(ok, format-or-None): may this CastTo be folded into a fused elementwise kernel?  SAME, or a FloatingPoint format that is a
    pure range cast for `dtype` (decided by the library: ops.binary_cast returns None otherwise); switches on, no observer, no
    pre-transform."""
    from .format import FloatingPoint
    if cast is None:
        return True, None
    fmt = cast.format
    if isinstance(fmt, Same):
        return (not cast.pre_transform), None
    if (not isinstance(fmt, FloatingPoint) or cast.pre_transform or not cast._flag("fake_quant_enabled") or cast._flag("observer_enabled")):
        return False, None
    # format.py:209-212: the dtype's own format passes the tensor through untouched (no flush, NaN stays NaN) -- SAME for the kernel
    if fmt.native_of() == dtype:
        return True, None
    if fmt.mantissa == 23:
        return False, None
    return True, fmt


def _tag_precast(out, casts):
    """mark a producer's result as already holding its consumers' first input cast (nn.link_consumer): the consumers' cast OBJECTS and
    the tensor's version counter -- an in-place change of the value afterwards voids the mark"""
    out._dmx_precast = casts
    out._dmx_precast_version = out._version


def _precast_of(t):
    casts = getattr(t, "_dmx_precast", None)
    if casts is None or getattr(t, "_dmx_precast_version", -1) != t._version:
        return ()
    return casts


class _BinaryElementwise(DmxModule):
    """ResAdd / Mul: two cast inputs, one elementwise op, one cast output (torch_modules.py:36-80).  In inference on same-shape
    tensors whose three casts are SAME or nearest-rounding FloatingPoint formats the whole module is ONE launch (dmxq_binary_cast:
    a third of the traffic of the four launches; range-only casts of 16-bit tensors -- the BASIC rules on a bf16 model -- on packed
    words); otherwise the general DmxModule.forward."""
    _op = "add"
    fuse_binary = True

    def _fused_forward(self, a, b=None, *args, **kwargs):
        if (not self.fuse_binary or args or kwargs or not isinstance(a, torch.Tensor) or not isinstance(b, torch.Tensor)
                or a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (torch.bfloat16, torch.float16, torch.float32) or a.device != b.device or not a.is_cuda
                or self.smoothquant is not None and self.smoothquant._flag("enabled")
                or torch.is_grad_enabled() and (a.requires_grad or b.requires_grad) or torch.compiler.is_compiling()
                or not isinstance(self.approximator.function, NoApproximation)):
            return None
        ics = list(self.input_casts.values())
        ocs = list(self.output_casts.values()) if self.output_casts is not None else []
        if len(ics) != 2 or len(ocs) > 1:
            return None
        fmts = []
        for c in (ics[0], ics[1], ocs[0] if ocs else None):
            ok, f = _range_only_format(c, a.dtype)
            if not ok:
                return None
            fmts.append(f)
        from . import ops
        nc = self._linked_bfp_cast(a)
        if nc is not None:   # the consumer's BFP input cast in this launch (dmxq_binary_cast_bfp); see Softmax._fused_forward
            out = ops.binary_cast(a.detach(), b.detach(), self._op, *fmts, then_bfp=(nc[1].precision, nc[1].block_size))
            if out is not None:
                _tag_precast(out, nc[0])
                return out
        return ops.binary_cast(a.detach(), b.detach(), self._op, *fmts)


class ResAdd(_BinaryElementwise):
    _op = "add"

    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init(input_names=("input_cast", "residual_cast"))

    def _forward(self, _input, residual):
        return _input + residual


class ActActMatMul(DmxModule):
    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init(input_names=("input_cast", "multiplier_cast"))
        self.input_casts.input_cast.block_dim = -1      # torch_modules.py:197-204
        self.input_casts.multiplier_cast.block_dim = -2
        for c in self.input_casts.values():             # torch.matmul takes any strides: no contiguous copy of q / k^T / v views
            c.keep_layout = True

    def _forward(self, _input, multiplier):
        return torch.matmul(_input, multiplier)


class Softmax(DmxModule, torch.nn.Softmax):
    def __init__(self, dim: int = -1):
        torch.nn.Softmax.__init__(self, dim=dim)
        self._dmx_init()
        self.functional_forward = F.softmax

    def approximator_wrapper(self, inputs, approx_args, approx_kwargs, **wrapper_kwargs):
        if "input_clamp" in wrapper_kwargs:  # torch_modules.py:989-994
            inputs = [torch.clamp(x, min=wrapper_kwargs["input_clamp"]) for x in inputs]
        return self.approximator(*inputs, *approx_args, **approx_kwargs)

    def _forward(self, _input):
        return self.approx_forward((_input,), dim=self.dim)

    def _fused_forward(self, x, *args, **kwargs):
        """input cast -> softmax over the last dim -> output cast as ONE launch (dmxq_softmax_cast)"""
        c = None if (args or kwargs) else self._act_casts(x, "SOFTMAX")
        if c is None or self.dim is None or x.dim() == 0 or self.dim % x.dim() != x.dim() - 1:
            return None
        from . import ops
        nc = self._linked_bfp_cast(x)
        if nc is not None:
            out = ops.softmax_cast(x.detach(), -1, c[0], c[1], c[2].get("input_clamp"), then_bfp=(nc[1].precision, nc[1].block_size))
            if out is not None:
                _tag_precast(out, nc[0])   # a consumer's forward recognises ITS cast object among these and skips it (DmxModule.forward)
                self.approximation_error = None
                return out
        out = ops.softmax_cast(x.detach(), -1, c[0], c[1], c[2].get("input_clamp"))
        if out is not None:
            self.approximation_error = None
        return out



class LayerNorm(DmxModule, torch.nn.LayerNorm):
    def __init__(self, normalized_shape, eps: float = 1e-5, elementwise_affine: bool = True):
        torch.nn.LayerNorm.__init__(self, normalized_shape, eps=eps, elementwise_affine=elementwise_affine)
        self._dmx_init()
        self.functional_forward = F.layer_norm

    def _forward(self, _input):
        return self.approx_forward((_input,), self.normalized_shape, self._weight_ro, self._bias_ro, self.eps)

    def _fused_forward(self, x, *args, **kwargs):
        """input cast -> layer_norm -> output cast as ONE launch (dmxq_layernorm_cast); weight and bias as they are (SAME casts)"""
        c = None if (args or kwargs) else self._act_casts(x, "LAYER_NORM")
        if c is None or c[2]:
            return None
        w, b = self._plain_param("weight", x.dtype), self._plain_param("bias", x.dtype)
        if w is False or b is False:
            return None
        from . import ops
        nc = self._linked_bfp_cast(x) if len(self.normalized_shape) == 1 else None
        if nc is not None:   # the consumers' BFP input cast in the same launch (nn.link_consumer)
            out = ops.layernorm_cast(x.detach(), self.normalized_shape, w, b, self.eps, c[0], c[1], then_bfp=(nc[1].precision, nc[1].block_size))
            if out is not None:
                _tag_precast(out, nc[0])
                self.approximation_error = None
                return out
        out = ops.layernorm_cast(x.detach(), self.normalized_shape, w, b, self.eps, c[0], c[1])
        if out is not None:
            self.approximation_error = None
        return out


class GELU(DmxModule, torch.nn.GELU):
    def __init__(self, approximate: str = "none"):
        torch.nn.GELU.__init__(self, approximate=approximate)
        self._dmx_init()
        self.functional_forward = F.gelu

    def _forward(self, _input):
        return self.approx_forward((_input,), approximate=self.approximate)

    def _fused_forward(self, x, *args, **kwargs):
        return self._fused_unary(x, "GELU", "gelu_tanh" if self.approximate == "tanh" else "gelu", args, kwargs)


class SiLU(DmxModule, torch.nn.SiLU):
    """torch_modules.py:1559-1576"""

    def __init__(self, inplace: bool = False):
        torch.nn.SiLU.__init__(self, inplace=False)
        self._dmx_init()
        self.functional_forward = F.silu

    def _forward(self, _input):
        return self.approx_forward((_input,))

    def _fused_forward(self, x, *args, **kwargs):
        return self._fused_unary(x, "SILU", "silu", args, kwargs)


class QuickGELU(DmxModule):
    """custom_modules.py:112-117: transformers' QuickGELUActivation, `x * sigmoid(1.702 * x)` in the input dtype"""

    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init()
        self.functional_forward = _quick_gelu

    def _forward(self, _input):
        return self.approx_forward((_input,))

    def _fused_forward(self, x, *args, **kwargs):
        return self._fused_unary(x, "QUICK_GELU", "quick_gelu", args, kwargs)


class Exp(DmxModule):
    """torch_modules.py:236-242 (no approximator slot in the reference: plain torch.exp between the casts)"""

    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init()

    def _forward(self, _input):
        return torch.exp(_input)

    def _fused_forward(self, x, *args, **kwargs):
        return self._fused_unary(x, "EXP", "exp", args, kwargs)


class Mul(_BinaryElementwise):
    """torch_modules.py:67-80: elementwise product of two cast inputs (Llama's gate * up)"""
    _op = "mul"

    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init(input_names=("input_cast", "multiplier_cast"))

    def _forward(self, _input, multiplier):
        return _input * multiplier


class ApplyRotaryPosEmb(DmxModule):
    """custom_modules.py:142-194: rotary position embedding of q and k between four input casts and two output casts"""

    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init(input_names=("q_cast", "k_cast", "cos_cast", "sin_cast"))
        self.output_casts = CastToDict(OrderedDict({"q_embed_cast": CastTo(), "k_embed_cast": CastTo()}))
        self._mark_internal_casts()
        self.functional_forward = self._rope

    @staticmethod
    def rotate_half(x):
        x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
        return torch.cat((-x2, x1), dim=-1)

    @classmethod
    def _rope(cls, q, k, cos, sin, unsqueeze_dim=1):
        cos, sin = cos.unsqueeze(unsqueeze_dim), sin.unsqueeze(unsqueeze_dim)
        return (q * cos) + (cls.rotate_half(q) * sin), (k * cos) + (cls.rotate_half(k) * sin)

    def _forward(self, q, k, cos, sin, unsqueeze_dim=1):
        return self.approx_forward((q, k, cos, sin, unsqueeze_dim))

    fuse_rope = True

    def _fused_forward(self, q, k=None, cos=None, sin=None, unsqueeze_dim=1, *args, **kwargs):
        """Four input casts, the exact function (about six torch kernels per operand) and two output casts as TWO launches
        (dmxq_rope_cast for q and for k) when the four tensors share one float dtype and every cast is SAME or a nearest-rounding
        FloatingPoint format; else the general path."""
        ts = (q, k, cos, sin)
        if (not self.fuse_rope or args or kwargs or any(not isinstance(t, torch.Tensor) for t in ts) or q.dtype not in (torch.bfloat16, torch.float16, torch.float32) or any(t.dtype != q.dtype for t in ts)
                or not q.is_cuda or q.dim() != 4 or k.dim() != 4 or unsqueeze_dim not in (1, 2)
                or self.smoothquant is not None and self.smoothquant._flag("enabled")
                or torch.is_grad_enabled() and any(t.requires_grad for t in ts) or torch.compiler.is_compiling()
                or not isinstance(self.approximator.function, NoApproximation)):
            return None
        ics, ocs = list(self.input_casts.values()), list(self.output_casts.values())
        if len(ics) != 4 or len(ocs) != 2:
            return None
        fmts = []
        for c in ics + ocs:
            ok, f = _range_only_format(c, q.dtype)
            if not ok:
                return None
            fmts.append(f)
        from . import ops
        qe = ops.rope_cast(q.detach(), cos.detach(), sin.detach(), unsqueeze_dim, fmts[0], fmts[2], fmts[3], fmts[4])
        if qe is None:
            return None
        ke = ops.rope_cast(k.detach(), cos.detach(), sin.detach(), unsqueeze_dim, fmts[1], fmts[2], fmts[3], fmts[5])
        return None if ke is None else (qe, ke)


class RMSNorm(DmxModule, torch.nn.RMSNorm):
    """torch_modules.py:1144-1170"""

    def __init__(self, normalized_shape, eps: float = 1e-6):
        torch.nn.RMSNorm.__init__(self, normalized_shape, eps=eps)
        self._dmx_init()
        self.functional_forward = F.rms_norm

    def _forward(self, _input):
        return self.approx_forward((_input,), self.normalized_shape, self._weight_ro, self.eps)

    def _fused_forward(self, x, *args, **kwargs):
        """input cast -> rms_norm -> output cast as ONE launch (dmxq_rmsnorm_cast); the weight as it is (SAME cast)"""
        c = None if (args or kwargs) else self._act_casts(x, "RMS_NORM")
        if c is None or c[2]:
            return None
        w = self._plain_param("weight", x.dtype)
        if w is False:
            return None
        from . import ops
        nc = self._linked_bfp_cast(x) if len(self.normalized_shape) == 1 else None
        if nc is not None:   # the consumers' BFP input cast in the same launch (nn.link_consumer)
            out = ops.rmsnorm_cast(x.detach(), self.normalized_shape, w, self.eps, c[0], c[1], then_bfp=(nc[1].precision, nc[1].block_size))
            if out is not None:
                _tag_precast(out, nc[0])
                self.approximation_error = None
                return out
        out = ops.rmsnorm_cast(x.detach(), self.normalized_shape, w, self.eps, c[0], c[1])
        if out is not None:
            self.approximation_error = None
        return out


class ReLU(DmxModule, torch.nn.ReLU):
    def __init__(self, inplace: bool = False):
        torch.nn.ReLU.__init__(self, inplace=False)
        self._dmx_init()

    def _forward(self, _input):
        return F.relu(_input)

    fuse_relu = True

    def _fused_forward(self, x, *args, **kwargs):
        """input cast -> F.relu -> output cast as ONE launch (dmxq_relu_cast) on a 16-bit tensor with range-only casts"""
        if (not self.fuse_relu or args or kwargs or not isinstance(x, torch.Tensor) or x.dtype not in (torch.bfloat16, torch.float16, torch.float32)
                or not x.is_cuda or self.smoothquant is not None and self.smoothquant._flag("enabled")
                or torch.is_grad_enabled() and x.requires_grad or torch.compiler.is_compiling()
                or not isinstance(self.approximator.function, NoApproximation)):
            return None
        ics, ocs = list(self.input_casts.values()), list(self.output_casts.values())
        if len(ics) != 1 or len(ocs) != 1:
            return None
        (ok_i, fi), (ok_o, fo) = _range_only_format(ics[0], x.dtype), _range_only_format(ocs[0], x.dtype)
        if not (ok_i and ok_o):
            return None
        from . import ops
        nc = self._linked_bfp_cast(x)
        if nc is not None:   # the consumer's BFP input cast in this launch (dmxq_relu_cast_bfp)
            out = ops.relu_cast(x.detach(), fi, fo, then_bfp=(nc[1].precision, nc[1].block_size))
            if out is not None:
                _tag_precast(out, nc[0])
                return out
        return ops.relu_cast(x.detach(), fi, fo)


# ---- the remaining module types of modeling/nn/torch_modules.py and custom_modules.py (round 3): each is torch's own op between this
# library's casts (the general DmxModule.forward), so that a model built for the reference's module set configures and runs here.
class ReLU6(DmxModule, torch.nn.ReLU6):
    """torch_modules.py:1501-1557"""

    def __init__(self, inplace: bool = False):
        torch.nn.ReLU6.__init__(self, inplace=False)
        self._dmx_init()
        self.functional_forward = F.relu6

    def _forward(self, _input):
        return self.approx_forward((_input,))


class Tanh(DmxModule, torch.nn.Tanh):
    """torch_modules.py:1619-1673"""

    def __init__(self):
        torch.nn.Tanh.__init__(self)
        self._dmx_init()
        self.functional_forward = torch.tanh

    def _forward(self, _input):
        return self.approx_forward((_input,))


class Dropout(DmxModule, torch.nn.Dropout):
    """torch_modules.py:1379-1440: F.dropout(x, p, training, inplace) between the casts (identity in eval mode)"""

    def __init__(self, p: float = 0.5, inplace: bool = False):
        torch.nn.Dropout.__init__(self, p=p, inplace=False)
        self._dmx_init()
        # (the function itself, never a closure over `self`: copy.deepcopy treats functions as atomic, so a copied module's closure
        #  would keep reading the ORIGINAL's p / training, and a local lambda cannot be pickled -- ADVICE r3)
        self.functional_forward = F.dropout

    def _forward(self, _input, *args, **kwargs):
        return self.approx_forward((_input,), self.p, self.training, False)


class _GELUVariant(DmxModule):
    """custom_modules.py:96-140 (GELUBase subclasses over transformers' activation classes): the formulas of transformers.activations,
    evaluated in the input dtype by torch like the originals"""

    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init()
        self.functional_forward = self._f

    def _forward(self, _input):
        return self.approx_forward((_input,))


class NewGELU(_GELUVariant):
    @staticmethod
    def _f(x):  # NewGELUActivation
        return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * torch.pow(x, 3.0))))


class FastGELU(_GELUVariant):
    @staticmethod
    def _f(x):  # FastGELUActivation
        return 0.5 * x * (1.0 + torch.tanh(x * 0.7978845608 * (1.0 + 0.044715 * x * x)))


class BloomGELU(_GELUVariant):
    @staticmethod
    def _f(x):  # transformers.models.bloom.modeling_bloom.BloomGelu (inference form)
        return x * 0.5 * (1.0 + torch.tanh(0.79788456 * x * (1 + 0.044715 * x * x)))


class ClippedGELU(_GELUVariant):
    def __init__(self, min: float = -10.0, max: float = 10.0):
        self.min, self.max = min, max
        super().__init__()

    def _f(self, x):  # ClippedGELUActivation
        return torch.clip(F.gelu(x), self.min, self.max)


class AdaptiveAvgPool2d(DmxModule, torch.nn.AdaptiveAvgPool2d):
    """torch_modules.py:829-865"""

    def __init__(self, output_size):
        torch.nn.AdaptiveAvgPool2d.__init__(self, output_size)
        self._dmx_init()

    def _forward(self, _input):
        return torch.nn.AdaptiveAvgPool2d.forward(self, _input)


class BatchNorm2d(DmxModule, torch.nn.BatchNorm2d):
    """torch_modules.py:1222-1308: F.batch_norm with the cast weight / bias and the running statistics"""

    def __init__(self, num_features, eps: float = 1e-5, momentum: float = 0.1, affine: bool = True, track_running_stats: bool = True):
        torch.nn.BatchNorm2d.__init__(self, num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats)
        self._dmx_init()

    def _forward(self, _input):
        self._check_input_dim(_input)
        eaf = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked = self.num_batches_tracked + 1
            eaf = 1.0 / float(self.num_batches_tracked) if self.momentum is None else self.momentum
        bn_training = True if self.training else (self.running_mean is None and self.running_var is None)
        use_stats = not self.training or self.track_running_stats
        return F.batch_norm(_input, self.running_mean if use_stats else None, self.running_var if use_stats else None,
                            self._weight_ro if self.weight is not None else None, self._bias_ro, bn_training, eaf, self.eps)


class GroupNorm(DmxModule, torch.nn.GroupNorm):
    """torch_modules.py:1310-1377"""

    def __init__(self, num_groups: int, num_channels: int, eps: float = 1e-5, affine: bool = True):
        torch.nn.GroupNorm.__init__(self, num_groups, num_channels, eps=eps, affine=affine)
        self._dmx_init()

    def _forward(self, _input):
        _weight = self._weight_ro if self.weight is not None else None
        if _weight is not None:
            _input = _input.to(_weight.dtype)
        return F.group_norm(_input, self.num_groups, _weight, self._bias_ro, self.eps)


class ConvTranspose2d(_ConvNd, torch.nn.ConvTranspose2d):
    """torch_modules.py:716-827 (input and weight blocked along dim 1, like Conv2d)"""

    def __init__(self, *a, **kw):
        torch.nn.ConvTranspose2d.__init__(self, *a, **kw)
        self._conv_init()

    def _forward(self, _input, output_size=None):
        if self.padding_mode != "zeros":
            raise ValueError("Only `zeros` padding mode is supported for ConvTranspose2d")
        output_padding = self._output_padding(_input, output_size, self.stride, self.padding, self.kernel_size, 2, self.dilation)
        _weight = self._weight_ro
        _convolution = self.accum_cast(F.conv_transpose2d(_input.to(_weight.dtype), _weight, None, self.stride, self.padding, output_padding,
                                                          self.groups, self.dilation))
        if self.bias is not None:
            return torch.add(_convolution, self._bias_ro.unsqueeze(-1).unsqueeze(-1))
        return _convolution


class BAddBMM(DmxModule):
    """torch_modules.py:267-312: torch.baddbmm(input, batch1, batch2) with batch2 blocked along dim -2"""

    def __init__(self):
        torch.nn.Module.__init__(self)
        self._dmx_init(input_names=("input_cast", "batch1_cast", "batch2_cast"))
        self.input_casts.input_cast.block_dim = -1
        self.input_casts.batch1_cast.block_dim = -1
        self.input_casts.batch2_cast.block_dim = -2

    def _forward(self, input, batch1, batch2, **kwargs):
        return torch.baddbmm(input, batch1, batch2, **kwargs)


def _quick_gelu(x):
    """transformers.activations.QuickGELUActivation, a module-level function (picklable, nothing captured)"""
    return x * torch.sigmoid(1.702 * x)


def _all_same_casts(m) -> bool:
    return all(isinstance(c.format, Same) and not c.pre_transform for c in list(m.input_casts.values()) + list(m.output_casts.values()))


class ScaledDotProductAttention(DmxModule):
    """torch_modules.py:108-192: a COMPOUND module -- F.scaled_dot_product_attention spelled out with ResAdd / ActActMatMul / Mul / Softmax /
    Dropout submodules, each with its own casts (so every fused path above applies inside it)"""
    is_compound = True

    def __init__(self, dropout_p: float = 0.0):
        torch.nn.Module.__init__(self)
        self._dmx_init(input_names=("query_states_cast", "key_states_cast", "value_states_cast", "attn_mask_cast"))
        for c in self.input_casts.values():
            c.block_dim = -1
        self.resadd, self.actmatmul, self.softmax = ResAdd(), ActActMatMul(), Softmax(dim=-1)
        self.dropout, self.mul = Dropout(p=dropout_p), Mul()
        link_consumer(self.softmax, self.actmatmul)   # probabilities -> (identity dropout) -> `attn_weight @ value`: see forward

    def forward(self, query, key, value, attn_mask=None, is_causal=False, scale=None, enable_gqa=False):
        import math
        L, S = query.size(-2), key.size(-2)
        scale_factor = torch.tensor(1 / math.sqrt(query.size(-1)), dtype=torch.float16) if scale is None else scale
        attn_bias = torch.zeros(L, S, dtype=query.dtype).to(query.device)
        if is_causal:
            assert attn_mask is None
            attn_bias.masked_fill_(torch.ones(L, S, dtype=torch.bool, device=query.device).tril(diagonal=0).logical_not(), -10000.0)
        if attn_mask is not None:
            if attn_mask.dtype == torch.bool:
                attn_bias.masked_fill_(attn_mask.logical_not(), -10000.0)
            else:
                attn_bias = self.resadd(attn_bias, attn_mask)
        if enable_gqa:
            key = key.repeat_interleave(query.size(-3) // key.size(-3), -3)
            value = value.repeat_interleave(query.size(-3) // value.size(-3), -3)
        attn_weight = self.actmatmul(query, key.transpose(-2, -1))
        attn_weight = self.resadd(attn_weight, attn_bias)
        if not isinstance(scale_factor, torch.Tensor):
            scale_factor = torch.tensor(float(scale_factor), dtype=attn_weight.dtype)
        attn_weight = self.mul(attn_weight, scale_factor.to(attn_weight.device))
        # softmax -> (dropout) -> matmul inside ONE compound module: when the dropout is the identity (inference, SAME casts) the
        # probabilities have exactly one consumer, and the softmax launch applies the matmul's input cast too (link_consumer)
        identity_dropout = (not self.training or self.dropout.p == 0.0) and _all_same_casts(self.dropout)
        if identity_dropout:
            attn_weight = self.softmax(attn_weight)           # (linked to self.actmatmul in __init__)
            if torch.compiler.is_compiling() or getattr(attn_weight, "_dmx_precast", None) is None:
                attn_weight = self.dropout(attn_weight)       # not fused after all: the reference's call sequence
        else:
            self.softmax.__dict__["fuse_next_cast"] = False   # an active dropout sits between the two: no fusion for this call
            try:
                attn_weight = self.softmax(attn_weight)
            finally:
                del self.softmax.__dict__["fuse_next_cast"]
            attn_weight = self.dropout(attn_weight)
        return self.actmatmul(attn_weight, value)


class MaxPool2d(DmxModule, torch.nn.MaxPool2d):
    def __init__(self, *a, **kw):
        torch.nn.MaxPool2d.__init__(self, *a, **kw)
        self._dmx_init()

    def _forward(self, _input):
        return torch.nn.MaxPool2d.forward(self, _input)


class AvgPool2d(DmxModule, torch.nn.AvgPool2d):
    def __init__(self, *a, **kw):
        torch.nn.AvgPool2d.__init__(self, *a, **kw)
        self._dmx_init()

    def _forward(self, _input):
        return torch.nn.AvgPool2d.forward(self, _input)


class Embedding(DmxModule, torch.nn.Embedding):
    def __init__(self, *a, **kw):
        torch.nn.Embedding.__init__(self, *a, **kw)
        self._dmx_init(sparsifiable=True)

    def forward(self, input):  # integer indices: no input cast, no dtype alignment (reference Embedding)
        _output = F.embedding(input, self._weight_ro, self.padding_idx, self.max_norm, self.norm_type,
                              self.scale_grad_by_freq, self.sparse)
        return self.output_casts(_output, output=True)


def link_consumer(producer: "DmxModule", *consumers: "DmxModule") -> None:
    """Declare that `consumers` are the ONLY users of `producer`'s output and take it as their first input: the Softmax whose
    probabilities go -- through an inactive dropout -- into the `attn_probs @ value` ActActMatMul; the pre-attention norm whose result
    feeds the q / k / v Linears (gate / up, fc1 after the pre-MLP norm).  The producer (Softmax, LayerNorm, RMSNorm) may then apply the
    consumers' first input cast -- the same BFP format for all of them -- in its own launch (dmxq_softmax_cast_bfp,
    dmxq_layernorm_cast_bfp, dmxq_rmsnorm_cast_bfp: one pass over the activation instead of 1 + len(consumers)) and the consumers skip
    theirs; results are bit-identical.  A GEMM-bearing producer (Linear, ActActMatMul, Conv) uses the link the other way round: its
    output cast -- a launch of its own -- is skipped when the consumer applies the same FloatingPoint cast to the value as its first
    input cast (`_output_cast_absorbed`: o_proj -> ResAdd, fc1 -> GELU, `q @ k^T` -> Softmax).  The link is by object: reconfiguring a module's formats or switching on SmoothQuant is picked up at
    the next forward, and whatever the fused kernels do not cover falls back to separate launches.  Do NOT link when anything else reads
    the producer's output (`output_attentions=True`, a residual taken AFTER the norm): it would see the BFP-cast values.
    REPLACING a linked module in the model (or rerouting the value) invalidates the declaration: link again (or run
    `link_consumers_from_fx` again) after structural changes.
    A consumer that takes the value as its k-th positional tensor (the `up` projection into `Mul(act(gate), up)`) is given as
    `(module, k)`: only the redundant-output-cast form applies there.  Unlink: link_consumer(producer) or link_consumer(producer, None)."""
    consumers = tuple((c if isinstance(c, tuple) else (c, 0)) for c in consumers if c is not None)
    if not consumers:
        producer.__dict__.pop("_next_consumers", None)
        return
    if any(c._input_cast_at(i) is None for c, i in consumers):
        raise ValueError("a consumer has no input cast at that position to link")
    # a plain attribute (not registered submodules: no extra state_dict keys), held strongly so that copy.deepcopy / pickle of the model
    # keep the modules together
    producer.__dict__["_next_consumers"] = consumers


def link_consumers_from_fx(gm) -> int:
    """Set `link_consumer` for every producer of a traced model whose users allow it: `gm` is a torch.fx.GraphModule in which DmxModules
    are leaf `call_module` nodes (what the reference's transform produces: modeling/model.py).  For each DmxModule node, follow its
    value through dropouts that are the identity in inference (p == 0 or eval mode, SAME casts); if EVERY remaining
    user is a `call_module` of a DmxModule that takes the value as its FIRST positional argument (and the value is not a graph output),
    link the producer to those modules.  Never linked: a producer instance called at more than one site, consumers that do not run
    DmxModule.forward (compound modules).  Returns the number of producers linked.  (`DmxTracer` below traces a plain nn.Module that way.)"""
    import torch.fx as fx
    from collections import Counter
    mods = dict(gm.named_modules())
    linked = 0
    # The link lives on the MODULE, the graph speaks of call SITES: a module instance called at several sites (a shared `self.act`)
    # has no single set of consumers, and linking it by its last call would hand the BFP-cast (or uncast) value to the readers of
    # its other calls.  Such producers are never linked.
    calls = Counter(n.target for n in gm.graph.nodes if n.op == "call_module")

    def is_identity_dropout(n):
        m = mods.get(n.target) if n.op == "call_module" else None
        return isinstance(m, Dropout) and (not m.training or m.p == 0.0) and _all_same_casts(m)

    def applies_its_input_casts(m):
        # a consumer whose forward is not DmxModule's (the compound ScaledDotProductAttention, which -- like the reference's,
        # torch_modules.py:108-192 -- never applies its own query / key / value casts) would drop an absorbed cast entirely
        return isinstance(m, DmxModule) and not getattr(m, "is_compound", False) and type(m).forward is DmxModule.forward

    for node in gm.graph.nodes:
        prod = mods.get(node.target) if node.op == "call_module" else None
        if not isinstance(prod, DmxModule) or isinstance(prod, Dropout):
            continue
        if calls[node.target] > 1:
            link_consumer(prod)
            continue
        frontier, consumers, ok = [node], [], True
        while frontier and ok:
            n = frontier.pop()
            for u in n.users:
                if u.op == "output":
                    ok = False
                elif is_identity_dropout(u) and u.args and u.args[0] is n:
                    frontier.append(u)
                else:
                    m = mods.get(u.target) if u.op == "call_module" else None
                    pos = [i for i, a in enumerate(u.args) if a is n]
                    if (applies_its_input_casts(m) and len(pos) == 1 and not any(a is n for a in u.kwargs.values())
                            and all(isinstance(a, fx.Node) for a in u.args[:pos[0]])):   # (earlier args are tensors: the index counts tensors)
                        consumers.append((m, pos[0]))
                    else:
                        ok = False
        if ok and consumers:
            link_consumer(prod, *dict.fromkeys(consumers))
            linked += 1
        else:
            link_consumer(prod)
    return linked


def DmxTracer():
    """a torch.fx.Tracer that keeps every DmxModule a leaf (the granularity of the reference's transformed graphs)"""
    import torch.fx as fx

    class _T(fx.Tracer):
        def is_leaf_module(self, m, qualname):
            return isinstance(m, DmxModule) or super().is_leaf_module(m, qualname)

    return _T()


# ---------------------------------------------------------------------------------------------------- rules
class DmxConfigRule:
    """(module types, name regex) -> module config  (modeling/model.py:721-792)"""

    def __init__(self, module_types=(), name_re: str = "", module_config: Optional[dict] = None):
        assert all(issubclass(mt, DmxModule) for mt in module_types)
        self.module_types = tuple(module_types)
        self.name_rule = re.compile(name_re)
        self.module_config = dict(module_config or {})

    def names_in(self, model: torch.nn.Module):
        return [n for n, m in model.named_modules() if isinstance(m, self.module_types) and self.name_rule.match(n)]

    def apply_to(self, model: torch.nn.Module):
        for n, m in model.named_modules():
            if isinstance(m, DmxModule) and isinstance(m, self.module_types) and self.name_rule.match(n):
                m.configure(self.module_config)


def _weight_batches(mods):
    """The modules of `mods` whose weight path can run as a SET: -> (groups, hyper).
    groups: {("bfp" | "fixed", dtype, device, format parameters ...): [module, ...]} -- a single plain cast (BFP along the last dim; calibrated
    INT8 / INT4 per tensor or per row group) through `ops.bfp_qdq_multi` / `ops.fixed_qdq_multi`;
    hyper: {(weight dtype, device, score dtype, K, M, has SmoothQuant scale, precision, block size, symmetric): [(module, score, scale), ...]}
    -- the fused N:M mask / SmoothQuant scale -> BFP chain along the last dim through `ops.weight_hypernet_multi`.
    Everything else (other sparsifiers, storage formats, pre-transforms, per-channel affine casts) is the module's own business."""
    from .format import BlockFloatingPoint, FixedPoint
    from .sparse import Dense
    groups = {}
    with torch.no_grad():
        for m in mods:
            wc, st, sp, sq = m.weight_cast, m.weight_storage_cast, m.weight_sparsifier, m.smoothquant
            if (wc is None or wc.pre_transform or not wc._flag("fake_quant_enabled") or wc._flag("observer_enabled") or not m.weight.is_cuda
                    or (sp is not None and not isinstance(sp.sparseness, Dense))
                    or (sq is not None and not sq._flag("fused_to_weight") and sq._flag("enabled"))
                    or (st is not None and not (isinstance(st.format, Same) and not st.pre_transform))):
                continue
            fmt, w = wc.format, m.weight
            if isinstance(fmt, BlockFloatingPoint) and fmt.block_size > 1 and wc.block_dim in (-1, w.dim() - 1):
                key = ("bfp", w.dtype, w.device, fmt.precision, fmt.block_size, fmt.symmetric, fmt.rounding)
            elif isinstance(fmt, FixedPoint) and not wc.is_per_channel and (wc.group_size is None or wc.ch_axis in (0, -w.dim())):
                key = ("fixed", w.dtype, w.device, fmt.precision, fmt.fraction, fmt.clamp, fmt.symmetric, fmt.rounding, wc.group_size)
            else:
                continue
            groups.setdefault(key, []).append(m)
        # modules whose weight path is the FUSED chain along the last dim (N:M mask and / or SmoothQuant scale -> BFP: Llama-3-8B under
        # BASELINE.json configs[3]): the whole set in one launch per 32 weights (dmxq_weight_hypernet_multi) -- the loop of
        # modeling/nn/core.py:178-198 over a layer's weights; the modules' own fold then finds every stage done
        hyper = {}
        for m in mods:
            if any(m in ms for ms in groups.values()) or not m.weight.is_cuda or m.training:
                continue
            with torch.no_grad():
                plan = m._fused_weight_plan(m.weight.data)
            if plan is None or not plan[2]:
                continue
            fmt, _, _, score, K, M, sq = plan
            key = (m.weight.dtype, m.weight.device, score.dtype if score is not None else None, K, M, sq is not None,
                   fmt.precision, fmt.block_size, fmt.symmetric)
            hyper.setdefault(key, []).append((m, score, sq))
    return groups, hyper


def _bias_batches(mods):
    """The modules of `mods` whose BIAS cast can run as a set: {(dtype, device, mantissa, exponent, bias, flush, unsigned): [module, ...]}
    -- a plain FloatingPoint cast with nearest rounding, or BFP with blocks of ONE element (BFP32_1, the BASIC rules' bias format), no
    pre-transform, no observer, fake-quant on, through `ops.float_qdq_multi`; native formats that pass the tensor through, FLOAT32 and every other format stay the module's own."""
    from .format import BlockFloatingPoint, FloatingPoint
    groups = {}
    for m in mods:
        bc, b = getattr(m, "bias_cast", None), getattr(m, "bias", None)
        if bc is None or b is None or not b.is_cuda or bc.pre_transform or not bc._flag("fake_quant_enabled") or bc._flag("observer_enabled"):
            continue
        fmt = bc.format
        if isinstance(fmt, FloatingPoint) and fmt.rounding == "nearest" and fmt.mantissa < 23 and fmt.native_of() != b.dtype:
            key = (b.dtype, b.device, fmt.mantissa, fmt.exponent, fmt.bias, bool(fmt.flush_subnormal), bool(fmt.unsigned))
        elif isinstance(fmt, BlockFloatingPoint) and fmt.block_size == 1 and fmt.rounding == "nearest" and 2 <= fmt.precision <= 24:
            # BFP32_1, the BASIC rules' bias format: a block of ONE element borrows float_quantize (numerical/format.py:312-320; the
            # library's dmxq_bfp_qdq makes the same detour): precision - 2 mantissa bits, float32's exponent field, subnormals kept
            key = (b.dtype, b.device, fmt.precision - 2, 8, 127, False, False)
        else:
            continue
        groups.setdefault(key, []).append(m)
    return groups


def fold_weights_and_biases(model: torch.nn.Module) -> torch.nn.Module:
    """`DmxModel.fold_weights_and_biases` (modeling/model.py: every DmxModule's `fold_weight_and_bias`, core.py:146-176) for a
    model built from these modules: weights are quantised ONCE instead of on every forward.  The per-module weight casts are
    the launch-bound part on a small model (opt-125m: 73 Linear weights of 0.6-2.4 M elements, ~4 us of launch each for
    0.4-1.6 us of streaming), so the modules whose weight path is a single plain cast are BATCHED through the multi-tensor
    entry points (`ops.bfp_qdq_multi` for BFP formats, `ops.fixed_qdq_multi` for calibrated INT8 / INT4 per-tensor or
    row-group quantisation: one launch per 40-48 tensors; `ops.weight_hypernet_multi` for the fused N:M mask / SmoothQuant scale ->
    BFP chain along the last dim: one launch per 32 weights); everything else -- sparsifiers, SmoothQuant, storage formats,
    pre-transforms, the biases -- goes through the module's own `fold_weight_and_bias`, which also finishes the batched ones
    (their weight cast is already SAME by then).  Bit-identical to folding module by module."""
    from . import ops
    from .sparse import Dense
    mods = [m for m in model.modules() if isinstance(m, DmxModule) and getattr(m, "weight", None) is not None]
    with torch.no_grad():
        groups, hyper = _weight_batches(mods)
        for (_, _, _, K, M, has_sq, precision, block_size, symmetric), items in hyper.items():
            outs = ops.weight_hypernet_multi([m.weight.data for m, _, _ in items], precision, block_size, symmetric,
                                             [sc for _, sc, _ in items] if M else None, K, M, [q for _, _, q in items] if has_sq else None)
            if outs is None:   # some member is not fusable as a set (alignment, dtype triple): the modules fold one by one below
                continue
            for (m, _, _), o in zip(items, outs):
                m.weight.data = o
                if M:
                    m.weight_sparsifier = _LazySparsify(sparseness=Dense())
                if has_sq:
                    m.smoothquant._set_flag("fused_to_weight", True)
                m.weight_cast = CastTo(format=Same())
        for key, ms in groups.items():
            ws = [m.weight.data for m in ms]
            if key[0] == "bfp":
                _, _, _, precision, block_size, symmetric, rounding = key
                outs = ops.bfp_qdq_multi(ws, precision, block_size, -1, symmetric, rounding)
            else:
                _, _, _, precision, fraction, clamp, symmetric, rounding, gs = key
                outs = ops.fixed_qdq_multi(ws, precision, fraction, clamp, symmetric, [m.weight_cast.scale for m in ms],
                                           [m.weight_cast.zero_point for m in ms], group_size=gs, rounding=rounding)
            for m, o in zip(ms, outs):
                m.weight.data = o
                m.weight_cast = CastTo(format=Same())
        for m in model.modules():
            if isinstance(m, DmxModule):
                m.fold_weight_and_bias()
    return model


def _live_scopes(root: torch.nn.Module, max_bytes):
    """The modules a `LiveWeightBatch` hooks: the LARGEST disjoint submodules of `root` that hold at least two weight-bearing DmxModules
    and at most `max_bytes` of their weights (None: `root` itself, whatever its size).  A scope's weights are quantised together when its
    forward starts and released when it ends, so `max_bytes` bounds the quantised copies alive at any time: an opt-125m (250 MB) is one
    scope, a Llama-3-8B (16 GB in bf16) becomes one scope per decoder layer (436 MB), its lm_head (one module) is left to the per-module
    path.  Containers that are never CALLED (ModuleList / ModuleDict: no forward, their hooks would never fire) are only descended into."""
    def weights(mod):
        return [m.weight for m in mod.modules() if isinstance(m, DmxModule) and getattr(m, "weight", None) is not None]

    def walk(mod):
        ws = weights(mod)
        if len(ws) < 2:
            return []
        callable_ = not isinstance(mod, (torch.nn.ModuleList, torch.nn.ModuleDict))
        if callable_ and (max_bytes is None or sum(w.numel() * w.element_size() for w in ws) <= max_bytes):
            return [mod]
        out = []
        for c in mod.children():
            out += walk(c)
        return out

    return walk(root)


class LiveWeightBatch:
    """The weight chains of a (sub)model's DmxModules as a few multi-tensor launches per forward, for weights that are NOT folded --
    the reference re-runs every module's weight hypernet on every forward (modeling/nn/core.py:178-203), one launch chain per weight:
    an opt-125m decoder layer launches six INT8 casts of 0.6-2.4 M elements per forward, each ~4 us of launch for ~1 us of streaming.

        batch = LiveWeightBatch(model)      # forward hooks on the model's scopes (the model itself when it is small, see below)
        y = model(x)                        # a scope's pre-hook: ONE launch per group of sibling weights; the modules then pick their
        batch.remove()                      # result up; the scope's post-hook drops the results again

    The bias casts of the same modules (a plain FloatingPoint format, the BASIC rules' bias format) go the same way: one
    `ops.float_qdq_multi` launch per group instead of one launch per module (six per opt-125m decoder layer).
    A scope's pre-hook plans the groups afresh on every forward (`_weight_batches`: same dtype / device / format / N:M pattern, the rule
    of `fold_weights_and_biases`), launches `ops.fixed_qdq_multi` / `ops.bfp_qdq_multi` / `ops.weight_hypernet_multi` and leaves each
    result on its module, stamped with the Parameter's storage and version: `DmxModule.weight_hypernet` returns it for THAT state of
    the Parameter only (an optimiser step or an assignment in between invalidates it) and only in inference.  The stamps live for the
    scope's forward ONLY (round 6): a forward hook registered with `always_call=True` pops them when the scope's forward
    returns or raises, so a later direct call of a submodule, of `m.weight_hypernet(m.weight)` or of `m._weight_ro` -- layer-wise
    calibration, a recalibrated scale, a new sparsifier score, none of which the stamp can see -- runs the module's own chain, and no
    quantised copy is held between forwards.  Weights stay "live": a changed weight, scale or configuration shows in the next forward.
    Results are bit-identical to the per-module path (the multi-tensor entry points are, tests/test_gpu_round2.py / round4.py).

    Memory: the quantised copies of ONE scope at a time.  `max_bytes` (default 1 GiB of weights) picks the scopes (`_live_scopes`): the
    whole of a small model, one decoder layer at a time of a large one.  `GraphedForward(..., batch_live_weights=True)` installs one for
    the capture and removes it afterwards."""

    def __init__(self, root: torch.nn.Module, replan: bool = True, max_bytes: Optional[int] = 1 << 30):
        """replan: plan the groups afresh on every forward (any reconfiguration shows at once; ~13 us of Python per weight-bearing
        module -- nothing inside a graph capture, 10 % of an EAGER opt-125m layer).  False: plan at the first forward and keep it
        until `refresh()` -- for a configuration that is frozen while the batch is installed."""
        self.root, self.replan = root, replan
        self._plans, self._stamped = {}, {}
        self.scopes = _live_scopes(root, max_bytes)
        self._handles = []
        for sc in self.scopes:
            self._handles.append(sc.register_forward_pre_hook(self._prepare))
            self._handles.append(sc.register_forward_hook(self._release, always_call=True))

    @property
    def _plan(self):
        """the kept plan of the first scope (replan = False), None before its first forward"""
        return self._plans.get(id(self.scopes[0])) if self.scopes else None

    def refresh(self):
        """drop the kept plans (replan = False): the next forward groups the weights again"""
        self._plans.clear()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        self._stamped.clear()
        for m in self.root.modules():
            if isinstance(m, DmxModule):
                m.__dict__.pop("_live_weight", None)
                m.__dict__.pop("_live_bias", None)

    def _release(self, module, args, output):
        """the scope's forward is over (returned or raised): its stamps, and the quantised copies they hold, go"""
        for m in self._stamped.pop(id(module), ()):
            m.__dict__.pop("_live_weight", None)
            m.__dict__.pop("_live_bias", None)

    def _prepare(self, module, args):
        from . import ops
        plan = self._plans.get(id(module))
        if self.replan or plan is None:
            mods = [m for m in module.modules() if isinstance(m, DmxModule) and getattr(m, "weight", None) is not None]
        else:
            mods = plan[0]
        for m in mods:
            m.__dict__.pop("_live_weight", None)
            m.__dict__.pop("_live_bias", None)
        self._stamped[id(module)] = mods
        if torch.compiler.is_compiling():
            return
        if torch.is_grad_enabled() and any(m.weight.requires_grad or (getattr(m, "bias", None) is not None and m.bias.requires_grad) for m in mods):
            mods = [m for m in mods if not (m.weight.requires_grad or (getattr(m, "bias", None) is not None and m.bias.requires_grad))]
            kept = None
        else:
            kept = plan if not self.replan else None
        if len(mods) < 2:
            return
        with torch.no_grad():
            if kept is not None:
                _, groups, hyper, biases = kept
            else:
                live = [m for m in mods if m.weight.is_cuda]
                groups, hyper = _weight_batches(live)
                biases = _bias_batches(live)
                if not self.replan and not torch.is_grad_enabled():
                    self._plans[id(module)] = (mods, groups, hyper, biases)
            def stamp_bias(m, o):
                m.__dict__["_live_bias"] = (o, m.bias._version, m.bias.data_ptr())

            # ONE launch for the INT8 weight casts AND the bias casts when the layer has one group of each, same dtype and device
            # (`dmxq_fixed_float_qdq_multi`: opt-125m's six weights + six biases; 2.3 us of a 185 us forward)
            fixed_keys = [k for k, ms in groups.items() if k[0] == "fixed" and len(ms) >= 2 and k[7] == "nearest"]
            bias_keys = [k for k, ms in biases.items() if len(ms) >= 2]
            done_fixed = done_bias = None
            if len(fixed_keys) == 1 and len(bias_keys) == 1 and fixed_keys[0][1:3] == bias_keys[0][0:2]:
                fk, bk = fixed_keys[0], bias_keys[0]
                ms, bs = groups[fk], biases[bk]
                _, _, _, precision, fraction, clamp, symmetric, _, gs = fk
                if fraction == 0 and clamp and len(ms) <= 12 and len(bs) <= 12 and len(ms) + len(bs) <= 21:
                    wo, bo = ops.fixed_float_qdq_multi([m.weight.detach() for m in ms], precision, fraction, clamp, symmetric,
                                                       [m.weight_cast.scale for m in ms], [m.weight_cast.zero_point for m in ms], gs,
                                                       [m.bias.detach() for m in bs], bk[2], bk[3], bk[4], bk[5], bk[6])
                    for m, o in zip(ms, wo):
                        m.__dict__["_live_weight"] = (o, m.weight._version, m.weight.data_ptr(), m.weight.dtype)
                    for m, o in zip(bs, bo):
                        stamp_bias(m, o)
                    done_fixed, done_bias = fk, bk
            # the bias casts of the same modules (a plain FloatingPoint format): one launch per group instead of one per module
            for key, ms in biases.items():
                if len(ms) < 2 or key == done_bias:
                    continue
                (_, _, man, exp, ebias, flush, unsigned) = key
                outs = ops.float_qdq_multi([m.bias.detach() for m in ms], man, exp, ebias, flush, unsigned)
                for m, o in zip(ms, outs):
                    stamp_bias(m, o)

            def stamp(m, o, natural=None):
                m.__dict__["_live_weight"] = (o, m.weight._version, m.weight.data_ptr(), natural or m.weight.dtype)

            for (_, _, _, K, M, has_sq, precision, block_size, symmetric), items in hyper.items():
                if len(items) < 2:
                    continue
                # (rounded straight to the weight's dtype, what `Linear._forward` asks its own fused launch for: `_weight.to(_input.dtype)`)
                outs = ops.weight_hypernet_multi([m.weight.detach() for m, _, _ in items], precision, block_size, symmetric,
                                                 [sc for _, sc, _ in items] if M else None, K, M, [q for _, _, q in items] if has_sq else None,
                                                 out_dtype=items[0][0].weight.dtype)
                if outs is not None:
                    for (m, sc, _), o in zip(items, outs):
                        stamp(m, o, torch.promote_types(m.weight.dtype, sc.dtype) if (M and sc is not None) else m.weight.dtype)
            for key, ms in groups.items():
                if len(ms) < 2 or key == done_fixed:
                    continue
                ws = [m.weight.detach() for m in ms]
                if key[0] == "bfp":
                    _, _, _, precision, block_size, symmetric, rounding = key
                    if rounding == "stochastic":
                        continue   # (an implicit seed per call: the per-module path draws one per weight)
                    outs = ops.bfp_qdq_multi(ws, precision, block_size, -1, symmetric, rounding)
                else:
                    _, _, _, precision, fraction, clamp, symmetric, rounding, gs = key
                    if rounding == "stochastic":
                        continue
                    outs = ops.fixed_qdq_multi(ws, precision, fraction, clamp, symmetric, [m.weight_cast.scale for m in ms],
                                               [m.weight_cast.zero_point for m in ms], group_size=gs, rounding=rounding)
                for m, o in zip(ms, outs):
                    stamp(m, o)


class GraphedForward:
    """A configured model's inference forward as ONE hipGraph replay.

    Small layers are HOST-bound in eager mode: an opt-125m decoder layer under the BASIC rules is ~60 launches of a few microseconds
    each, and Python + dispatcher + hipLaunchKernel cost more than the kernels (775 us eager vs 258 us of GPU time per forward,
    profiles/r03_layer_opt125m.json).  Every launch of this library goes to torch's current stream with no host synchronisation
    (DESIGN.md §6.7), so a forward can be captured once and replayed:

        g = GraphedForward(model, example_x)          # warm-up forwards on a side stream, then capture
        y = g(x)                                       # copies x into the static input, replays, returns the static output

    Static shapes and dtypes (one graph per input signature); the returned tensors are the graph's own buffers and are
    overwritten by the next call -- clone what must outlive it.  Inference only (captured under torch.no_grad); calibration
    (observers, SmoothQuant `calibrating`) and stochastic rounding with an implicit seed must be done before capture."""

    def __init__(self, model: torch.nn.Module, *example_inputs: torch.Tensor, warmup: int = 3, batch_live_weights: bool = True,
                 live_batch_max_bytes: Optional[int] = 1 << 30, calibrating: bool = False):
        """calibrating: capture a forward whose MinMax observers are ENABLED (round 6).  An observer step is then part of the graph --
        `dmxq_group_minmax_accumulate` folds the batch's extrema into the running `min_val` / `max_val` IN PLACE, `dmxq_qparams` derives
        scale and zero point, and they are copied into the cast's persistent buffers: three launches, no host read -- so every replay
        `g(batch)` is one calibration step on that batch, with the observer state of the same steps run eagerly, bit for bit
        (numerical/cast.py:179-226, numerical/observer.py:173-193).  The warm-up forwards observe the example input (a running min / max
        is idempotent under repeats, but it must be a batch the calibration may see).  Only MinMax observers without a process group:
        a HistogramObserver searches its thresholds on the host, a sharded exchange is a collective.  Default False: an enabled
        observer in a model that is captured for inference is a mistake and is refused."""
        if not example_inputs or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise ValueError("GraphedForward: tensor inputs on the GPU required")
        # calibration must precede an INFERENCE capture: an enabled observer updates its running min / max IN PLACE and a calibrating
        # SmoothQuant recomputes its scale -- captured, every replay would repeat that on the static input
        from .cast import CastTo
        from .observer import DummyObserver, MinMaxObserver
        from .smoothquant import ActivationWeightSmoothQuant as SmoothQuant
        for name, mod in model.named_modules():
            if isinstance(mod, CastTo) and mod._flag("observer_enabled"):
                obs = mod.activation_post_process
                if calibrating and isinstance(obs, (MinMaxObserver, DummyObserver)) and getattr(obs, "process_group", None) is None:
                    continue
                if calibrating:
                    raise RuntimeError(f"GraphedForward(calibrating=True): the observer of {name or 'the model'} is a {type(obs).__name__}"
                                       f"{' with a process group' if getattr(obs, 'process_group', None) is not None else ''}; only MinMax "
                                       "observers without a process group run without the host")
                raise RuntimeError(f"GraphedForward: the observer of {name or 'the model'} is enabled; finish calibration (disable_observer) before capture")
            if isinstance(mod, SmoothQuant) and (getattr(mod, "calibrating", False) or mod._flag("dynamic")):
                raise RuntimeError(f"GraphedForward: SmoothQuant of {name or 'the model'} is calibrating / dynamic; capture a frozen scale")
        self.model = model
        self.static_in = [t.detach().clone() for t in example_inputs]
        dev = self.static_in[0].device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # un-folded weights: the sibling weights' chains as a few multi-tensor launches inside the captured forward (LiveWeightBatch),
        # installed AFTER the validation above and for the warm-up and the capture only (round 6): the user's model leaves
        # this constructor without hooks, whether it returns or raises, and a scope's quantised copies are released when its forward
        # ends (inside the capture that returns them to the graph's private pool, where the next scope's results reuse them)
        live_batch = LiveWeightBatch(model, max_bytes=live_batch_max_bytes) if batch_live_weights else None
        self.live_batch_scopes = len(live_batch.scopes) if live_batch is not None else 0
        try:
            with torch.no_grad(), torch.cuda.stream(side):
                for _ in range(max(1, warmup)):   # lazily created state (scores, allocator pools, rocBLAS workspaces) before the capture
                    model(*self.static_in)
                torch.cuda.synchronize(dev)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=side):
                    self.static_out = model(*self.static_in)
        finally:
            if live_batch is not None:
                live_batch.remove()
        torch.cuda.current_stream(dev).wait_stream(side)

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self.static_in):
            raise ValueError(f"GraphedForward: captured with {len(self.static_in)} inputs, called with {len(inputs)}")
        for dst, src in zip(self.static_in, inputs):
            if src.shape != dst.shape or src.dtype != dst.dtype or src.device != dst.device:
                raise ValueError(f"GraphedForward: captured for {tuple(dst.shape)} {dst.dtype} on {dst.device}, got {tuple(src.shape)} {src.dtype} on {src.device}")
            dst.copy_(src)
        self.graph.replay()
        return self.static_out


class GraphedModule:
    """`GraphedForward` per input SIGNATURE, behind the call syntax of the module (round 6): the eager forward of a configured layer is
    host-bound (an opt-125m decoder layer: 734 us eager, 176-219 us as a graph), and a serving loop sees a handful
    of shapes.

        fast = GraphedModule(layer)          # nothing is captured yet
        y = fast(x)                          # first call with this (shapes, dtypes, devices): warm-up + capture; then ONE replay per call
        y2 = fast(x_other_batch)             # another signature: its own graph (at most `max_graphs`, least recently used dropped)

    Results are CLONED out of the graph's static buffers by default (`clone_outputs=True`: a caller may keep them across calls; False hands
    out the static buffers, overwritten by the next call with the same signature).  Inference only, frozen configuration: after
    `configure()`, calibration or any other change of formats / flags / scales that are BAKED into launches as arguments (formats are;
    scale and zero-point TENSORS are read at replay time and may change), call `invalidate()` -- the captured launches carry the
    arguments of their capture.  Non-tensor and keyword arguments are not supported (use `GraphedForward` on a closure)."""

    def __init__(self, model: torch.nn.Module, clone_outputs: bool = True, max_graphs: int = 8, **graphed_forward_kwargs):
        self.model, self.clone_outputs, self.max_graphs, self._kw = model, clone_outputs, max_graphs, graphed_forward_kwargs
        self._graphs = OrderedDict()

    def invalidate(self):
        """drop every captured graph (the next call of each signature captures again)"""
        self._graphs.clear()

    def __call__(self, *inputs: torch.Tensor):
        key = tuple((tuple(t.shape), t.dtype, t.device) for t in inputs)
        g = self._graphs.get(key)
        if g is None:
            g = GraphedForward(self.model, *inputs, **self._kw)
            self._graphs[key] = g
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        out = g(*inputs)
        if not self.clone_outputs:
            return out
        if isinstance(out, torch.Tensor):
            return out.clone()
        if isinstance(out, (tuple, list)):
            return type(out)(o.clone() if isinstance(o, torch.Tensor) else o for o in out)
        return out


def configure_model(model: torch.nn.Module, *rules: DmxConfigRule):
    """DmxModel.configure(None, *rules) for a model already built from these modules (model.py:61-78)"""
    for r in rules:
        r.apply_to(model)
    return model
