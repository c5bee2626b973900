"""dmx_compressor_amd — MI355X (gfx950) native fake-quantisation / sparsity operator library.

A drop-in for the hot path of d-matrix-ai/dmx-compressor (CastTo -> Format.cast -> {block,float,fixed}_quantize,
the N:M mask, SmoothQuant scaling, the exact-function approximator slot): same vocabulary, hand-written HIP
kernels behind the C ABI of include/dmxq.h.  There is no CPU fallback: ops raise without libdmxq.so / a GPU.
"""
from types import SimpleNamespace

from . import ops, quant
from ._lib import DmxqError, LIB_PATH
from .cast import CastTo, CastToDict, CastToFormat
from .format import (ROUNDING_MODE, BlockFloatingPoint, FixedPoint, FloatingPoint, Format, MXFP, MXINT, Same,
                     ScaledBlockFloatingPoint)
from . import nn
from .approximate import Approximate, ApproximationFunction, NoApproximation, TorchFunctionApproximation
from .nn import (DmxConfigRule, DmxModule, DmxModuleQuantizerCalibrationHyperparams, DmxModuleSmoothQuantHyperparams,
                 DmxQuantizerCalibrationHyperparams, configure_model)
from .observer import DummyObserver, HistogramObserver, MinMaxObserver, PercentileObserver
from .smoothquant import ActivationWeightSmoothQuant
from .config import apply_legacy_config, load_legacy_config
from .sparse import Bernoulli, BlockTopK, Dense, Sparseness, Sparsify, TopK

__version__ = "0.1.0"


def _format_aliases():
    """Alias vocabulary of the reference (src/dmx/compressor/__init__.py:20-97), generated rather than listed.
    `BFPnn` means an (nn-8)-bit mantissa + 8-bit shared exponent.  Reference quirk kept:
    BFP16A_16 is defined with a 6-bit mantissa there."""
    a = dict(
        SAME="SAME", FLOAT32="FP[1|8|23,127](_N)", FLOAT16="FP[1|5|10,15](FN)", BFLOAT16="FP[1|8|7,127](FN)",
        AFLOAT8="FP[1|4|3,7](_N)", BFLOAT8="FP[1|5|2,15](_N)", INT8="XP[8,0](CSN)", INT4="XP[4,0](CSN)",
        BFP32_1="BFP[24|8]{1}(SN)",
    )
    for b in (64, 32, 16):
        a[f"BFP24_{b}"] = f"BFP[16|8]{{{b}}}(SN)"
    for total, man in ((16, 8), (14, 6), (12, 4)):
        for b in (128, 64, 32, 16):
            a[f"BFP{total}_{b}"] = f"BFP[{man}|8]{{{b}}}(SN)"
            a[f"BFP{total}A_{b}"] = f"BFP[{man}|8]{{{b}}}(_N)"
    a["BFP16A_16"] = "BFP[6|8]{16}(_N)"
    a["SBFP12_16"] = "SBFP<XP[4,0](CSN)><FP[0|4|4,7](FN)>{16}"
    for bias in range(4, 19):
        a[f"SBFP12_16_{bias}"] = f"SBFP<XP[4,0](CSN)><FP[0|4|4,{bias}](FN)>{{16}}"
    for p, e, m in ((8, 4, 3), (8, 5, 2), (6, 2, 3), (6, 3, 2), (4, 2, 1)):
        for b in (128, 64, 32):
            a[f"MXFP{p}_E{e}M{m}K{b}"] = f"MXFP{p}[E{e}M{m}]{{{b}}}"
    for p in (8, 6, 4):
        for b in (128, 64, 32):
            a[f"MXINT{p}_K{b}"] = f"MXINT{p}{{{b}}}"
    return SimpleNamespace(**{k: Format.from_shorthand(v) for k, v in a.items()})


format = _format_aliases()

# sparseness aliases (reference __init__.py:100-105): LD = last dim, FD = dim 1
sparseness = SimpleNamespace(**{
    f"BTK8_{k}_{tag}": Sparseness.from_shorthand(f"BTOPK{{{k}:8,{dim}}}(U)")
    for k in (4, 2) for tag, dim in (("LD", -1), ("FD", 1))
})


# default approximation functions (reference __init__.py:108-139).  The reference's defaults name the private "vsimd"
# algorithm and collapse to NONE when it is absent — the state of the public repository, mirrored here.
default_approx = SimpleNamespace(**{k: ApproximationFunction.from_shorthand("NONE") for k in (
    "RELU", "RELU6", "SILU", "SOFTMAX", "GELU", "QUICK_GELU", "TANH", "BATCH_NORM_2D", "LAYER_NORM", "RMS_NORM",
    "GROUP_NORM", "EXP", "APPLY_LLAMA_ROPE", "NONE")})


def _rule_sets():
    """config_rules.{BASELINE, BASIC, FP8, SBFP_WEIGHT_STORAGE} (reference __init__.py:142-483) for the module types of
    `dmx_compressor_amd.nn`.  Each row: module types -> (input formats, weight, bias, output formats)."""
    f = format
    S, F16, B64 = f.SAME, f.FLOAT16, f.BFP16_64
    conv_like = (nn.Conv1d, nn.Conv2d, nn.ConvTranspose2d)
    act_like = (nn.Softmax, nn.LayerNorm, nn.GELU, nn.ReLU, nn.SiLU, nn.QuickGELU, nn.RMSNorm, nn.Exp, nn.ReLU6, nn.Tanh, nn.NewGELU,
                nn.FastGELU, nn.BloomGELU, nn.ClippedGELU, nn.BatchNorm2d, nn.GroupNorm)
    pools = (nn.MaxPool2d, nn.AvgPool2d, nn.AdaptiveAvgPool2d)

    def wb(inp, w, b, out):
        return dict(input_formats=[inp], weight_format=w, bias_format=b, output_formats=[out])

    def io(n_in, inp, out, approx=None):
        d = dict(input_formats=[inp] * n_in, output_formats=[out])
        if approx is not None:
            d["approximation_function"] = approx
        return d

    def build(lin, mm_in, elt):
        return [
            DmxConfigRule(module_types=(nn.Linear,), module_config=wb(*lin)),
            DmxConfigRule(module_types=conv_like, module_config=wb(*lin)),
            DmxConfigRule(module_types=(nn.ResAdd,), module_config=io(2, elt, elt)),
            DmxConfigRule(module_types=(nn.ActActMatMul,), module_config=io(2, mm_in, elt)),
            DmxConfigRule(module_types=(nn.Embedding,), module_config=dict(output_formats=[elt])),
            DmxConfigRule(module_types=pools, module_config=io(1, elt, elt)),
            DmxConfigRule(module_types=act_like, module_config=io(1, elt, elt, default_approx.NONE)),
            DmxConfigRule(module_types=(nn.ApplyRotaryPosEmb,), module_config=dict(input_formats=[elt] * 4, output_formats=[elt] * 2,
                                                                                   approximation_function=default_approx.NONE)),
        ]

    return SimpleNamespace(
        BASELINE=build((S, S, S, S), S, S),
        BASIC=build((B64, B64, f.BFP32_1, F16), B64, F16),
        FP8=build((f.AFLOAT8, f.AFLOAT8, f.FLOAT32, F16), f.AFLOAT8, F16),
        SBFP_WEIGHT_STORAGE=[DmxConfigRule(module_types=(nn.Linear,) + conv_like,
                                           module_config=dict(weight_storage_format=f.SBFP12_16))],
    )


config_rules = _rule_sets()
