"""Config loading for the module mirror.

Current schema = the `DmxModuleConfig` keys `DmxModule.configure` reads (reference modeling/nn/core.py:73-108):
input_formats, output_formats, accum_format, weight_storage_format, weight_format, bias_format,
smoothquant_scale_format, weight_sparseness, approximation_function, pre_*_transform.

Legacy schema = the reference's stale example files (configs/dmx_example_config_lenet5.yaml, docs/numerics.rst:64-95):
bare module names, singular `input_format` / `output_format`, an `instance` class name, and BFP shorthands that carry
the block dimension inside the braces, `BFP[8|8]{64,1}(SN)`.  Under the reference's current code such a file is a
SILENT NO-OP (names lack the `_gm.` prefix and the singular keys are never read), so
BASELINE.json's config 1 really is the unquantised model.  `load_legacy_config` additionally offers the file's
evident INTENT: the same entries translated to the current schema, with the block dimension moved onto the CastTo.
"""
from typing import Dict, Tuple, Union

import torch

from .format import BlockFloatingPoint, Format
from .nn import DmxModule

_LEGACY_SINGULAR = {"input_format": "input_formats", "output_format": "output_formats"}
_FORMAT_KEYS = ("accum_format", "weight_storage_format", "weight_format", "bias_format", "smoothquant_scale_format")


def _fmt(sh) -> Tuple[Format, Union[int, None]]:
    """shorthand (current or legacy `{B,d}`) -> (format, block_dim or None)"""
    if isinstance(sh, Format):
        return sh, None
    if isinstance(sh, str) and sh.startswith("BFP") and "," in sh[sh.index("{"): sh.index("}")]:
        return BlockFloatingPoint.parse_legacy(sh)
    return Format.from_shorthand(sh), None


def load_legacy_config(src: Union[str, dict]) -> Dict[str, dict]:
    """yaml path / yaml text / dict in the legacy schema -> {module name: {"config": current-schema dict,
    "block_dims": {cast name: dim}, "instance": class name}}"""
    if isinstance(src, str):
        import yaml

        text = open(src).read() if "\n" not in src and src.endswith((".yaml", ".yml")) else src
        src = yaml.safe_load(text)
    out = {}
    for name, entry in (src or {}).items():
        cfg, dims = {}, {}
        for k, v in entry.items():
            if k == "instance":
                continue
            if k in _LEGACY_SINGULAR:
                f, d = _fmt(v)
                cfg[_LEGACY_SINGULAR[k]] = [f]
                if d is not None:
                    dims["input_cast" if k == "input_format" else "output_cast"] = d
            elif k in _FORMAT_KEYS:
                f, d = _fmt(v)
                cfg[k] = f
                if d is not None:
                    dims[k.replace("_format", "_cast")] = d
            else:
                cfg[k] = v
        out[name] = {"config": cfg, "block_dims": dims, "instance": entry.get("instance")}
    return out


def apply_legacy_config(model: torch.nn.Module, src: Union[str, dict], strict_instance: bool = True) -> int:
    """Applies the INTENT of a legacy config to a model built from dmx_compressor_amd.nn modules (bare names, as in the
    file).  Returns the number of modules configured."""
    cfg = load_legacy_config(src) if not (isinstance(src, dict) and all("config" in v for v in src.values())) else src
    n = 0
    for name, m in model.named_modules():
        if name in cfg and isinstance(m, DmxModule):
            e = cfg[name]
            if strict_instance and e["instance"] and type(m).__name__ != e["instance"]:
                raise TypeError(f"{name}: config says {e['instance']}, module is {type(m).__name__}")
            m.configure(e["config"])
            for cast_name, d in e["block_dims"].items():
                cast = (m.input_casts[cast_name] if cast_name in m.input_casts else
                        m.output_casts[cast_name] if cast_name in m.output_casts else getattr(m, cast_name, None))
                if cast is not None:
                    cast.block_dim = d
            n += 1
    return n
