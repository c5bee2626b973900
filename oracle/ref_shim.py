"""oracle/ref_shim.py — TEST INFRASTRUCTURE, THIS CONTAINER ONLY.

Imports the *Python* reference (d-matrix-ai/dmx-compressor, mounted read-only at /root/reference) so that
`oracle/gen_golden.py` can (1) validate this repo's C restatement (`oracle/oracle.c`) against the reference's
own code path and (2) emit the small input/expected-output fixtures committed under `tests/golden/`.

Nothing in `tests/ -m gpu`, `bench.py`, `__graft_entry__.smoke()` or the product package imports this module:
/root/reference does not exist on the GPU box and the reference's Python cannot travel in any form.

The reference depends on a few pure-Python third-party modules that are not installed in this image
(`bidict`, `parse`, `pptree`, `skopt`, `graphviz`, `evaluate`, and `transformers.utils.fx`, removed in the
installed transformers).  None of them performs tensor arithmetic: `parse` only decodes the shorthand strings
("BFP[8|8]{16}(SN)"), `bidict` is a two-way dict for the rounding-letter table.  Minimal stand-ins are
registered in `sys.modules` BEFORE the reference is imported, exactly as SURVEY.md Appendix E describes.
The numerical path being pinned (format.py -> quant_function.py -> quant_cpu/*.cpp) is the reference's own.
"""
import os
import re
import sys
import types

REF_ROOT = os.environ.get("DMX_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")

# never write bytecode into the read-only reference tree
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("TORCH_EXTENSIONS_DIR", "/tmp/dmx_ref_torch_ext")


def available() -> bool:
    return os.path.isdir(REF_SRC)


# --------------------------------------------------------------------------------------------- stand-ins
class _Bidict(dict):
    @property
    def inverse(self):
        return {v: k for k, v in self.items()}


_TYPE_RE = {"d": r"[-+]?\d+", "w": r"\w+", "l": r"[A-Za-z]+", "f": r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?"}
_CONV = {"d": int, "f": float}


def _parse(fmt, string, extra_types=None):
    """Regex re-implementation of the subset of `parse.parse` the reference's shorthand grammars use:
    `{name:d}` `{name:w}` `{name:l}` `{name:f}` `{name}` and `{{` `}}` escapes; custom types with `.pattern`."""
    extra_types = extra_types or {}
    out, i, conv = "", 0, {}
    while i < len(fmt):
        c = fmt[i]
        if fmt.startswith("{{", i):
            out += re.escape("{")
            i += 2
        elif fmt.startswith("}}", i):
            out += re.escape("}")
            i += 2
        elif c == "{":
            j = fmt.index("}", i)
            field = fmt[i + 1 : j]
            name, _, typ = field.partition(":")
            if typ in extra_types:
                pat = getattr(extra_types[typ], "pattern", r".+?")
                conv[name] = extra_types[typ]
            elif typ in _TYPE_RE:
                pat = _TYPE_RE[typ]
                if typ in _CONV:
                    conv[name] = _CONV[typ]
            else:
                pat = r".+?" if j + 1 < len(fmt) else r".+"
            out += f"(?P<{name}>{pat})"
            i = j + 1
        else:
            out += re.escape(c)
            i += 1
    m = re.fullmatch(out, string)
    if m is None:
        return None
    d = m.groupdict()
    for k, f in conv.items():
        d[k] = f(d[k])
    return d


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_standins():
    import torch
    import torch.fx

    if "bidict" not in sys.modules:
        _module("bidict", bidict=_Bidict)
    if "parse" not in sys.modules:
        _module("parse", parse=_parse)
    for name in ("pptree", "evaluate"):
        if name not in sys.modules:
            _module(name, print_tree=lambda *a, **k: None, evaluator=lambda *a, **k: None, load=lambda *a, **k: None)
    if "graphviz" not in sys.modules:
        _module("graphviz", Digraph=type("Digraph", (), {"__init__": lambda self, *a, **k: None}))
    if "skopt" not in sys.modules:
        _module("skopt", gp_minimize=lambda *a, **k: None)
        _module("skopt.space", Space=object, Categorical=object)
        _module("skopt.utils", use_named_args=lambda *a, **k: (lambda f: f))
    import transformers  # noqa: F401  (present in the image)
    import transformers.utils

    if "transformers.utils.fx" not in sys.modules:
        try:
            import transformers.utils.fx  # noqa: F401
        except Exception:

            class HFTracer(torch.fx.Tracer):
                def __init__(self, autowrap_modules=(), autowrap_functions=()):
                    super().__init__()

                def trace(self, root, concrete_args=None, dummy_inputs=None, **kw):
                    return super().trace(root, concrete_args=concrete_args)

            fx = _module("transformers.utils.fx", HFTracer=HFTracer, get_concrete_args=lambda *a, **k: {})
            transformers.utils.fx = fx
    import transformers.modeling_utils as mu

    if not hasattr(mu, "ModelOutput"):
        from transformers.utils.generic import ModelOutput

        mu.ModelOutput = ModelOutput


_REF = None


def load_reference():
    """Returns the imported `dmx.compressor` package of the reference (JIT-builds its quant_cpu on first use)."""
    global _REF
    if _REF is not None:
        return _REF
    if not available():
        raise RuntimeError(f"reference not present at {REF_SRC}: golden generation only runs in the build container")
    _install_standins()
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import dmx.compressor as ref

    _REF = ref
    return ref
