"""Approximator slot — mirror of the reference's `functional/approximate.py` containers.

`FUNC[algorithm]{wrapper_kwargs}(extra_kwargs)` shorthands are parsed like the reference
(approximate.py:128-132).  The reference's only algorithm is "vsimd", a private package that is absent from the
public repository, where every default therefore collapses to NONE.  This mirror registers two algorithms:
  "dmxq"          the exact torch function contract evaluated by libdmxq's HIP kernels (GELU, SILU, QUICK_GELU, EXP,
                  SOFTMAX, LAYER_NORM, RMS_NORM, APPLY_LLAMA_ROPE) -- what the reference computes with vsimd absent;
  "experimental"  the reference's one in-repo approximation, `experimental.silu` (functional/functions.py:7-21,
                  dispatched by approximate.py:148-151), reproduced bit for bit by a HIP kernel.
vsimd approximation arithmetic itself is parity-unpinned and is not invented here.
"""
import ast
import re
from typing import Any, Dict, Union

import torch

from . import ops

__all__ = ["ApproximationFunction", "NoApproximation", "TorchFunctionApproximation", "Approximate"]

_FUNCS = ("GELU", "SILU", "RMS_NORM", "LAYER_NORM", "SOFTMAX", "EXP", "QUICK_GELU", "APPLY_LLAMA_ROPE")


def _parse_kwargs(s: str) -> Dict[str, Any]:
    out = {}
    for item in filter(None, (t.strip() for t in s.split(","))):
        k, v = item.split("=")
        try:
            out[k.strip()] = ast.literal_eval(v.strip())
        except Exception:
            out[k.strip()] = v.strip()
    return out


def _kwargs_str(d: Dict[str, Any]) -> str:
    return ",".join(f"{k}={v}" for k, v in d.items())


class ApproximationFunction:
    def execute(self, *args, **kwargs):
        raise NotImplementedError

    @staticmethod
    def from_shorthand(sh: str):
        if isinstance(sh, ApproximationFunction):
            return sh
        if sh.startswith("NONE"):
            return NoApproximation()
        if sh.startswith(_FUNCS):
            return TorchFunctionApproximation.from_shorthand(sh)
        raise ValueError(f"unrecognized approximation function shorthand: {sh}")


class NoApproximation(ApproximationFunction):
    wrapper_params: Dict[str, Any] = {}

    def execute(self, *args, **kwargs):
        raise RuntimeError("NoApproximation is not supposed to be executed")

    def __str__(self):
        return "Dummy approximation function: no approximation"

    def __repr__(self):
        return "NONE"


class TorchFunctionApproximation(ApproximationFunction):
    def __init__(self, func_id: str, algorithm: str = "dmxq", wrapper_params=None, extra_params=None):
        self.func_id, self.algorithm = func_id, algorithm
        self.wrapper_params, self.extra_params = dict(wrapper_params or {}), dict(extra_params or {})

    def execute(self, *args, **kwargs):
        kw = {**kwargs, **self.extra_params}
        if self.algorithm == "experimental":
            # approximate.py:148-151: eval(f"experimental.{func_name}")(*args, **kwargs, **extra_params)
            if self.func_id != "SILU":
                raise AttributeError(f"experimental has no approximation of {self.func_id} (functional/functions.py defines silu only)")
            assert not kw.get("inplace", False), "inplace has to be False, not functionally meaningful anyway"
            return ops.silu_experimental(args[0], kw["scale"])
        if self.algorithm != "dmxq":
            raise NotImplementedError(
                f"approximation algorithm {self.algorithm!r}: 'dmxq' (exact function on the HIP kernels) and 'experimental' "
                "exist here; the reference's 'vsimd' arithmetic lives in a private package (parity unpinned)")
        if self.func_id == "GELU":
            return ops.gelu(args[0], approximate=kw.get("approximate", "none"))
        if self.func_id == "SILU":
            return ops.silu(args[0])
        if self.func_id == "QUICK_GELU":
            return ops.quick_gelu(args[0])
        if self.func_id == "EXP":
            return ops.exp(args[0])
        if self.func_id == "SOFTMAX":
            return ops.softmax(args[0], dim=kw.get("dim", -1))
        if self.func_id == "LAYER_NORM":
            x, normalized_shape = args[0], args[1]
            w = args[2] if len(args) > 2 else kw.get("weight")
            b = args[3] if len(args) > 3 else kw.get("bias")
            eps = args[4] if len(args) > 4 else kw.get("eps", 1e-5)
            return ops.layernorm(x, normalized_shape, w, b, eps)
        if self.func_id == "RMS_NORM":
            x, normalized_shape = args[0], args[1]
            w = args[2] if len(args) > 2 else kw.get("weight")
            eps = args[3] if len(args) > 3 else kw.get("eps")
            return ops.rmsnorm(x, normalized_shape, w, eps)
        if self.func_id == "APPLY_LLAMA_ROPE":
            # custom_modules.py:142-172 forward(q, k, cos, sin, unsqueeze_dim=1) -> (q_embed, k_embed)
            q, k, cos, sin = args[:4]
            ud = args[4] if len(args) > 4 else kw.get("unsqueeze_dim", 1)
            qe, ke = ops.rope(q, cos, sin, ud), ops.rope(k, cos, sin, ud)
            if qe is None or ke is None:
                raise NotImplementedError("APPLY_LLAMA_ROPE[dmxq]: this shape / dtype mix is not taken by the HIP kernel "
                                          "(needs 4-d q / k, 3-d cos / sin of the same dtype, head_dim a multiple of 16)")
            return qe, ke
        raise NotImplementedError(f"{self.func_id}: unknown function id")

    @classmethod
    def from_shorthand(cls, sh: str):
        m = re.fullmatch(r"(\w+)\[(\w+)\]\{(.*)\}\((.*)\)", sh)
        if m is None:
            raise ValueError(f"unrecognized approximation function shorthand: {sh}")
        return cls(m[1], m[2], _parse_kwargs(m[3]), _parse_kwargs(m[4]))

    def __str__(self):
        return f"Approximated version of {self.func_id}: algorithm = {self.algorithm}, with extra_params = {self.extra_params}"

    def __repr__(self):
        return f"{self.func_id}[{self.algorithm}]{{{_kwargs_str(self.wrapper_params)}}}({_kwargs_str(self.extra_params)})"


class Approximate(torch.nn.Module):
    """approximation operator container (approximate.py:229-247)"""

    def __init__(self, function=None):
        super().__init__()
        self.set_function(function or NoApproximation())

    def set_function(self, function: Union[str, ApproximationFunction]) -> None:
        self.function = ApproximationFunction.from_shorthand(function)

    def forward(self, input, *args, **kwargs):
        return self.function.execute(input, *args, **kwargs)

    def extra_repr(self):
        return f"function = {self.function!r}"
