"""Numerical formats — host-side mirror of the reference's `numerical/format.py` (d-matrix-ai/dmx-compressor).

Same vocabulary (shorthand strings, `repr()` round trip, class names, `cast(x, block_dim)`), different engine:
every `cast` is ONE libdmxq kernel launch over the whole tensor instead of the reference's
`.float() -> transpose -> split -> per-chunk native call -> cat -> transpose` loop (format.py:322-341).

Contract notes
  * `Format.cast(x, block_dim)` returns float32 like the reference (S2 seam); the fused
    `cast(x, block_dim, out_dtype=...)` form is what `CastTo` uses to skip the fp32 round trip.
  * The result is always a fresh contiguous tensor.  (The reference returns a transposed *view* for
    block_dim != -1; values and shape are identical, strides are not.)
  * Parameters for which the reference itself is undefined (negative shift counts in
    quant_cpu.cpp:211-237: FP mantissa = 23 through float_quantize, BFP precision > 22 with block > 1) are
    rejected with NotImplementedError, except FLOAT32 on a 16-bit tensor, which is the exact widening it is
    evidently meant to be.
"""
import re
from typing import Optional

import torch

from . import ops

__all__ = [
    "Format", "Same", "FixedPoint", "FloatingPoint", "BlockFloatingPoint", "ScaledBlockFloatingPoint", "MXFP",
    "MXINT", "ROUNDING_MODE",
]

# rounding letter <-> mode (format.py:23-30)
ROUNDING_MODE = {"U": "up", "D": "down", "N": "nearest", "S": "stochastic"}
_ROUNDING_LETTER = {v: k for k, v in ROUNDING_MODE.items()}


def _match(pattern: str, sh: str, what: str):
    m = re.fullmatch(pattern, sh.strip())
    if m is None:
        raise ValueError(f"unrecognized {what} shorthand: {sh!r}")
    return m


class Format:
    """Abstract tensor numerical format."""

    blocked: bool = False

    def cast(self, x: torch.Tensor, block_dim: int = -1, out_dtype: Optional[torch.dtype] = torch.float32):
        raise NotImplementedError

    @property
    def bytes_per_elem(self) -> Optional[float]:
        raise NotImplementedError

    @property
    def bit_precision(self) -> Optional[float]:
        raise NotImplementedError

    @staticmethod
    def from_shorthand(sh: str) -> "Format":
        if isinstance(sh, Format):
            return sh
        for prefix, cls in _PREFIXES:  # longest prefixes first (SBFP before ... BFP, MXFP/MXINT)
            if sh.startswith(prefix):
                return cls.from_shorthand(sh)
        raise ValueError(f"unrecognized format shorthand: {sh}")

    def __eq__(self, other):
        return isinstance(other, Format) and repr(self) == repr(other)

    def __hash__(self):
        return hash(repr(self))


class Same(Format):
    """No-op format: `cast` clones (format.py:89-90)."""

    def cast(self, x, block_dim: int = -1, out_dtype=None):
        return x.clone()

    @property
    def bytes_per_elem(self):
        return None

    @property
    def bit_precision(self):
        return None

    @classmethod
    def from_shorthand(cls, sh: str):
        return cls()

    def __str__(self):
        return "Dummy numerical format: no casting"

    def __repr__(self):
        return "SAME"


class FixedPoint(Format):
    """XP[precision,fraction](C|_ S|_ rounding) — fixed point simulated in fp32 (format.py:111-169)."""

    def __init__(self, precision, fraction, clamp=True, symmetric=True, rounding="nearest"):
        if not 1 <= precision <= 24:
            raise AssertionError(f"highest integer precision simulated by FP32 is 25, got {precision}")
        self.precision, self.fraction = precision, fraction
        self.clamp, self.symmetric, self.rounding = clamp, symmetric, rounding

    def cast(self, x, block_dim: int = -1, out_dtype=torch.float32):
        return ops.fixed_qdq(x, self.precision, self.fraction, self.clamp, self.symmetric, self.rounding,
                             out_dtype=out_dtype)

    @property
    def bytes_per_elem(self):
        return self.precision / 8.0

    @property
    def bit_precision(self):
        return float(self.precision)

    @classmethod
    def from_shorthand(cls, sh: str):
        m = _match(r"XP\[(\d+),([+-]?\d+)\]\(([C_])([S_])([UDNS])\)", sh, "fixed point")
        return cls(int(m[1]), int(m[2]), m[3] == "C", m[4] == "S", ROUNDING_MODE[m[5]])

    def __str__(self):
        return (f"Simulated fixed point format: precision bits = {self.precision}, fraction bits = {self.fraction}, "
                f"\ncasting behavior: symmetric = {self.symmetric}, clamp = {self.clamp}, rounding = {self.rounding}")

    def __repr__(self):
        frac = "0" if self.fraction == 0 else f"{self.fraction:+d}"
        return (f"XP[{self.precision},{frac}]({'C' if self.clamp else '_'}{'S' if self.symmetric else '_'}"
                f"{_ROUNDING_LETTER[self.rounding]})")


class FloatingPoint(Format):
    """FP[sign|exponent|mantissa,bias](F|_ rounding) — low-bit float simulated in fp32 (format.py:172-270)."""

    def __init__(self, mantissa=23, exponent=8, bias=None, flush_subnormal=True, unsigned=False, rounding="nearest"):
        if not 0 <= mantissa <= 23:
            raise AssertionError(f"number of mantisa bits simulatable by FP32 is between 0 and 23, got{mantissa}")
        if not 0 < exponent <= 8:
            raise AssertionError(f"number of exponent bits simulatable by FP32 is between 1 and 8, got {exponent}")
        if bias is None:
            bias = 2 ** (exponent - 1) - 1
        lo = 127 if exponent == 8 else -128 + 2 ** exponent
        if not lo <= bias <= 127:
            raise AssertionError(
                f"exponent bias simulatable by FP32 for {exponent}-bit exponent is constrained between {lo} and 127, got {bias}")
        self.mantissa, self.exponent, self.bias = mantissa, exponent, bias
        self.flush_subnormal, self.unsigned, self.rounding = flush_subnormal, unsigned, rounding

    def native_of(self):
        """torch.float32 for FP[1|8|23,127](_N), torch.float16 for FP[1|5|10,15](_N) (the two formats the reference passes through
        untouched when the tensor already has that dtype, format.py:209-212), else None.  Field compares: this sits on the hot path
        of every cast call and building the shorthand string costs more than the kernel launch it guards."""
        if self.flush_subnormal or self.unsigned or self.rounding != "nearest":
            return None
        if self.mantissa == 23 and self.exponent == 8 and self.bias == 127:
            return torch.float32
        if self.mantissa == 10 and self.exponent == 5 and self.bias == 15:
            return torch.float16
        return None

    def cast(self, x, block_dim: int = -1, out_dtype=torch.float32):
        native = self.native_of()
        # format.py:209-212: native formats pass the input tensor object through untouched
        if native is not None and x.dtype == native:
            return x
        if self.mantissa == 23:
            if native == torch.float32:  # FLOAT32 applied to a 16-bit tensor: exact widening (reference: UB)
                return x.to(out_dtype or torch.float32)
            raise NotImplementedError(f"{self!r}: a 23-bit mantissa through float_quantize is undefined in the reference")
        # the extra fp16 subnormal flush of format.py:222-232 is implied by flush_subnormal on this path
        return ops.float_qdq(x, self.mantissa, self.exponent, self.bias, self.flush_subnormal, self.unsigned,
                             self.rounding, out_dtype=out_dtype)

    @property
    def largest_representable_power_of_two(self):
        return 2 ** (2 ** (self.exponent - 1))

    @property
    def bytes_per_elem(self):
        return (self.mantissa + self.exponent + 1) / 8.0

    @property
    def bit_precision(self):
        return float(self.mantissa + self.exponent + (0 if self.unsigned else 1))

    @classmethod
    def from_shorthand(cls, sh: str):
        m = _match(r"FP\[(\d+)\|(\d+)\|(\d+),([+-]?\d+)\]\(([F_])([UDNS])\)", sh, "floating point")
        return cls(mantissa=int(m[3]), exponent=int(m[2]), bias=int(m[4]), flush_subnormal=m[5] == "F",
                   unsigned=int(m[1]) == 0, rounding=ROUNDING_MODE[m[6]])

    def __str__(self):
        return (f"Simulated floating point format: mantissa bits = {self.mantissa}, exponent bits = {self.exponent}, "
                f"exponent bias = {self.bias}, unsigned = {self.unsigned}, \ncasting behavior: flush subnormal = "
                f"{self.flush_subnormal}, rounding = {self.rounding}")

    def __repr__(self):
        return (f"FP[{'0' if self.unsigned else '1'}|{self.exponent}|{self.mantissa},{self.bias}]"
                f"({'F' if self.flush_subnormal else '_'}{_ROUNDING_LETTER[self.rounding]})")


class BlockFloatingPoint(Format):
    """BFP[precision|8]{block_size}(S|_ rounding) — shared 8-bit exponent per block (format.py:273-397)."""

    blocked = True

    def __init__(self, precision=8, block_size=64, symmetric=True, rounding="nearest"):
        if not 2 <= precision <= 25:
            raise AssertionError(f"highest integer precision simulated by FP32 is 25, got {precision}")
        if block_size <= 0:
            raise AssertionError(f"block size has to be positive, got {block_size}")
        self.precision, self.block_size = precision, block_size
        self.symmetric, self.rounding = symmetric, rounding

    def cast(self, x, block_dim: int = -1, out_dtype=torch.float32):
        # block_size == 1 (e.g. BFP32_1 bias format) is routed to the float kernel inside the library,
        # like format.py:312-320; the asymmetric post-pass of format.py:337-339 is a kernel flag.
        sym = True if self.block_size == 1 else self.symmetric
        return ops.bfp_qdq(x, self.precision, self.block_size, block_dim, sym, self.rounding, out_dtype=out_dtype)

    @property
    def bytes_per_elem(self):
        return (self.precision + 8.0 / self.block_size) / 8.0

    @property
    def bit_precision(self):
        return self.precision + 8.0 / self.block_size

    @classmethod
    def from_shorthand(cls, sh: str):
        m = _match(r"BFP\[(\d+)\|8\]\{(\d+)\}\(([S_])([UDNS])\)", sh, "block floating point")
        return cls(int(m[1]), int(m[2]), m[3] == "S", ROUNDING_MODE[m[4]])

    @staticmethod
    def parse_legacy(sh: str):
        """Legacy `BFP[p|8]{B,d}(X)` form of docs/numerics.rst:64-80 and configs/*.yaml, where the block
        dimension lived inside the braces.  Returns (format, block_dim)."""
        m = _match(r"BFP\[(\d+)\|8\]\{(\d+),\s*(-?\d+)\}\(([S_])([UDNS])\)", sh, "legacy block floating point")
        return BlockFloatingPoint(int(m[1]), int(m[2]), m[4] == "S", ROUNDING_MODE[m[5]]), int(m[3])

    def __str__(self):
        return (f"Simulated block floating point format: precision bits = {self.precision}, block size = "
                f"{self.block_size}\ncasting behavior: symmetric = {self.symmetric}, rounding = {self.rounding}")

    def __repr__(self):
        return (f"BFP[{self.precision}|8]{{{self.block_size}}}({'S' if self.symmetric else '_'}"
                f"{_ROUNDING_LETTER[self.rounding]})")


class MXINT(BlockFloatingPoint):
    """MXINT{p}{{B}} = BFP[p|8]{B}(SN) (format.py:612-653)."""

    def __init__(self, precision=8, block_size=32):
        super().__init__(precision=precision, block_size=block_size, symmetric=True, rounding="nearest")

    @classmethod
    def from_shorthand(cls, sh: str):
        m = _match(r"MXINT(\d+)\{(\d+)\}", sh, "MXINT")
        return cls(int(m[1]), int(m[2]))

    def __str__(self):
        return (f"Simulated MXINT format: precision bits = {self.precision}, block size = {self.block_size}\n"
                f"casting behavior: symmetric = {self.symmetric}, rounding = {self.rounding}")

    def __repr__(self):
        return f"MXINT{self.precision}{{{self.block_size}}}"

    def __reduce__(self):
        return (self.__class__, (self.precision, self.block_size))


class ScaledBlockFloatingPoint(Format):
    """SBFP<XP..><FP..>{B}: integer block elements times a low-bit float scaler (format.py:400-511).
    One fused kernel (csrc/blockfmt.hip)."""

    blocked = True

    def __init__(self, block_format: FixedPoint, scaler_format: FloatingPoint, block_size=64):
        assert isinstance(block_format, FixedPoint), "block format needs to be fixed point"
        assert isinstance(scaler_format, FloatingPoint), "scaler format needs to be floating point"
        assert block_format.fraction == 0, "block format needs to have zero fraction"
        assert block_format.symmetric, "block format needs to have symmetric range"
        assert block_size > 0, f"block size has to be positive, got {block_size}"
        self.block_format, self.scaler_format, self.block_size = block_format, scaler_format, block_size
        self.man_scaling = 2 ** (block_format.precision - 1) - 1

    def cast(self, x, block_dim: int = -1, out_dtype=torch.float32):
        bf, sf = self.block_format, self.scaler_format
        if bf.rounding != "nearest" or sf.rounding != "nearest":
            raise NotImplementedError("SBFP: only nearest rounding (every SBFP alias of the reference) has a fused kernel")
        return ops.sbfp_qdq(x, bf.precision, self.block_size, sf.mantissa, sf.exponent, sf.bias, sf.flush_subnormal,
                            bf.clamp, bf.symmetric, block_dim, out_dtype=out_dtype)

    @property
    def bytes_per_elem(self):
        return self.block_format.bytes_per_elem + self.scaler_format.bytes_per_elem / self.block_size

    @property
    def bit_precision(self):
        return self.block_format.bit_precision + self.scaler_format.bit_precision / self.block_size

    @classmethod
    def from_shorthand(cls, sh: str):
        m = _match(r"SBFP<(.+?)><(.+?)>\{(\d+)\}", sh, "SBFP")
        return cls(FixedPoint.from_shorthand(m[1]), FloatingPoint.from_shorthand(m[2]), int(m[3]))

    def __str__(self):
        return (f"Simulated scaled block floating point format: block format = {self.block_format}, scaler format = "
                f"{self.scaler_format},\n block size = {self.block_size}")

    def __repr__(self):
        return f"SBFP<{self.block_format!r}><{self.scaler_format!r}>{{{self.block_size}}}"


class MXFP(Format):
    """MXFP{p}[E{e}M{m}]{B}: low-bit float elements with a power-of-two (E8M0) block scale (format.py:514-609).
    One fused kernel (csrc/blockfmt.hip); blocks run along `block_dim` as for BFP (the reference's
    `cat(dim=block_dim)` slip is not reproduced) and an all-zero block stays zero (reference: NaN via log2(0))."""

    blocked = True

    def __init__(self, element_format: FloatingPoint, block_size=32):
        assert isinstance(element_format, FloatingPoint), "block format needs to be floating point"
        assert block_size > 0, f"block size has to be positive, got {block_size}"
        self.element_format = element_format
        self.scaler_format = FloatingPoint(mantissa=0, exponent=8, bias=127, unsigned=True)
        self.block_size = block_size

    def cast(self, x, block_dim: int = -1, out_dtype=torch.float32):
        ef = self.element_format
        return ops.mxfp_qdq(x, ef.mantissa, ef.exponent, self.block_size, block_dim, out_dtype=out_dtype)

    @property
    def bytes_per_elem(self):
        return self.element_format.bytes_per_elem + self.scaler_format.bytes_per_elem / self.block_size

    @property
    def bit_precision(self):
        return self.element_format.bit_precision + 8.0 / self.block_size

    @classmethod
    def from_shorthand(cls, sh: str):
        m = _match(r"MXFP(\d+)\[E(\d+)M(\d+)\]\{(\d+)\}", sh, "MXFP")
        p, e, man, b = (int(m[i]) for i in (1, 2, 3, 4))
        assert p == e + man + 1
        return cls(FloatingPoint(mantissa=man, exponent=e, bias=2 ** (e - 1) - 1, flush_subnormal=False,
                                 unsigned=False, rounding="nearest"), b)

    def __str__(self):
        return (f"Simulated MXFP format: element format = {self.element_format}, scaler format = "
                f"{self.scaler_format},\n block size = {self.block_size}")

    def __repr__(self):
        ef = self.element_format
        return f"MXFP{ef.exponent + ef.mantissa + 1}[E{ef.exponent}M{ef.mantissa}]{{{self.block_size}}}"

    def __reduce__(self):
        return (self.__class__, (self.element_format, self.block_size))


_PREFIXES = [("SAME", Same), ("SBFP", ScaledBlockFloatingPoint), ("MXFP", MXFP), ("MXINT", MXINT),
             ("XP", FixedPoint), ("FP", FloatingPoint), ("BFP", BlockFloatingPoint)]
