"""Portable synthetic inputs: a counter-based generator (splitmix64 on the linear index -> Box-Muller in fp64)
so the build container and the GPU box produce identical bits without shipping tensors."""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(idx, seed):
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx.astype(np.uint64) + np.uint64(1)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(n, seed, start=0):
    """float64 in (0, 1); element i of the stream is a function of (seed, start + i) only"""
    r = splitmix64(np.arange(start, start + n, dtype=np.uint64), seed)
    return ((r >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def normal(n, seed, start=0):
    u1, u2 = uniform(n, seed, start), uniform(n, seed ^ 0xABCDEF1234567, start)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def make_chunked(kind, shape, seed=0, dtype=torch.float32, workers=None, chunk=1 << 20, start=0):
    """make("normal" | "heavy", ...) computed in index chunks on a thread pool (numpy releases the GIL): the SAME bits
    as make() -- every element is a function of (seed, linear index) -- at a fraction of the time for the 16.8 M-element
    benchmark tensors (bench.py generates 20 of them per rank; tests/test_abi_and_host.py checks the equality).
    kind "uniform": U(0, 1) (the N:M scores of bench.py's llama-shard workload).  start: the linear index of the first element --
    a ROW SHARD of a larger tensor is make_chunked(kind, (rows_of_the_shard, cols), seed, start = first_row * cols): every rank of a
    sharded run generates only its own rows, and the bits are those of the whole tensor's rows (round 6)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    if kind not in ("normal", "heavy", "uniform"):
        raise ValueError(kind)
    n = int(np.prod(shape))
    out = torch.empty(n, dtype=dtype)

    def one(off):
        m = min(chunk, n - off)
        if kind == "uniform":
            v = uniform(m, seed, start + off)
        else:
            v = normal(m, seed, start + off)
            if kind == "heavy":
                v = v * np.exp(4.0 * normal(m, seed + 1, start + off))
        t = torch.from_numpy(v.astype(np.float32))
        if dtype == torch.float16:
            t = t.clamp(-65504.0, 65504.0)
        out[off:off + m] = t.to(dtype)

    workers = workers or max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(one, range(0, n, chunk)))
    return out.reshape(shape)


def make(kind, shape, seed=0, dtype=torch.float32, block=16):
    """kinds: normal | heavy (exponent spread) | outlier (one 2^10 outlier per block) | ties (half-quantum
    lattice: every value is a multiple of 2^-8 with |x| < 2) | zeros | denormal | mixed (all of them, by rows) |
    mixed_nd (mixed without the denormal rows)"""
    n = int(np.prod(shape))
    if kind == "normal":
        v = normal(n, seed)
    elif kind == "heavy":
        v = normal(n, seed) * np.exp(4.0 * normal(n, seed + 1))
    elif kind == "outlier":
        v = normal(n, seed)
        pos = np.arange(0, n, block) + (splitmix64(np.arange(0, n, block, dtype=np.uint64), seed + 7) % np.uint64(block)).astype(np.int64)
        pos = pos[pos < n]
        v[pos] *= 1024.0
    elif kind == "ties":
        k = (splitmix64(np.arange(n, dtype=np.uint64), seed) % np.uint64(1023)).astype(np.int64) - 511
        v = k.astype(np.float64) / 256.0
    elif kind == "zeros":
        v = np.zeros(n)
    elif kind == "denormal":
        v = normal(n, seed) * 1e-40
    elif kind in ("mixed", "mixed_nd"):  # mixed_nd: no denormal-maximum blocks
        kinds = ("normal", "heavy", "outlier", "ties", "zeros") + (("denormal",) if kind == "mixed" else ())
        parts = [make(k, (n,), seed + i, torch.float32, block).double().numpy() for i, k in enumerate(kinds)]
        sel = (np.arange(n) // max(block * 4, 1)) % len(parts)
        v = np.choose(sel, parts)
    else:
        raise ValueError(kind)
    t = torch.from_numpy(v.astype(np.float32)).reshape(shape)
    if dtype == torch.float16:  # keep inputs finite: Inf/NaN blocks are covered by their own test
        t = t.clamp(-65504.0, 65504.0)
    return t.to(dtype)


def sha256_bits(t: torch.Tensor) -> str:
    """SHA-256 of a tensor's raw bytes (contiguous, host order): the digest form of tests/golden/*.json"""
    import hashlib

    t = t.detach().cpu().contiguous()
    it = {4: torch.int32, 2: torch.int16, 8: torch.int64, 1: torch.uint8}[t.element_size()]
    return hashlib.sha256(t.view(it).numpy().tobytes()).hexdigest()


def mismatches_nan_aware(a: torch.Tensor, b: torch.Tensor) -> int:
    """like bits_equal, but any NaN matches any NaN (payload/sign of a generated NaN is platform-defined)"""
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    both_nan = torch.isnan(a) & torch.isnan(b)
    diff = (a.view(torch.int32) != b.view(torch.int32)) & ~both_nan
    return int(diff.sum())


def bits_equal(a: torch.Tensor, b: torch.Tensor) -> int:
    """number of elements whose bit patterns differ (NaN-safe, distinguishes -0.0 from +0.0)"""
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    it = {4: torch.int32, 2: torch.int16, 8: torch.int64}[a.element_size()]
    return int((a.view(it) != b.view(it)).sum())


_EPS = {torch.float32: 2.0 ** -23, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}
_TINY = {torch.float32: 2.0 ** -126, torch.bfloat16: 2.0 ** -126, torch.float16: 2.0 ** -14}


def round_once(y64: torch.Tensor, dtype) -> torch.Tensor:
    """float64 -> `dtype` with ONE rounding.  torch converts double -> half / bfloat16 THROUGH float32 (two roundings: 2 of the 65,536
    fp16 gelu values come out one ulp off); here the float32 step rounds to odd (truncate, set the last bit when inexact), after which
    the round-to-nearest-even to 16 bits equals a single rounding of the double (float32 targets: torch's own single rounding)."""
    if dtype == torch.float32 or dtype == torch.float64:
        return y64.to(dtype)
    f = y64.float()
    b = f.view(torch.int32).clone()
    inexact = (f.double() != y64) & torch.isfinite(f) & ~torch.isnan(y64)
    away = f.double().abs() > y64.abs()
    b = torch.where(inexact & away, b - 1, b)
    b = torch.where(inexact, b | 1, b)
    return b.view(torch.float32).to(dtype)


def err_in_ulps(got: torch.Tensor, truth64: torch.Tensor, dtype, floor=None) -> float:
    """Largest |got - truth| in units of the last place OF THE OUTPUT FORMAT `dtype`, against `truth64` (the function
    evaluated in float64 on the same inputs; it is rounded to `dtype` here, once: the correctly rounded result).
    ulp(v) = 2^floor(log2 |v|) * eps(dtype), never smaller than the format's smallest normal ulp.  `floor`: magnitude of
    the terms that CANCEL in the formula (|x|/2 in gelu's 0.5 x + 0.5 x erf(.), (|x| + |mean|) rstd |w| + |b| in
    layer_norm's centring): where the result is much smaller than those terms, no fp32 evaluation of the formula --
    torch's own included -- is accurate relative to the RESULT, and the error is counted in ulps of the cancelling terms
    (what a backward-stable evaluation guarantees).  NaN must meet NaN, +-inf must meet the same inf."""
    g, t = got.detach().cpu().double(), round_once(truth64.detach().cpu().double(), dtype).double()
    nan_g, nan_t = torch.isnan(g), torch.isnan(t)
    if bool((nan_g ^ nan_t).any()):
        return float("inf")
    inf_t = torch.isinf(t)
    if bool((inf_t & (g != t)).any()):
        return float("inf")
    ok = ~(nan_t | inf_t)
    mag = t.abs()
    if floor is not None:
        mag = torch.maximum(mag, floor.detach().cpu().double().abs())
    mag = mag.clamp_min(_TINY[dtype])
    ulp = torch.exp2(torch.floor(torch.log2(mag))) * _EPS[dtype]
    e = ((g - t).abs() / ulp)[ok]
    return float(e.max()) if e.numel() else 0.0


def ulp_of(t64: torch.Tensor, dtype, floor=None) -> torch.Tensor:
    """unit in the last place of `dtype` at |t64| (never below the smallest normal's; 0 where t64 is not finite)"""
    mag = t64.abs()
    if floor is not None:
        mag = torch.maximum(mag, floor.detach().cpu().double().abs())
    fin = torch.isfinite(mag)
    mag = torch.where(fin, mag, torch.ones_like(mag)).clamp_min(_TINY[dtype])
    return torch.where(fin, torch.exp2(torch.floor(torch.log2(mag))) * _EPS[dtype], torch.zeros_like(mag))


def outside_cast_bracket(got: torch.Tensor, truth64: torch.Tensor, cast_out, dtype, n_ulp: float = 1.0, floor=None) -> int:
    """Number of elements that violate the contract of the fused cast modules (include/dmxq.h dmxq_unary_cast ...):
        got == cast_out(v)  for some v within n_ulp ulps (of the tensor dtype) of the float64 truth.
    cast_out (a dtype -> same-dtype CPU function: the ORACLE's bit-exact cast followed by `.to(dtype)`) is monotone, so this is
    checked as  cast_out(truth - n ulp) <= got <= cast_out(truth + n ulp)  -- exact at saturation and flush thresholds, where
    a last-place difference before the cast legitimately decides between two far-apart results.  NaN truth: got must be what
    cast_out makes of a NaN (NaN, or +-max for formats without NaN codes), compared by magnitude."""
    g = got.detach().cpu().double()
    t = truth64.detach().cpu().double()
    u = ulp_of(t, dtype, floor) * n_ulp
    nan = torch.isnan(t)
    tz = torch.where(nan, torch.zeros_like(t), t)
    lo = cast_out((tz - u).to(dtype)).double()
    hi = cast_out((tz + u).to(dtype)).double()
    bad = ~((g >= lo) & (g <= hi)) & ~nan
    if bool(nan.any()):
        want = cast_out(t.to(dtype)).double()
        both_nan = torch.isnan(want) & torch.isnan(g)
        bad = bad | (nan & ~both_nan & ~(g.abs() == want.abs()))
    return int(bad.sum())
