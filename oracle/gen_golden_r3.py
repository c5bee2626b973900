#!/usr/bin/env python3
"""oracle/gen_golden_r3.py — BUILD-CONTAINER ONLY (round 3; the earlier generators are unchanged so that their fixtures stay
byte-identical).  Runs the real reference (/root/reference through oracle/ref_shim.py) on the two parity boundaries that round 2
left to prose and emits tests/golden/boundaries.npz:

  mx_*   MXFP casts (numerical/format.py:545-564) of float32 blocks whose maximum lies 1 .. 96 float32 ulps BELOW a power of
         two, 2^v (1 - j 2^-24): there `torch.floor(torch.log2(max))` comes out as v (one too high) for the smallest j, because
         the float32 log2 rounds to the integer -- which j depends on v.  The oracle and the kernel use a closed rule
         (oracle/oracle.c oracle_floor_log2f) instead of a libm; this script checks the rule against torch.log2 ITSELF for every
         float32 exponent and every j <= 256 plus 2 M random maxima (assertions, nothing stored), then stores the reference's
         casts of such blocks for three MXFP formats.
  asym_* asymmetric BFP casts (`BFP[p|8]{16}(_N)`, numerical/format.py:304-372) of tensors with PLANTED Inf / NaN / denormal-maximum
         blocks.  make_mantissa_asymmetric rebuilds a whole [rows, B] chunk from integers iff ANY row of the chunk holds an edge
         code: a block whose maximum is Inf / NaN then turns into `ldexp(int(NaN), ...)` garbage instead of staying NaN, and a
         -0.0 result (possible only in a block whose maximum is denormal) becomes +0.0 -- effects that depend on OTHER rows.
         This library keeps blocks independent; the fixture pins the reference's output so that the tests can
         assert bit-equality everywhere else and state the differing set exactly (tests/test_golden.py::test_boundaries_*).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_r3.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
import ref_shim  # noqa: E402
from _data import make, splitmix64  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ref = ref_shim.load_reference()
from dmx.compressor import numerical as rnum  # noqa: E402

NP_BITS = {torch.float32: np.uint32, torch.bfloat16: np.uint16, torch.float16: np.uint16}
T_BITS = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.float16: torch.int16}


def bits(t):
    t = t.detach().contiguous()
    return t.view(T_BITS[t.dtype]).numpy().view(NP_BITS[t.dtype]).copy()


def nan_eq(a, b):
    return (bits(a) == bits(b)) | (torch.isnan(a.float()) & torch.isnan(b.float())).numpy()


# ------------------------------------------------------------------------------------------------ floor(log2) rule vs torch
def check_floor_log2_rule():
    lib = O.lib()
    import ctypes
    lib.oracle_floor_log2f.restype = ctypes.c_int
    lib.oracle_floor_log2f.argtypes = [ctypes.c_float]
    pats = []
    for eb in range(1, 255):                              # every normal exponent: the 256 patterns below the next power of two
        top = np.uint32((eb + 1) << 23)
        pats.append(top - np.arange(1, 257, dtype=np.uint32))
        pats.append(np.uint32(eb << 23) + np.arange(0, 4, dtype=np.uint32))   # ... and the first ones of the binade
    rng = np.random.default_rng(0)
    pats.append(rng.integers(0x00800000, 0x7F800000, size=2_000_000, dtype=np.uint32))
    b = np.concatenate(pats).astype(np.uint32)
    m = torch.from_numpy(b.view(np.float32).copy())
    want = torch.floor(torch.log2(m)).numpy().astype(np.int64)          # the reference's expression, numerical/format.py:552
    got = np.fromiter((lib.oracle_floor_log2f(float(v)) for v in m.numpy()), dtype=np.int64, count=len(b))
    bad = int((want != got).sum())
    assert bad == 0, f"floor(log2) rule != torch.log2 on {bad} of {len(b)} maxima"
    ups = int((want != ((b >> 23).astype(np.int64) - 127)).sum())
    print(f"[r3] floor(log2) rule == torch.floor(torch.log2(.)) on {len(b)} float32 maxima ({ups} of them round up to the next integer)")


MX_SH = ["MXFP8[E4M3]{8}", "MXFP6[E2M3]{8}", "MXFP4[E2M1]{8}"]
MX_J = [1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 16, 21, 22, 23, 24, 32, 43, 44, 45, 46, 64, 87, 88, 89, 90, 96]
MX_V = list(range(-20, 21)) + [-125, -124, -100, -65, -64, -63, -33, -32, -31, 31, 32, 33, 63, 64, 65, 100, 126, 127]


def mx_inputs():
    """[len(V) * len(J), 8] float32: row (v, j) is ONE block whose maximum is 2^v (1 - j 2^-24); the other elements are the maximum
    times fractions in (-1, 1) from the counter generator, the maximum's position and sign vary."""
    rows = []
    n = 0
    for v in MX_V:
        for j in MX_J:
            mx = np.float32(np.exp2(float(v)) * (1.0 - j * 2.0 ** -24))
            assert float(mx) < 2.0 ** v and np.float32(2.0 ** v).view(np.uint32) - mx.view(np.uint32) == j
            fr = (splitmix64(np.arange(8, dtype=np.uint64), 1000 + n) >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 1.98 - 0.99
            blk = (fr * float(mx)).astype(np.float32)
            blk[n % 8] = mx if (n // 8) % 2 == 0 else -mx
            rows.append(blk)
            n += 1
    return torch.from_numpy(np.stack(rows))


def mx_cases(store):
    x = mx_inputs()
    store["mx_x"] = bits(x)
    store["mx_sh"] = np.array(MX_SH)
    for i, sh in enumerate(MX_SH):
        f = rnum.Format.from_shorthand(sh)
        ef = f.element_format
        y = rnum.CastTo(format=sh)(x)
        o = O.mxfp_cast(x, ef.mantissa, ef.exponent, f.block_size)
        eq = nan_eq(y, o)
        assert bool(eq.all()), f"ORACLE != REFERENCE: {sh}: {int((~eq).sum())} elements"
        store[f"mx_y{i}"] = bits(y)
        # how many of these blocks the naive exponent-field scale would get wrong (the point of the fixture)
        naive = torch.empty_like(x)
        for r in range(x.shape[0]):
            mxv = x[r].abs().max()
            sc = torch.exp2(torch.floor(torch.tensor(float(np.floor(np.log2(float(mxv))))))) / f.element_format.largest_representable_power_of_two
            naive[r] = f.element_format.cast(x[r:r + 1] / sc)[0] * sc
        print(f"[r3] {sh}: oracle == reference on {x.shape[0]} near-power-of-two blocks; an exact floor(log2) would differ on "
              f"{int((~nan_eq(y, naive)).any(1).sum())} of them")


# ------------------------------------------------------------------------------------------------ asymmetric BFP boundaries
def asym_inputs(dtype):
    x = make("heavy", (48, 64), seed=77, dtype=torch.float32, block=16)
    # edge codes in (almost) every chunk: a value just above -2 quanta-steps of the block maximum (-1.99 x 2^e rounds to -127 / -128)
    for r in range(0, 48, 5):
        for b in range(4):
            blk = x[r, 16 * b: 16 * b + 16]
            e = torch.floor(torch.log2(blk.abs().max()))
            blk[(r + b) % 16] = -1.992 * 2.0 ** e
    x[3, 5] = float("inf")            # poisoned blocks: row 3 block 0, row 7 block 1, row 11 block 2, row 19 block 3
    x[7, 20] = float("nan")
    x[11, 40] = float("-inf")
    x[19, 50] = float("nan")
    # blocks whose maximum is denormal (fp32 denormals survive only in float32 tensors; 16-bit dtypes flush or cannot hold them)
    den = (make("normal", (16,), seed=78).double() * 1e-41).float()
    den[2] = -0.0
    den[5] = -1e-45
    x[13, 16:32] = den
    x[29, 0:16] = -den
    x[37, 48:64] = 0.0                # an all-zero block
    x[41, 32:48] = torch.tensor([-0.0] * 16)
    return x.to(dtype)


ASYM_SH = ["BFP[8|8]{16}(_N)", "BFP[4|8]{16}(_N)", "BFP[6|8]{16}(_N)"]


def asym_cases(store):
    store["asym_sh"] = np.array(ASYM_SH)
    for dt, nm in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        x = asym_inputs(dt)
        store[f"asym_x_{nm}"] = bits(x)
        for i, sh in enumerate(ASYM_SH):
            f = rnum.Format.from_shorthand(sh)
            y = rnum.CastTo(format=sh)(x)
            store[f"asym_y{i}_{nm}"] = bits(y)
            o = O.bfp_cast(x, f.precision, f.block_size, -1, False).to(dt)
            xf = x.float().reshape(48, 4, 16)
            poisoned = ~torch.isfinite(xf).all(-1, keepdim=True).expand_as(xf).reshape(48, 64)
            mx = xf.abs().amax(-1, keepdim=True)
            den_zero = ((mx < 2.0 ** -126).expand_as(xf).reshape(48, 64)) & (o.float() == 0)
            diff = torch.from_numpy(~nan_eq(y, o))
            outside = int((diff & ~poisoned & ~den_zero).sum())
            assert outside == 0, f"{sh} {nm}: oracle != reference on {outside} elements OUTSIDE the documented set"
            print(f"[r3] {sh} {nm}: oracle == reference except {int((diff & poisoned).sum())} elements of Inf/NaN-poisoned blocks and "
                  f"{int((diff & den_zero).sum())} zero signs in denormal-maximum blocks")


if __name__ == "__main__":
    check_floor_log2_rule()
    store = {}
    mx_cases(store)
    asym_cases(store)
    np.savez_compressed(os.path.join(GOLD, "boundaries.npz"), **store)
    print("[r3] wrote tests/golden/boundaries.npz:", sorted(store))
