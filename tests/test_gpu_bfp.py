"""-m gpu parity tests: HIP BFP Q->DQ (through the C ABI) vs the CPU oracle, bit-exact.

Reference behaviour under test: numerical/format.py:304-343 BlockFloatingPoint.cast + quant_cpu.cpp:239-311,
CastTo's dtype contract numerical/cast.py:262,306.  Tolerance: NONE (bit patterns must match, incl. -0.0).
"""
import pytest
import torch

from _data import bits_equal, make, mismatches_nan_aware

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16, torch.float32]


def _run(dmx, cuda, O, x, wl, B, dim=-1, sym=True, rounding="nearest", out_dtype=None, seed=0):
    got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, dim, sym, rounding, out_dtype=out_dtype, seed=seed)
    want = O.bfp_cast(x, wl, B, dim, sym, rounding, seed).to(out_dtype or x.dtype)
    assert got.shape == x.shape and got.dtype == (out_dtype or x.dtype) and got.is_contiguous()
    return bits_equal(got, want.contiguous())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("wl", [4, 6, 8, 16])
@pytest.mark.parametrize("B", [16, 32, 64, 128])
@pytest.mark.parametrize("sym", [True, False])
def test_rows_mixed_inputs(dmx, cuda, oracle, dtype, wl, B, sym):
    x = make("mixed", (64, 512), seed=wl * 131 + B, dtype=dtype, block=B)
    assert _run(dmx, cuda, oracle, x, wl, B, sym=sym) == 0


@pytest.mark.parametrize("kind", ["normal", "heavy", "outlier", "ties", "zeros", "denormal"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_headline_format_each_input_kind(dmx, cuda, oracle, kind, dtype):
    # BFP[8|8]{16}(SN) = "BFP16 group 16", the north-star format
    x = make(kind, (128, 1024), seed=3, dtype=dtype, block=16)
    assert _run(dmx, cuda, oracle, x, 8, 16) == 0


@pytest.mark.parametrize("din,dout", [(torch.bfloat16, torch.float32), (torch.float16, torch.float32),
                                      (torch.float32, torch.bfloat16), (torch.float32, torch.float16)])
def test_mixed_io_dtypes(dmx, cuda, oracle, din, dout):
    x = make("heavy", (32, 256), seed=9, dtype=din)
    assert _run(dmx, cuda, oracle, x, 8, 64, out_dtype=dout) == 0


@pytest.mark.parametrize("B", [1, 2, 3, 4, 8, 24, 40, 256, 512])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_odd_block_sizes_and_ragged_tails(dmx, cuda, oracle, B, dtype):
    # torch.split semantics: last block shorter (format.py:324-326); LeNet fc1 in=400, 120, 84
    for shape in [(4, 40), (7, 400), (3, 120), (5, 84), (2, 1000)]:
        x = make("heavy", shape, seed=B, dtype=dtype)
        assert _run(dmx, cuda, oracle, x, 8, B) == 0, (shape, B)


@pytest.mark.parametrize("dim", [1, -2, 0, 2])
@pytest.mark.parametrize("B", [16, 64, 5])
def test_block_dim_layouts(dmx, cuda, oracle, dim, B):
    # conv activations / weights block along dim 1, attention multipliers along -2
    # (the last three: rows of whole 128-byte lines -- B = 64 then runs 8 rows per lane in 8 row groups, csrc/bfp_cols.hip launch_cols --
    #  with ragged last blocks: L = 100, 70)
    for shape in [(2, 32, 5, 5), (3, 6, 5, 5), (2, 1, 32, 32), (4, 64, 48), (3, 100, 64), (2, 2, 128, 128), (2, 70, 192)]:
        if dim >= len(shape):
            continue
        x = make("normal", shape, seed=dim + 10, dtype=torch.bfloat16)
        assert _run(dmx, cuda, oracle, x, 8, B, dim=dim) == 0, (shape, dim, B)
        assert _run(dmx, cuda, oracle, x.float(), 6, B, dim=dim, sym=False) == 0, (shape, dim, B)


@pytest.mark.parametrize("B", [8, 16, 64, 256])
@pytest.mark.parametrize("dtype", DTYPES)
def test_ragged_and_unaligned_rows(dmx, cuda, oracle, B, dtype):
    """Row lengths that are not whole blocks and/or not 16-byte aligned: the direct unaligned-access kernel
    (bfp_urows.hip; same-size dtypes) and the LDS-staged one (bfp_ragged.hip; widening / narrowing casts, every copy
    width), every tail length of the last lane-vector, rows longer than one LDS segment, a view that starts
    mid-buffer."""
    for L in range(4, 24):
        x = make("mixed", (5, L), seed=B + L, dtype=dtype, block=4)
        assert _run(dmx, cuda, oracle, x, 8, B) == 0, L
    for shape in ((7, 400), (33, 1500), (5, 84), (3, 1001), (2, 9001), (1, 13), (64, 120), (3, 5, 2, 50)):
        x = make("mixed", shape, seed=B + shape[-1], dtype=dtype, block=8)
        assert _run(dmx, cuda, oracle, x, 8, B) == 0, shape
        xa = make("mixed_nd", shape, seed=B, dtype=dtype, block=8)
        assert _run(dmx, cuda, oracle, xa, 4, B, sym=False) == 0, shape
    x = make("heavy", (9, 1500), seed=3, dtype=dtype)
    for rounding in ("down", "up", "stochastic"):
        assert _run(dmx, cuda, oracle, x, 8, B, rounding=rounding, seed=5) == 0, rounding
    assert _run(dmx, cuda, oracle, x, 22, B) == 0
    # misaligned base pointer: a contiguous slice that starts 2 bytes (16-bit) / 4 bytes (fp32) into the allocation
    flat = make("normal", (1 + 6 * 200,), seed=9, dtype=dtype).to(cuda)
    xv = flat[1:].view(6, 200)
    got = dmx.ops.bfp_qdq(xv, 8, B)
    want = oracle.bfp_cast(xv.cpu(), 8, B).to(dtype)
    assert bits_equal(got, want) == 0
    if dtype != torch.float32:
        assert _run(dmx, cuda, oracle, x, 8, B, out_dtype=torch.float32) == 0
    else:
        assert _run(dmx, cuda, oracle, x, 8, B, out_dtype=torch.bfloat16) == 0


@pytest.mark.parametrize("B", [8, 16, 32, 64, 128])
@pytest.mark.parametrize("dtype", DTYPES)
def test_column_block_kernel(dmx, cuda, oracle, B, dtype):
    """Blocks along a non-contiguous dim with inner % (16 B) == 0: the register-tiled kernel of bfp_cols.hip
    (row split across half / quarter waves for B = 64 / 128), incl. ragged last block and column tails."""
    for shape, dim in (((3, 128, 64), 1), ((2, 200, 520), -2), ((1, 64, 8), 1), ((260, 1032), 0), ((2, 3, 96, 40), 2)):
        x = make("mixed", shape, seed=B + len(shape), dtype=dtype, block=8)
        assert _run(dmx, cuda, oracle, x, 8, B, dim=dim) == 0, (shape, dim)
        xa = make("mixed_nd", shape, seed=B, dtype=dtype, block=8)
        assert _run(dmx, cuda, oracle, xa, 6, B, dim=dim, sym=False) == 0, (shape, dim)
    x = make("heavy", (2, 128, 64), seed=1, dtype=dtype)
    for rounding in ("down", "up", "stochastic"):
        assert _run(dmx, cuda, oracle, x, 8, B, dim=1, rounding=rounding, seed=77) == 0, rounding
    assert _run(dmx, cuda, oracle, x, 22, B, dim=1) == 0                            # wl > 20: literal path
    if dtype != torch.float32:
        assert _run(dmx, cuda, oracle, x, 8, B, dim=1, out_dtype=torch.float32) == 0
        assert _run(dmx, cuda, oracle, x, 16, B, dim=1) == 0                         # double-rounding form on 16-bit input


@pytest.mark.parametrize("rounding", ["down", "up", "stochastic"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_other_rounding_modes(dmx, cuda, oracle, rounding, dtype):
    x = make("heavy", (64, 256), seed=5, dtype=dtype)
    assert _run(dmx, cuda, oracle, x, 8, 16, rounding=rounding, seed=1234) == 0
    assert _run(dmx, cuda, oracle, x, 6, 24, rounding=rounding, seed=99) == 0          # generic kernel
    assert _run(dmx, cuda, oracle, x.reshape(8, 8, 256), 8, 4, dim=1, rounding=rounding, seed=7) == 0


def test_fp32_double_rounding_class(dmx, cuda, oracle):
    """fp32 inputs within 2^-22 relative of a rounding tie: the reference rounds twice (x + 6*2^e, then the
    mantissa); a single-rounding implementation differs on these (SURVEY.md §7 'Hard parts')."""
    g = torch.Generator().manual_seed(0)
    B = 16
    base = torch.randint(-127, 127, (4096, B), generator=g).float() + 0.5      # exact ties at quantum 1
    eps = (torch.randint(-3, 4, (4096, B), generator=g).float()) * 2.0 ** -17   # nudges below fp32 ulp of 6*64
    x = base + eps
    x[:, 0] = 100.0  # block max exponent e = 6 -> quantum 2^0
    assert _run(dmx, cuda, oracle, x, 8, B) == 0


def _all_patterns(dtype):
    """every finite 16-bit pattern of the dtype, as a tensor"""
    bits = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16)
    v = bits.view(dtype)
    return v[torch.isfinite(v.float())]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("wl", [2, 4, 6, 8, 11, 12, 14, 15, 16, 20, 21, 22])
def test_exhaustive_16bit_inputs_against_every_block_exponent_class(dmx, cuda, oracle, dtype, wl):
    """The nearest-even kernel replaces the reference's bit-level mantissa rounding by fp32 magic-constant adds
    (bfp_math.hpp (2)), collapsing the two roundings into one where the input dtype allows it.  Check EVERY
    finite 16-bit input value against block maxima of every kind: ordinary, denormal, tiny-normal, near the top
    of the range (where the kernel must fall back to the literal bit path) and a block maximum equal to x itself."""
    vals = _all_patterns(dtype)
    fin = vals.float().abs()
    tops = [1.0, 1.99, 3.5e-3, 2.0 ** -126, 1.2e-38, 3e-39, 1e-41, 2.0 ** 100, 2.0 ** 110, 2.0 ** 112, 3e38] if dtype == torch.bfloat16 \
        else [1.0, 1.99, 6.1e-5, 5.96e-8, 3e-6, 1000.0, 65504.0, 0.33]
    B = 16
    for top in tops:
        t = torch.tensor(top).to(dtype)
        keep = vals[fin <= float(t.float())]
        n = (keep.numel() + B - 2) // (B - 1)
        blk = torch.zeros(n, B, dtype=dtype)
        blk[:, 0] = t
        flat = torch.zeros(n * (B - 1), dtype=dtype)
        flat[: keep.numel()] = keep
        blk[:, 1:] = flat.reshape(n, B - 1)
        for sym in (True, False):
            got = dmx.ops.bfp_qdq(blk.to(cuda), wl, B, -1, sym)
            want = oracle.bfp_cast(blk, wl, B, -1, sym).to(dtype)
            assert mismatches_nan_aware(got, want) == 0, (top, sym)
    # self-maximum blocks: each value alone with zeros
    solo = torch.zeros(vals.numel(), B, dtype=dtype)
    solo[:, 3] = vals
    assert mismatches_nan_aware(dmx.ops.bfp_qdq(solo.to(cuda), wl, B), oracle.bfp_cast(solo, wl, B).to(dtype)) == 0


@pytest.mark.parametrize("wl", [4, 8, 12, 16, 20, 22])
def test_fp32_inputs_near_every_tie(dmx, cuda, oracle, wl):
    """fp32 inputs: the double-rounding form.  Values at, just below and just above every rounding tie of the
    second rounding, at several magnitudes of the block max, plus random mantissas."""
    g = torch.Generator().manual_seed(wl)
    B = 32
    rows = []
    for e in (-126, -120, -60, -1, 0, 7, 60, 100, 120):
        quantum = 2.0 ** (e + 2 - wl)
        k = torch.randint(-(2 ** (wl - 1)) + 1, 2 ** (wl - 1) - 1, (512, B), generator=g).double()
        nud = torch.randint(-2, 3, (512, B), generator=g).double() * 2.0 ** (e - 22)   # around 1 ulp of x + 6*2^e
        x = ((k + 0.5) * quantum + nud)
        x[:, 0] = 1.5 * 2.0 ** e      # block max with exponent e
        x = x.clamp(-1.99 * 2.0 ** e, 1.99 * 2.0 ** e)
        rows.append(x.float())
        r = (torch.rand(256, B, generator=g).double() * 4 - 2) * 2.0 ** e
        r[:, 0] = 1.999 * 2.0 ** e
        rows.append(r.float())
    x = torch.cat(rows)
    for sym in (True, False):
        got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, -1, sym)
        assert mismatches_nan_aware(got, oracle.bfp_cast(x, wl, B, -1, sym)) == 0, sym


def test_full_size_headline_config(dmx, cuda, oracle):
    """BASELINE.json config 2: 4096x4096 bf16, BFP[8|8]{16}(SN)."""
    x = make("heavy", (4096, 4096), seed=0, dtype=torch.bfloat16)
    assert _run(dmx, cuda, oracle, x, 8, 16) == 0


def _c2_gold():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_digests.json")))


@pytest.mark.parametrize("kind", ["normal", "heavy", "outlier", "ties"])
def test_c2_reference_digests_by_input_kind(dmx, cuda, oracle, kind):
    """BASELINE.json config 2 against the REFERENCE itself: the SHA-256 of what the reference's CastTo("BFP[8|8]{16}(SN)") returned in
    the build container for the four SURVEY §8(d) input kinds of the counter-based generator (oracle/gen_golden_r4.py ->
    tests/golden/c2_digests.json).  The oracle comparison stands beside it, so a host whose libm rounds the generator's fp64
    log / cos differently (input digest mismatch) still tests the kernel."""
    from _data import make_chunked, sha256_bits

    g = _c2_gold()["kinds"][kind]
    x = make_chunked(kind, (4096, 4096), 0, torch.bfloat16) if kind in ("normal", "heavy") else make(kind, (4096, 4096), seed=0, dtype=torch.bfloat16)
    got = dmx.CastTo(format="BFP[8|8]{16}(SN)")(x.to(cuda))
    assert got.dtype == torch.bfloat16
    assert bits_equal(got, oracle.bfp_cast(x, 8, 16).to(torch.bfloat16)) == 0
    assert sha256_bits(x) == g["input_sha256"], "the generator produced different bits on this host"
    assert sha256_bits(got) == g["output_sha256"]


@pytest.mark.parametrize("seed", [1, 7, 19, 1000, 1013])
def test_c2_reference_digests_of_bench_slots(dmx, cuda, seed):
    """the inputs bench.py rotates over (rank 0 slots 1 / 7 / 19, rank 1 slots 0 / 13) through the C-ABI front end"""
    from _data import make_chunked, sha256_bits

    g = _c2_gold()["slots"][str(seed)]
    x = make_chunked("heavy", (4096, 4096), seed, torch.bfloat16)
    assert sha256_bits(x) == g["input_sha256"]
    assert sha256_bits(dmx.ops.bfp_qdq(x.to(cuda), 8, 16)) == g["output_sha256"]


def test_properties_at_full_size(dmx, cuda):
    """Size-independent properties on the 4096x4096 bf16 tensor (no oracle involved)."""
    x = make("heavy", (4096, 4096), seed=1, dtype=torch.bfloat16).to(cuda)
    q = dmx.ops.bfp_qdq(x, 8, 16)
    assert bits_equal(dmx.ops.bfp_qdq(q, 8, 16), q) == 0                      # idempotent
    qn = dmx.ops.bfp_qdq(-x, 8, 16)
    assert torch.equal(qn, -q)                                               # odd symmetry (symmetric format)
    # shard invariance: rows are independent -> quantising 8 row shards == quantising the whole (§8e)
    shards = torch.cat([dmx.ops.bfp_qdq(s, 8, 16) for s in x.chunk(8, dim=0)])
    assert bits_equal(shards, q) == 0
    # every block has at most 2^8-1 distinct codes on a common quantum: max/quantum <= 127
    qb = q.float().reshape(-1, 16)
    m = qb.abs().amax(dim=1, keepdim=True)
    quantum = torch.where(m > 0, torch.exp2(torch.floor(torch.log2(m)) - 6), torch.ones_like(m))
    codes = qb / quantum
    assert torch.all(codes == codes.round()) and float(codes.abs().max()) <= 127


@pytest.mark.parametrize("dtype", DTYPES)
def test_inf_nan_poison_the_block_like_the_reference(dmx, cuda, oracle, dtype):
    """Unspecified by the reference's docs but observable: a block whose max|x| is Inf/NaN (torch.max propagates
    NaN, quant_cpu.cpp:277-297) gets base = inf and every element of THAT block becomes NaN; others are untouched."""
    x = make("normal", (16, 256), seed=12, dtype=dtype)
    x[1, 5] = float("inf")
    x[2, 40] = float("-inf")
    x[3, 100] = float("nan")
    x[4, 255] = float("nan"); x[4, 254] = float("inf")
    for B in (16, 64, 24):
        got = dmx.ops.bfp_qdq(x.to(cuda), 8, B)
        want = oracle.bfp_cast(x, 8, B).to(dtype)
        assert mismatches_nan_aware(got, want) == 0, B
        assert int(torch.isnan(got).sum()) == (4 * B if B != 24 else int(torch.isnan(want).sum()))


def test_nan_payloads_in_nan_and_inf_blocks(dmx, cuda, oracle):
    """The reference rounds the mantissa bits of a NaN element like any other value: what it becomes depends on its PAYLOAD (and on
    wl).  Every row geometry (tile plans by size, ragged rows, column blocks), NaNs of several payloads and signs next to Inf and
    finite values, against the oracle (bit for bit, any NaN == any NaN: the payload of a NaN RESULT is the host FPU's)."""
    payloads = [0x7FC00000, 0xFFC00000, 0x7F800001, 0x7FFFFFFF, 0xFFFF0000, 0x7FA00000, 0xFF812345, 0x7FC00001]
    for shape, dim in (((1, 128), -1), ((3, 50, 256), -1), ((64, 4096), -1), ((37, 1500), -1), ((128, 96), 0), ((5, 84), -1)):
        x = make("heavy", shape, seed=21).clamp(-1e4, 1e4)
        flat = x.view(-1).view(torch.int32)
        n = flat.numel()
        for i, pl in enumerate(payloads):
            flat[(i * 997 + 3) % n] = pl - (1 << 32) if pl >= (1 << 31) else pl
        x.view(-1)[(5 * 997) % n] = float("inf")
        x.view(-1)[(11 * 997 + 1) % n] = float("-inf")
        for wl in (4, 8, 12):
            for B in (16, 64, 128):
                for sym in (True, False):
                    got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, dim, sym).cpu()
                    want = oracle.bfp_cast(x, wl, B, dim, sym)
                    assert mismatches_nan_aware(got, want) == 0, (shape, dim, wl, B, sym)
    # 16-bit inputs carry 7 / 10 payload bits
    for dt, pats in ((torch.bfloat16, [0x7FC0, 0xFFC0, 0x7F81, 0x7FFF, 0xFFFF, 0xFFA0]), (torch.float16, [0x7E00, 0xFE00, 0x7C01, 0x7FFF, 0xFFFF, 0xFD55])):
        x = make("heavy", (40, 512), seed=22).clamp(-1e4, 1e4).to(dt)
        flat = x.view(-1).view(torch.int16)
        for i, pl in enumerate(pats):
            flat[(i * 991 + 7) % flat.numel()] = pl - (1 << 16) if pl >= (1 << 15) else pl
        for wl in (4, 8):
            for B in (16, 64):
                got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, -1, True, out_dtype=torch.float32).cpu()
                want = oracle.bfp_cast(x, wl, B, -1, True)
                assert mismatches_nan_aware(got, want) == 0, (dt, wl, B)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_blocks_with_a_small_inner_extent(dmx, cuda, oracle, dtype):
    """csrc/bfp_smallinner.hip: conv weights blocked along `in` ([out, in, kh, kw]: inner = 9, 3, 49, 2 ...), 7 x 7 maps blocked
    along channels -- sub-slabs staged through the LDS, one lane per block -- against the oracle, incl. denormal / Inf / NaN /
    all-zero blocks, asymmetric codes, several precisions, and in place."""
    for shape, dim in (((16, 128, 3, 3), 1), ((8, 64, 7, 7), 1), ((4, 256, 3), 1), ((2, 64, 2), 1), ((3, 128, 5, 5), 1), ((300, 64, 9), 1),
                       ((2, 3, 128, 64), -2), ((5, 512, 33), 1), ((2, 3, 1500, 64), -2), ((3, 100, 32), 1), ((2, 92, 16), 1)):
        x = make("heavy", shape, seed=31).clamp(-6e4, 6e4).to(dtype)
        flat = x.view(-1)
        flat[0], flat[7], flat[31] = float("inf"), float("nan"), -float("inf")
        x.select(dim if dim >= 0 else x.dim() + dim, 0).zero_()  # zeros spread over blocks
        tiny = 2.0 ** -130 if dtype == torch.bfloat16 else 2.0 ** -20
        x.view(-1)[-shape[-1] * 8:] *= 0
        x.view(-1)[-3] = tiny
        for wl, B, sym in ((8, 64, True), (8, 16, False), (4, 128, True), (12, 8, True), (8, 32, False)):
            if x.shape[dim] % B and shape[1:] not in ((3, 1500, 64), (100, 32), (92, 16)):
                continue
            want = oracle.bfp_cast(x, wl, B, dim, sym).to(dtype)
            got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, dim, sym)
            assert got.dtype == dtype and mismatches_nan_aware(got.cpu(), want) == 0, (shape, dim, wl, B, sym)
    import ctypes
    from dmx_compressor_amd import _lib
    x = make("normal", (6, 128, 3, 3), seed=32).to(dtype)
    t = x.to(cuda).contiguous()
    vp = ctypes.c_void_p
    assert _lib.lib().dmxq_bfp_qdq(vp(t.data_ptr()), vp(t.data_ptr()), _lib.dtype_code(dtype), _lib.dtype_code(dtype), 6, 128, 9, 64, 8, 2, 1, 0,
                                   vp(torch.cuda.current_stream().cuda_stream)) == 0
    assert bits_equal(t.cpu(), oracle.bfp_cast(x, 8, 64, 1, True).to(dtype)) == 0


def test_inplace_and_noncontiguous(dmx, cuda, oracle):
    x = make("normal", (64, 96), seed=2, dtype=torch.bfloat16)
    xt = x.t()  # non-contiguous view
    got = dmx.ops.bfp_qdq(xt.to(cuda), 8, 16)
    want = oracle.bfp_cast(xt, 8, 16).to(torch.bfloat16)
    assert bits_equal(got, want.contiguous()) == 0


def test_empty_and_scalar(dmx, cuda):
    assert dmx.ops.bfp_qdq(torch.empty(0, 16, device=cuda), 8, 16).shape == (0, 16)
    assert dmx.ops.bfp_qdq(torch.empty(4, 0, device=cuda), 8, 16).shape == (4, 0)
    assert float(dmx.ops.bfp_qdq(torch.tensor(1.2345, device=cuda), 8, 16)) == 1.234375


def test_error_behaviour(dmx, cuda):
    with pytest.raises(dmx.DmxqError):
        dmx.ops.bfp_qdq(torch.randn(4, 16), 8, 16)                          # CPU tensor: loud, no fallback
    with pytest.raises(NotImplementedError):
        dmx.ops.bfp_qdq(torch.randn(4, 16, device=cuda), 24, 16)            # reference UB region
    with pytest.raises(TypeError):
        dmx.ops.bfp_qdq(torch.zeros(4, 16, device=cuda, dtype=torch.int32), 8, 16)


def test_inplace_through_the_c_abi(dmx, cuda, oracle):
    """include/dmxq.h: `in` and `out` may alias exactly.  Every layout path (flat rows, column blocks, ragged/LDS,
    generic) and the elementwise / mask kernels, called with out == in."""
    import ctypes
    from dmx_compressor_amd import _lib
    L = _lib.lib()
    vp = ctypes.c_void_p
    s = vp(torch.cuda.current_stream().cuda_stream)
    # (37, 1500) / (9, 1004): rows that are not a whole number of 16-byte vectors (bfp_urows.hip), whose last vector is read as
    # the 16 bytes ENDING at the row end when out != in -- and must NOT be when out == in (the overlap would re-read stored data)
    for shape, dim, B in (((64, 512), -1, 16), ((4, 64, 48), 1, 16), ((7, 400), -1, 64), ((5, 84), -1, 24), ((37, 1500), -1, 64),
                          ((37, 1500), -1, 16), ((9, 1004), -1, 16), ((300, 1500), -1, 64)):
        x = make("heavy", shape, seed=B, dtype=torch.bfloat16)
        want = oracle.bfp_cast(x, 8, B, dim).to(torch.bfloat16).contiguous()
        t = x.to(cuda).contiguous()
        outer, Ld, inner = _lib.split3(t.shape, dim)
        o = torch.empty_like(t)
        assert L.dmxq_bfp_qdq(vp(t.data_ptr()), vp(o.data_ptr()), _lib.BF16, _lib.BF16, outer, Ld, inner, B, 8, 2, 1, 0, s) == 0
        assert bits_equal(o, want) == 0, ("out of place", shape, dim, B)
        assert L.dmxq_bfp_qdq(vp(t.data_ptr()), vp(t.data_ptr()), _lib.BF16, _lib.BF16, outer, Ld, inner, B, 8, 2, 1, 0, s) == 0
        assert bits_equal(t, want) == 0, (shape, dim, B)
    x = make("heavy", (1000,), seed=1)
    t = x.to(cuda)
    assert L.dmxq_float_qdq(vp(t.data_ptr()), vp(t.data_ptr()), _lib.F32, _lib.F32, 1000, 3, 4, 7, 0, 0, 2, 0, s) == 0
    assert bits_equal(t, oracle.float_quantize(x, 3, 4, 7, False)) == 0
    sc = make("normal", (64, 64), seed=2)
    t = sc.to(cuda)
    assert L.dmxq_nm_mask(vp(t.data_ptr()), _lib.F32, vp(t.data_ptr()), _lib.F32, None, 0, vp(t.data_ptr()), _lib.F32, 64, 64, 1, 2, 4, s) == 0
    assert bits_equal(t, oracle.sparsify(sc, sc, 2, 4)) == 0


def test_more_than_2_to_the_31_elements(dmx, cuda):
    """64-bit sizes (the reference's kernels take `int size`): a 2^31 + 2^20 element bf16 tensor; the rows beyond the
    32-bit boundary must equal the same rows quantised on their own (shard invariance, no oracle needed)."""
    free, _ = torch.cuda.mem_get_info()
    n_rows = (1 << 19) + 256           # x 4096 columns = 2^31 + 2^20 elements, 4 GiB per tensor
    if free < 3 * n_rows * 4096 * 2:
        pytest.skip("not enough free device memory")
    base = make("heavy", (256, 4096), seed=11, dtype=torch.bfloat16).to(cuda)
    x = base.repeat(n_rows // 256, 1)
    assert x.numel() > 2 ** 31
    q = dmx.ops.bfp_qdq(x, 8, 16)
    ref = dmx.ops.bfp_qdq(base, 8, 16)
    assert torch.equal(q[-256:], ref) and torch.equal(q[:256], ref) and torch.equal(q[(1 << 19) - 256:(1 << 19)], ref)
    f = dmx.ops.float_qdq(x, 10, 5, 15, True)
    assert torch.equal(f[-256:], dmx.ops.float_qdq(base, 10, 5, 15, True))
    del x, q, f
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_masked_attention_scores(dmx, cuda, oracle, dtype):
    """-inf above the diagonal (a causal mask already applied): every block that touches the mask becomes NaN in the
    reference (base = inf, inf - inf), the others quantise normally.  These blocks stay on the magic-add path
    (bfp_math.hpp), so this is also its NaN-propagation test: symmetric / asymmetric, wl = 2 (limit = 2^e, the case
    where an un-poisoned clamp would return a finite limit), rows / columns / ragged layouts."""
    S = 192
    x = make("normal", (3, S, S), seed=21, dtype=dtype)
    x = x.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    x[1, 5, 2] = float("nan")
    x[2, 7, 0] = float("inf")
    for wl, B, dim, sym in ((8, 16, -1, True), (8, 64, -1, False), (2, 16, -1, True), (2, 16, -1, False), (16, 32, -1, True),
                            (8, 16, -2, True), (8, 64, 1, False), (8, 24, -1, True), (4, 128, -1, True)):
        want = oracle.bfp_cast(x, wl, B, dim, sym).to(dtype)
        got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, dim, sym)
        assert mismatches_nan_aware(got, want) == 0, (wl, B, dim, sym)
        assert int(torch.isnan(want.float()).sum()) > S * S  # the mask really poisons blocks
    # the same through the fused weight path and a widening cast
    w = x[0].contiguous()
    want = oracle.bfp_cast(w, 8, 64, -1, True)
    assert mismatches_nan_aware(dmx.ops.bfp_qdq(w.to(cuda), 8, 64, -1, True, out_dtype=torch.float32), want.float()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_column_blocks_on_odd_feature_maps(dmx, cuda, oracle, dtype):
    """Blocks along the channel dim of [N, C, H, W] activations whose H*W is not a whole number of 16-byte vectors
    (14x14, 7x7, 5x5: the unaligned form of the column kernel, whose last vector of a row ends AT the row end and
    overlaps its neighbour), every block size of that kernel, ragged channel counts, rounding modes, a widening cast, a
    view that starts mid-allocation, and in-place (which must still be right: it takes the generic kernel)."""
    for n, (shape, B) in enumerate([((2, 64, 14, 14), 64), ((3, 128, 7, 7), 128), ((2, 40, 5, 5), 16), ((1, 72, 3, 3), 8),
                                    ((2, 100, 14, 14), 32), ((4, 64, 9), 64), ((2, 256, 197), 64)]):
        x = make("mixed", shape, seed=60 + n, dtype=dtype, block=8)
        assert _run(dmx, cuda, oracle, x, 8, B, dim=1) == 0, (shape, B)
        assert _run(dmx, cuda, oracle, make("mixed_nd", shape, seed=n, dtype=dtype, block=8), 4, B, dim=1, sym=False) == 0, (shape, B)
    x = make("heavy", (2, 64, 7, 7), seed=3, dtype=dtype)
    for rounding in ("down", "up", "stochastic"):
        assert _run(dmx, cuda, oracle, x, 8, 64, dim=1, rounding=rounding, seed=9) == 0, rounding
    if dtype != torch.float32:
        assert _run(dmx, cuda, oracle, x, 8, 64, dim=1, out_dtype=torch.float32) == 0
    flat = make("normal", (1 + 2 * 64 * 49,), seed=4, dtype=dtype).to(cuda)
    xv = flat[1:].view(2, 64, 7, 7)
    assert bits_equal(dmx.ops.bfp_qdq(xv, 8, 64, 1), oracle.bfp_cast(xv.cpu(), 8, 64, 1).to(dtype).contiguous()) == 0
    # in place through the C ABI
    import ctypes
    from dmx_compressor_amd import _lib
    t = x.to(cuda).contiguous()
    vp = ctypes.c_void_p
    code = _lib.dtype_code(dtype)
    assert _lib.lib().dmxq_bfp_qdq(vp(t.data_ptr()), vp(t.data_ptr()), code, code, 2, 64, 49, 64, 8, 2, 1, 0,
                                   vp(torch.cuda.current_stream().cuda_stream)) == 0
    assert bits_equal(t, oracle.bfp_cast(x, 8, 64, 1).to(dtype).contiguous()) == 0
