"""-m gpu parity tests: low-bit float and fixed-point (+affine / group-quant) Q->DQ vs the CPU oracle, bit-exact.

Reference: quant_cpu.cpp:359-402 (float), :127-209 + sim_helper.cpp (fixed), numerical/cast.py:278-296 (affine).
"""
import pytest
import torch

from _data import bits_equal, make

pytestmark = pytest.mark.gpu

# (man, exp, bias, flush): FP16-FN, FP16 no-flush, BF16-FN, E4M3, E5M2, FP[0|8|0] scaler, BFP32_1 path, E2M1, E3M2
FLOAT_FORMATS = [(10, 5, 15, True), (10, 5, 15, False), (7, 8, 127, True), (3, 4, 7, False), (2, 5, 15, False),
                 (0, 8, 127, False), (22, 8, 127, False), (1, 2, 1, False), (2, 3, 3, False), (4, 4, 7, True)]


def _special(dtype):
    v = torch.tensor([0.0, -0.0, 65504.0, 65520.0, 3e38, -3e38, 1e-40, -1e-40, 6.0e-5, 6.1035e-5, 6.2e-5, 5.9e-8,
                      448.0, 464.0, 480.0, 1e-3, -1.5, 2.0 ** -14, 2.0 ** -15, 2.0 ** -24, 2.0 ** -25])
    return v.to(dtype)


@pytest.mark.parametrize("fmt", FLOAT_FORMATS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_float_qdq_nearest(dmx, cuda, oracle, fmt, dtype):
    man, exp, bias, flush = fmt
    x = torch.cat([make("heavy", (50003,), seed=man + exp, dtype=dtype), _special(dtype)])
    got = dmx.ops.float_qdq(x.to(cuda), man, exp, bias, flush)
    want = oracle.float_quantize(x, man, exp, bias, flush).to(dtype)
    assert bits_equal(got, want) == 0
    got32 = dmx.ops.float_qdq(x.to(cuda), man, exp, bias, flush, out_dtype=torch.float32)
    assert bits_equal(got32, oracle.float_quantize(x, man, exp, bias, flush)) == 0


@pytest.mark.parametrize("rounding", ["stochastic"])
def test_float_qdq_stochastic_matches_oracle_stream(dmx, cuda, oracle, rounding):
    x = make("heavy", (4099,), seed=4)
    got = dmx.ops.float_qdq(x.to(cuda), 3, 4, 7, False, rounding=rounding, seed=42)
    assert bits_equal(got, oracle.float_quantize(x, 3, 4, 7, False, rounding, 42)) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_float_qdq_every_16_bit_pattern(dmx, cuda, oracle, dtype):
    """EXHAUSTIVE: all 65536 bit patterns of the tensor dtype (every NaN payload, both infinities, every subnormal) through the
    formats that take the packed range-only kernel (common.hpp range16_of) and through ones that do not, against the oracle on the
    widened value followed by CastTo's `.to(dtype)`; any NaN == any NaN."""
    from _data import mismatches_nan_aware
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(dtype).repeat(2)     # 131072 elements: whole 16-byte vectors, two tiles
    for man, exp, bias, flush in ((10, 5, 15, True), (7, 8, 127, True), (10, 4, 7, True), (12, 5, 15, True), (22, 8, 127, True), (10, 5, 15, False),
                                  (3, 4, 7, True), (7, 5, 15, True), (10, 8, 127, True), (10, 3, 3, True), (11, 2, 1, True)):
        want = oracle.float_quantize(x.float(), man, exp, bias, flush).to(dtype)
        got = dmx.ops.float_qdq(x.to(cuda), man, exp, bias, flush)
        assert got.dtype == dtype and mismatches_nan_aware(got.cpu(), want) == 0, (dtype, man, exp, bias, flush)


def test_float_unsigned_and_bypass(dmx, cuda, oracle):
    x = make("normal", (1000,), seed=1)
    f = dmx.Format.from_shorthand("FP[0|4|4,7](FN)")
    got = f.cast(x.to(cuda))
    want = oracle.floating_point_cast(x, 4, 4, 7, True, unsigned=True)
    assert bits_equal(got, want) == 0
    xg = x.to(cuda)
    assert dmx.format.FLOAT32.cast(xg) is xg                                   # format.py:209-212 pass-through
    h = xg.half()
    assert dmx.Format.from_shorthand("FP[1|5|10,15](_N)").cast(h) is h
    with pytest.raises(NotImplementedError):
        dmx.ops.float_qdq(xg, 23, 8, 127, False)                               # reference UB (shift by -1)


def test_fp16_format_matches_torch_half_roundtrip(dmx, cuda):
    """FLOAT16 = FP[1|5|10,15](FN) on in-range values is IEEE fp16 rounding with subnormals flushed."""
    x = make("normal", (100000,), seed=8).to(cuda)
    got = dmx.format.FLOAT16.cast(x)
    ref = x.half().float()
    ref = torch.where(ref.abs() < 2.0 ** -14, torch.zeros_like(ref), ref)
    assert torch.equal(got, ref)


FIXED_FORMATS = [(8, 0, True, True), (8, 0, True, False), (4, 0, True, True), (8, 4, True, True), (8, -2, True, False),
                 (16, 8, False, True), (4, 2, True, False), (24, 0, True, True)]


@pytest.mark.parametrize("fmt", FIXED_FORMATS)
@pytest.mark.parametrize("rounding", ["nearest", "down", "up", "stochastic"])
def test_fixed_qdq(dmx, cuda, oracle, fmt, rounding):
    wl, fl, clamp, sym = fmt
    edge = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 0.5 + 2.0 ** -24, 126.5, 127.5, -127.5, 128.5, 1e9, -1e9, 0.0, -0.0])
    x = torch.cat([make("heavy", (50001,), seed=wl + fl), make("normal", (50000,), seed=3) * 40, edge])
    got = dmx.ops.fixed_qdq(x.to(cuda), wl, fl, clamp, sym, rounding, seed=11)
    want = oracle.fixed_point_cast(x, wl, fl, clamp, sym, rounding, seed=11)
    assert bits_equal(got, want) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_fixed_affine_per_tensor_channel_group(dmx, cuda, oracle, dtype):
    x = make("normal", (37, 50, 3), seed=21, dtype=dtype)
    g = torch.Generator().manual_seed(1)
    # per tensor
    sc, zp = torch.tensor([0.0173]), torch.tensor([3])
    got = dmx.ops.fixed_qdq(x.to(cuda), 8, 0, True, False, scale=sc, zero_point=zp)
    want = oracle.fixed_point_affine_cast(x, 8, 0, True, False, sc, zp).to(dtype)
    assert bits_equal(got, want) == 0
    for ax in (0, 1, 2, -1):
        C = x.shape[ax]
        sc = torch.rand(C, generator=g) * 0.05 + 1e-3
        zp = torch.randint(-5, 6, (C,), generator=g)
        got = dmx.ops.fixed_qdq(x.to(cuda), 8, 0, True, True, scale=sc, zero_point=zp, ch_axis=ax)
        want = oracle.fixed_point_affine_cast(x, 8, 0, True, True, sc, zp, ch_axis=ax).to(dtype)
        assert bits_equal(got, want) == 0, ax
        for gs in (1, 2, 7, C, C + 3):  # ragged last group, one group, group larger than C
            G = -(-C // gs)
            got = dmx.ops.fixed_qdq(x.to(cuda), 4, 0, True, True, scale=sc[:G], zero_point=zp[:G], ch_axis=ax, group_size=gs)
            want = oracle.fixed_point_affine_cast(x, 4, 0, True, True, sc[:G], zp[:G], ch_axis=ax, group_size=gs).to(dtype)
            assert bits_equal(got, want) == 0, (ax, gs)


def test_fixed_rounding_fp32_only_form_matches_the_double_step(dmx, cuda, oracle):
    """The kernel evaluates nearbyint((double)(float)(a + 0.5f) - 0.5) without f64 instructions; check the regions
    where the argument is not representable in fp32: |a| around 2^22 .. 2^25, odd/even integers, halves, +-inf, nan."""
    g = torch.Generator().manual_seed(0)
    parts = [torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 0.5 + 2.0 ** -24, 0.49999997, 4194303.5, 4194304.5, 8388607.0, 8388607.5,
                           8388608.0, 8388609.0, 8388610.0, 16777215.0, 16777216.0, 16777218.0, 3e9, -3e9, float("inf"),
                           float("-inf"), float("nan"), 1e30, -1e30])]
    for lo, hi in ((21, 22), (22, 23), (23, 24), (24, 25), (0, 3)):
        m = torch.rand(20000, generator=g) * (2.0 ** hi - 2.0 ** lo) + 2.0 ** lo
        parts += [m, -m, torch.floor(m), -torch.floor(m), torch.floor(m) + 0.5, -(torch.floor(m) + 0.5)]
    x = torch.cat(parts)
    for wl, fl, clamp, sym in ((24, 0, False, False), (24, 0, True, True), (16, -8, False, True), (20, 4, False, False), (8, -20, False, True)):
        for rounding in ("nearest", "stochastic"):
            got = dmx.ops.fixed_qdq(x.to(cuda), wl, fl, clamp, sym, rounding, seed=5)
            want = oracle.fixed_point_cast(x, wl, fl, clamp, sym, rounding, seed=5)
            both_nan = torch.isnan(got.cpu()) & torch.isnan(want)
            assert int(((got.cpu().view(torch.int32) != want.view(torch.int32)) & ~both_nan).sum()) == 0, (wl, fl, rounding)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fixed_affine_scale_lookup_modes(dmx, cuda, oracle, dtype):
    """One case per scale-lookup mode of the kernel (elementwise.hip ChanMode): vector inside one channel
    (weights, ch_axis 0), N consecutive channels (per-channel along the last dim), and the carrying walker."""
    g = torch.Generator().manual_seed(2)
    for shape, ax, gs in (((48, 64), 0, None), ((48, 64), 0, 16), ((48, 64), 0, 7), ((32, 64), -1, None), ((6, 32, 64), -1, None),
                          ((6, 32, 64), 1, None), ((6, 32, 64), 1, 5), ((5, 24, 12), 1, 4), ((33, 40), -1, 8), ((9, 7, 3), 2, None)):
        x = make("normal", shape, seed=len(shape) + (gs or 0), dtype=dtype)
        C = shape[ax]
        G = -(-C // (gs or 1))
        sc = torch.rand(G, generator=g) * 0.05 + 1e-3
        zp = torch.randint(-5, 6, (G,), generator=g)
        got = dmx.ops.fixed_qdq(x.to(cuda), 8, 0, True, False, scale=sc, zero_point=zp, ch_axis=ax, group_size=gs)
        want = oracle.fixed_point_affine_cast(x, 8, 0, True, False, sc, zp, ch_axis=ax, group_size=gs).to(dtype)
        assert bits_equal(got, want) == 0, (shape, ax, gs)
        s = torch.rand(C, generator=g) + 0.5
        sh = [1] * len(shape)
        sh[ax] = C
        got = dmx.ops.scale_channels(x.to(cuda), s, ax, divide=True)
        assert bits_equal(got, (x.float() / s.view(sh)).to(dtype)) == 0, (shape, ax)
        got = dmx.ops.scale_channels(x.to(cuda), s, ax, divide=False)
        assert bits_equal(got, (x.float() * s.view(sh)).to(dtype)) == 0, (shape, ax)


def test_reference_known_answers_group_quant(dmx, cuda):
    """tests/test_group_quant.py:49-63 of the reference: INT4, group_size 2 along dim 0, MinMax symmetric."""
    x = torch.tensor([[0, 1], [3, 7], [5.1, 8], [10, 14], [0.1, 0.7]])
    y = torch.tensor([[0, 1], [3, 7], [6, 8], [10, 14], [0.1, 0.7]])
    cast = dmx.CastTo(format=dmx.format.INT4, observer=dmx.MinMaxObserver, group_size=2,
                      qscheme=torch.per_tensor_symmetric, ch_axis=0)
    cast.enable_observer()
    out = cast(x.to(cuda))
    assert torch.allclose(out.cpu(), y, rtol=0.0, atol=1e-6)


def test_reference_known_answers_bfp_block1(dmx, cuda):
    """tests/test_bfp.py:26-65 of the reference (block size 1 -> float_quantize path)."""
    x = torch.tensor([1.0, 1.0 + 2 ** -7, 1.0 + 2 ** -6, 1.0 + 2 ** -6 + 2 ** -7]).to(cuda)
    y = torch.tensor([1.0, 1.0, 1.015625, 1.03125]).to(cuda)
    c = dmx.CastTo(format="BFP[8|8]{1}(SN)")
    assert torch.all(c(x) == y) and torch.all(c(-x) == -y)
    x = torch.tensor([1.0, 1.0 + 2 ** -3, 1.0 + 2 ** -2, 1.0 + 2 ** -2 + 2 ** -3]).to(cuda)
    y = torch.tensor([1.0, 1.0, 1.25, 1.5]).to(cuda)
    c = dmx.CastTo(format="BFP[4|8]{1}(SN)")
    assert torch.all(c(x) == y) and torch.all(c(-x) == -y)
    # test_bfp.py:11-23
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 1000), generator=g)
    x *= 0.5 / x.abs().max()
    x += 1.0
    x = x.to(cuda)
    assert torch.allclose(dmx.CastTo(format="BFP[8|8]{1}(SN)")(x), x, rtol=0.0, atol=2 ** -7)


def test_scale_channels(dmx, cuda):
    x = make("normal", (6, 40, 9), seed=2, dtype=torch.bfloat16)
    s = (torch.rand(40) + 0.5)
    got = dmx.ops.scale_channels(x.to(cuda), s, 1, divide=True)
    want = (x.float() / s.view(1, 40, 1)).to(torch.bfloat16)
    assert bits_equal(got, want) == 0
    got = dmx.ops.scale_channels(x.to(cuda), s, 1, divide=False, out_dtype=torch.float32)
    assert bits_equal(got, x.float() * s.view(1, 40, 1)) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_per_channel_along_the_contiguous_dim(dmx, cuda, oracle, dtype):
    """Per-channel affine quantisation and SmoothQuant-style scaling along the LAST dim (lastdim_kernel: channel parameters
    in registers): channel counts that give every lanes-per-row / rows-per-workgroup / column-strip layout, ragged row
    counts, INT8 (the short form) and a fractional format (the general form)."""
    for n, (rows, C) in enumerate([(1, 8), (5, 16), (37, 24), (33, 40), (130, 256), (7, 768), (19, 2048), (9, 2056), (3, 5120),
                                   (1030, 64), (2, 16384)]):
        x = make("heavy", (rows, C), seed=40 + n, dtype=dtype)
        sc = torch.rand(C) * 0.2 + 0.01
        zp = torch.randint(-10, 10, (C,))
        for p, f in ((8, 0), (8, 3)):
            got = dmx.ops.fixed_qdq(x.to(cuda), p, f, True, True, scale=sc.to(cuda), zero_point=zp.to(cuda), ch_axis=-1)
            want = oracle.fixed_point_affine_cast(x, p, f, True, True, sc, zp, ch_axis=-1).to(dtype)
            assert bits_equal(got, want) == 0, (rows, C, p, f)
        for divide in (True, False):
            out_dtype = torch.float32 if divide else dtype
            got = dmx.ops.scale_channels(x.to(cuda), sc.to(cuda), -1, divide, out_dtype=out_dtype)
            want = (x.float() / sc if divide else x.float() * sc).to(out_dtype)
            assert bits_equal(got, want) == 0, (rows, C, divide)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_affine_int_codes_at_rounding_boundaries(dmx, cuda, oracle, dtype):
    """INT8 / INT4 affine quantisation where x / scale + zero_point sits on, or a few ulp either side of, a rounding
    boundary k + 1/2 (and at the clamp limits), for per-tensor and per-group scales of any value: the codes depend on the
    correctly rounded IEEE quotient there (a multiply-by-reciprocal shortcut was tried and measured slower; this test is
    what any such shortcut has to survive)."""
    import numpy as np
    rng = np.random.default_rng(5)
    for p in (8, 4):
        lim = 2 ** (p - 1)
        for trial in range(6):
            G = 64
            sc = torch.from_numpy(rng.uniform(1e-3, 3.0, G).astype(np.float32))
            if trial == 0:
                sc[:] = torch.tensor([0.1, 1.0 / 3.0, 0.7, 1.9999999, 1e-3, 2.5e-2, 0.3, 3.0]).repeat(8)
            zp = torch.from_numpy(rng.integers(-lim, lim, G)) if trial % 2 else torch.zeros(G, dtype=torch.int64)
            k = torch.from_numpy(rng.integers(-lim - 2, lim + 2, (G, 512)).astype(np.float32)) + 0.5
            x = (k - zp[:, None].float()) * sc[:, None]                     # x / sc + z ~ k + 1/2
            bits = x.view(torch.int32) + torch.from_numpy(rng.integers(-3, 4, (G, 512)).astype(np.int32))
            x = bits.view(torch.float32).to(dtype)
            for mode, kw in (("group", dict(ch_axis=0, group_size=1)), ("tensor", dict())):
                s_, z_ = (sc, zp) if mode == "group" else (sc[:1], zp[:1])
                got = dmx.ops.fixed_qdq(x.to(cuda), p, 0, True, True, scale=s_.to(cuda), zero_point=z_.to(cuda), **kw)
                want = oracle.fixed_point_affine_cast(x, p, 0, True, True, s_, z_, **kw).to(dtype)
                assert bits_equal(got, want) == 0, (p, trial, mode)
