"""-m gpu: the S1 seam (`quant.quant_hip`, the pybind-level surface of quant_cuda.cpp:116-139) against the
REFERENCE'S OWN compiled CPU extension (oracle/_ref/quant_cpu.so, built from the reference sources by
oracle/Makefile and shipped as a binary), same call signature on both sides, bit-exact."""
import os
import sys

import pytest
import torch

from _data import bits_equal, make, mismatches_nan_aware

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def quant_cpu():
    d = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.exists(os.path.join(d, "quant_cpu.so")):
        pytest.skip("oracle/_ref/quant_cpu.so not built (needs /root/reference at build time)")
    sys.path.insert(0, d)
    import quant_cpu as m

    return m


@pytest.mark.parametrize("mode", ["nearest", "down", "up"])
def test_block_quantize_seam(dmx, cuda, quant_cpu, mode):
    q = dmx.quant.quant_hip
    for shape, wl in (((4096, 16), 8), ((512, 64), 4), ((33, 40), 6), ((7, 3, 5), 16)):
        x = make("mixed", shape, seed=wl, block=shape[-1])
        want = getattr(quant_cpu, f"block_quantize_{mode}")(x, wl, 0, True)
        got = getattr(q, f"block_quantize_{mode}")(x.to(cuda), wl, 0, True)
        assert got.dtype == torch.float32 and bits_equal(got, want) == 0, (shape, wl)
    x = make("heavy", (6, 10, 4), seed=3)
    for dim in (-1, 1, 2):   # whole tensor / one exponent per index of `dim` (get_max_entry, quant_cpu.cpp:277-297)
        want = quant_cpu.block_quantize_nearest(x, 8, dim, True)
        got = q.block_quantize_nearest(x.to(cuda), 8, dim, True)
        assert bits_equal(got, want) == 0, dim


def test_float_and_fixed_seams(dmx, cuda, quant_cpu):
    q = dmx.quant.quant_hip
    x = torch.cat([make("heavy", (20000,), seed=1), torch.tensor([0.0, -0.0, 65520.0, 3e38, 1e-40, float("inf"), float("nan")])])
    for man, exp, bias, flush in ((10, 5, 15, True), (3, 4, 7, False), (2, 5, 15, False), (22, 8, 127, False)):
        want = quant_cpu.float_quantize_nearest(x, man, exp, bias, flush)
        got = q.float_quantize_nearest(x.to(cuda), man, exp, bias, flush)
        assert mismatches_nan_aware(got, want) == 0, (man, exp)
    xf = x[:-2]
    for mode in ("nearest", "up", "down"):
        for wl, fl, clamp, sym in ((8, 0, True, True), (8, 3, True, False), (12, -2, False, True)):
            want = getattr(quant_cpu, f"fixed_point_quantize_{mode}")(xf, wl, fl, clamp, sym)
            got = getattr(q, f"fixed_point_quantize_{mode}")(xf.to(cuda), wl, fl, clamp, sym)
            assert bits_equal(got, want) == 0, (mode, wl, fl)


def test_fixed_point_mask_seams(dmx, cuda, quant_cpu):
    """quant_cpu.cpp:86-126: (clamped value, uint8 mask of the clamped elements) -- the pybind functions no Python wrapper of the
    reference calls (VERDICT r2 missing-5); nearest bit-exact against the reference's compiled extension, stochastic by its
    properties (mask == clamped, values on the grid, unbiased)."""
    q = dmx.quant.quant_hip
    x = torch.cat([make("heavy", (20000,), seed=4), torch.tensor([0.0, -0.0, 127.0, 127.5, 128.0, -127.5, -128.0, -128.5, 1e30, -1e30, float("inf"), float("-inf")])])
    for wl, fl, sym in ((8, 0, True), (8, 0, False), (8, 3, False), (4, 0, True), (12, -2, True)):
        wo, wm = quant_cpu.fixed_point_quantize_nearest_mask(x, wl, fl, sym)
        go, gm = q.fixed_point_quantize_nearest_mask(x.to(cuda), wl, fl, sym)
        assert go.dtype == torch.float32 and gm.dtype == torch.uint8
        assert bits_equal(go, wo) == 0 and torch.equal(gm.cpu(), wm), (wl, fl, sym)
        assert 0 < int(wm.sum()) < x.numel()
    so, sm = q.fixed_point_quantize_stochastic_mask((x[:20000] * 4).to(cuda), 8, 0, True)
    plain = (x[:20000] * 4)
    m, o = sm.bool().cpu(), so.cpu()
    assert float(o.abs().max()) <= 127.0 and torch.equal(o, o.round())
    assert bool((o[m].abs() == 127.0).all())                                      # a masked element sits on the clamp limit
    assert bool(m[plain.abs() >= 129.0].all()) and not bool(m[plain.abs() <= 126.0].any())
    inside = ~m
    assert abs(float((so.cpu()[inside] - plain[inside]).mean())) < 2e-2            # unbiased where nothing was clamped


def test_python_wrappers_keep_the_reference_contract(dmx, cuda, quant_cpu):
    x = make("normal", (64, 32), seed=2).to(cuda)
    y = dmx.quant.block_quantize(x, 8, dim=0, symmetric=True, rounding="nearest")
    assert bits_equal(y, quant_cpu.block_quantize_nearest(x.cpu(), 8, 0, True)) == 0
    y = dmx.quant.float_quantize(x, 5, 10, rounding="nearest")                    # bias defaults to 2^(e-1)-1
    assert bits_equal(y, quant_cpu.float_quantize_nearest(x.cpu(), 10, 5, 15, True)) == 0
    y = dmx.quant.fixed_point_quantize(x, 8, 4, rounding="nearest")               # clamp=True, symmetric=False defaults
    assert bits_equal(y, quant_cpu.fixed_point_quantize_nearest(x.cpu(), 8, 4, True, False)) == 0
    with pytest.raises(RuntimeError):
        dmx.quant.quant_hip.block_quantize_nearest(x.t(), 8, 0, True)              # CHECK_CONTIGUOUS
    with pytest.raises(RuntimeError):
        dmx.quant.quant_hip.block_quantize_nearest(x.half(), 8, 0, True)           # data_ptr<float>()


def test_stochastic_rounding_is_unbiased_like_the_reference(dmx, cuda, quant_cpu):
    """Statistical parity only: the reference's RNG is an unseeded global mt19937."""
    x = torch.full((200000,), 0.3)
    ref = quant_cpu.fixed_point_quantize_stochastic(x, 8, 0, True, True)
    got = dmx.quant.quant_hip.fixed_point_quantize_stochastic(x.to(cuda), 8, 0, True, True).cpu()
    assert set(got.unique().tolist()) <= {0.0, 1.0} and set(ref.unique().tolist()) <= {0.0, 1.0}
    assert abs(float(got.mean()) - 0.3) < 5e-3 and abs(float(ref.mean()) - 0.3) < 5e-3
    xb = (torch.rand(4096, 16) * 2 - 1)
    g = dmx.quant.quant_hip.block_quantize_stochastic(xb.to(cuda), 6, 0, True).cpu()
    r = quant_cpu.block_quantize_stochastic(xb, 6, 0, True)
    assert abs(float((g - xb).mean())) < 2e-3 and abs(float((r - xb).mean())) < 2e-3   # both unbiased
    assert float((g - xb).abs().max()) <= float((r - xb).abs().max()) * 1.01 + 1e-6
