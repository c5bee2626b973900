"""-m gpu, round 5: the placement-independent init gate of the reductions (csrc/reduce.hip), the per-tile straight-line forms of the
INT8 group cast (csrc/stream.hpp OpTileVariants), the compile-time rounding builds of the BFP row kernel."""
import ctypes

import pytest
import torch

from _data import bits_equal, make, mismatches_nan_aware

pytestmark = pytest.mark.gpu
BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32


def _gate_mode(dmx, mode):
    L = dmx._lib.lib()
    L.dmxq_internal_gate_mode.argtypes = [ctypes.c_int]
    L.dmxq_internal_gate_mode.restype = ctypes.c_int
    return L.dmxq_internal_gate_mode(mode)


def _reductions(dmx, x, hist_in):
    return (dmx.ops.group_minmax(x.reshape(1, -1), 0, 1), dmx.ops.group_minmax(x, 0, 64), dmx.ops.channel_maxabs(x, -1),
            dmx.ops.histc(hist_in, 2048, -4.0, 4.0))


def _expect(oracle, x, hist_in):
    return (oracle.group_minmax(x.reshape(1, -1), 0, 1), oracle.group_minmax(x, 0, 64), oracle.channel_maxabs(x, -1),
            oracle.histc(hist_in, 2048, -4.0, 4.0))


def _same(got, want):
    return (bits_equal(got[0][0], want[0][0]) == 0 and bits_equal(got[0][1], want[0][1]) == 0 and bits_equal(got[1][0], want[1][0]) == 0
            and bits_equal(got[1][1], want[1][1]) == 0 and bits_equal(got[2], want[2]) == 0 and bits_equal(got[3], want[3]) == 0)


# ------------------------------------------------------------------------------------------------ init gate
@pytest.mark.parametrize("mode", [2, 1, 0])
def test_init_gate_takeover_and_switch(dmx, cuda, oracle, mode):
    """WHO writes the identities of a gated reduction is decided by a claim on an election word, not by workgroup index.  mode 2
    (test hook) keeps workgroup (0, 0) from volunteering -- the situation of a dispatcher that has not started it yet: every launch
    must complete through the TAKEOVER path (a waiting workgroup claims the job after kGateTakeover polls) with the oracle's results.
    mode 1 = the switch DMXQ_NO_INIT_GATE sets (fill launch in front); mode 0 = normal.  The extremes sit in ONE workgroup's tile, a
    different one per launch, so a contribution that overtook the identities would be lost and show."""
    rows, cols = 2048, 4096
    base = make("normal", (rows, cols), seed=501, dtype=BF16).clamp(-8, 8)
    hist_in = make("normal", (512, 4096), seed=502, dtype=BF16)
    hd = hist_in.to(cuda)
    old = _gate_mode(dmx, mode)
    try:
        for it in range(6):
            x = base.clone()
            r = (it * 331) % rows
            x[r, :] = 50.0 + it
            x[(r + 9) % rows, 1::3] = -(70.0 + it)
            want = _expect(oracle, x, hist_in)
            xd = x.to(cuda)
            for _ in range(4):
                assert _same(_reductions(dmx, xd, hd), want), (mode, it)
        for dt in (F32, F16):
            x = make("heavy", (1024, 2048), seed=503, dtype=dt).clamp(-1e4, 1e4)
            want = _expect(oracle, x, hist_in)
            assert _same(_reductions(dmx, x.to(cuda), hd), want), (mode, dt)
    finally:
        _gate_mode(dmx, old)


def test_init_gate_under_a_filler_kernel_occupying_every_cu(dmx, cuda, oracle):
    """1,000 gated reductions on one stream while a second stream keeps every CU busy with long streaming kernels (torch's own
    elementwise kernels over 1 GiB, ~thousands of workgroups each): the reductions' workgroups are dispatched into whatever the
    filler leaves free, in whatever order.  No hang (the claim makes progress independent of placement), every result == oracle.
    Results are compared ON the device and counted, one host synchronisation at the end."""
    xs = [make("heavy", (2048, 4096), seed=520 + i, dtype=BF16).clamp(-1e4, 1e4) for i in range(4)]
    hist_in = make("normal", (512, 4096), seed=530, dtype=BF16)
    wants = [_expect(oracle, x, hist_in) for x in xs]
    dx, hd = [x.to(cuda) for x in xs], hist_in.to(cuda)
    dw = [tuple(t.to(cuda) for t in (w[0][0], w[0][1], w[1][0], w[1][1], w[2], w[3])) for w in wants]
    filler_buf = torch.ones(1 << 28, device=cuda)            # 1 GiB of float32
    s_fill, s_red = torch.cuda.Stream(), torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device=cuda)
    torch.cuda.synchronize()
    with torch.cuda.stream(s_fill):
        for _ in range(120):                                  # ~0.4 ms each: the reductions below run inside this window
            filler_buf.mul_(1.0000001)
    with torch.cuda.stream(s_red):
        for i in range(250):                                  # x 4 reductions = 1,000 gated launches
            k = i % 4
            g = _reductions(dmx, dx[k], hd)
            got = (g[0][0], g[0][1], g[1][0], g[1][1], g[2], g[3])
            for a, b in zip(got, dw[k]):
                bad += (a.view(torch.int32) != b.view(torch.int32)).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0
    # the same with the volunteer switched off: every launch goes through the takeover while the chip is busy
    old = _gate_mode(dmx, 2)
    try:
        with torch.cuda.stream(s_fill):
            for _ in range(40):
                filler_buf.mul_(1.0000001)
        with torch.cuda.stream(s_red):
            for i in range(25):
                k = i % 4
                g = _reductions(dmx, dx[k], hd)
                got = (g[0][0], g[0][1], g[1][0], g[1][1], g[2], g[3])
                for a, b in zip(got, dw[k]):
                    bad += (a.view(torch.int32) != b.view(torch.int32)).sum()
        torch.cuda.synchronize()
    finally:
        _gate_mode(dmx, old)
    assert int(bad) == 0


# ------------------------------------------------------------------------------------------------ INT8 per group: tile forms
@pytest.mark.parametrize("dtype", [BF16, F16, F32])
@pytest.mark.parametrize("shape,gs", [((4096, 4096), 128), ((768, 768), 128), ((3072, 768), 128), ((4096, 1000), 128), ((1000, 4096), 128),
                                      ((2048, 512), 2048), ((640, 4096), 64)])
def test_int8_group_tile_forms_against_the_oracle(dmx, cuda, oracle, dtype, shape, gs):
    """dmxq_fixed_qdq with one (scale, zero point) per slab of `gs` rows (cast.py:281-292): the stream kernel picks, per TILE, the
    straight-line reciprocal form without the zero-point steps (every zero point 0: symmetric schemes), the one with them, or the
    general code (a scale outside [2^-20, 2^20]); Inf / NaN / huge inputs are redone in a cold loop after the tile's stores.  All
    of them bit-exact against the oracle, on aligned runs (the scalar group lookup), ragged row lengths and ragged last groups."""
    rows, cols = shape
    x = make("heavy", shape, seed=rows + cols, dtype=dtype).clamp(-3e4, 3e4)
    x[3, 5], x[rows // 2, 7], x[rows - 1, cols - 1] = float("inf"), float("-inf"), float("nan")
    x[5, :16] = 3.0e38 if dtype != F16 else 6.0e4
    x[7, :16] = 1e-42 if dtype == F32 else 0.0
    x[8, :16] = -0.0
    G = -(-rows // gs)
    g = torch.Generator().manual_seed(rows)
    for name in ("zero", "mixed", "wild"):
        scale = torch.rand(G, generator=g) * 0.05 + 1e-3
        if name == "zero":
            zp = torch.zeros(G, dtype=torch.int64)
        else:
            zp = torch.randint(-5, 6, (G,), generator=g)
            zp[::3] = 0
        if name == "wild":
            scale[0], scale[G // 2], scale[G - 1] = 1e-9, 3e7, 2.0 ** -20    # outside / at the edge of the reciprocal form's range
        want = oracle.fixed_point_affine_cast(x, 8, 0, True, True, scale, zp, ch_axis=0, group_size=gs)
        got = dmx.ops.fixed_qdq(x.to(cuda), 8, 0, True, True, scale=scale.to(cuda), zero_point=zp.to(cuda), ch_axis=0, group_size=gs)
        # (a NaN compares equal to a NaN: torch's CPU float32 -> bfloat16 conversion, which narrows the oracle's result here, writes
        #  0xFFFF for a NaN in its vectorised part and 0x7FC0 in its tail; which ELEMENTS are NaN must agree, and every other bit)
        assert got.dtype == dtype and mismatches_nan_aware(got.float().cpu(), want.to(dtype).float()) == 0, (name, dtype, shape)
