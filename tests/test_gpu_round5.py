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


# ------------------------------------------------------------------------------------------------ live weights as a set (nn.LiveWeightBatch)
class _Stack(torch.nn.Module):
    def __init__(self, dmx, dims, dtype, bias=True):
        super().__init__()
        self.layers = torch.nn.ModuleList([dmx.nn.Linear(a, b, bias=bias) for a, b in zip(dims[:-1], dims[1:])])
        for i, l in enumerate(self.layers):
            l.weight.data = (make("normal", tuple(l.weight.shape), seed=900 + i) * 0.05).to(dtype)
            if bias:
                l.bias.data = (make("normal", tuple(l.bias.shape), seed=950 + i) * 0.02).to(dtype)

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x


def _configure_basic(dmx, model):
    for m in model.modules():
        if isinstance(m, dmx.nn.DmxModule):
            for r in dmx.config_rules.BASIC:
                if isinstance(m, r.module_types):
                    m.configure(r.module_config)


@pytest.mark.parametrize("kind", ["int8_group", "bfp", "nm_bfp"])
def test_live_weight_batch_equals_the_per_module_path(dmx, cuda, kind):
    """Un-folded weights re-quantised every forward (modeling/nn/core.py:178-203) as ONE multi-tensor launch per group of sibling
    weights (nn.LiveWeightBatch: opt-125m style INT8 per row group, plain BFP, Llama style 2:4 mask -> BFP) == the modules' own
    per-weight launches, bit for bit; a weight changed between two forwards shows in the next one; GraphedForward installs it."""
    dtype = F32 if kind == "int8_group" else BF16
    model = _Stack(dmx, [768, 768, 3072, 768], dtype, bias=kind != "nm_bfp").to(cuda).eval()
    _configure_basic(dmx, model)
    if kind == "int8_group":
        hp = dmx.nn.DmxModuleQuantizerCalibrationHyperparams(weight=dmx.nn.DmxQuantizerCalibrationHyperparams(
            observer_cls=dmx.MinMaxObserver, qscheme_to_overload=torch.per_tensor_symmetric, group_size=128, ch_axis=0))
        for l in model.layers:
            l.configure(dict(weight_format=dmx.format.INT8))
            with l.calibrating_quantizers(hp), torch.no_grad():
                l._weight
    elif kind == "nm_bfp":
        for i, l in enumerate(model.layers):
            l.configure(dict(weight_sparseness="BTOPK{2:4,-1}(U)"))
            with torch.no_grad():
                l.weight_sparsifier(l.weight)
                l.weight_sparsifier.score.data = make("normal", tuple(l.weight.shape), seed=970 + i).abs().to(cuda)
    x = (make("heavy", (2, 128, 768), seed=990).clamp(-100, 100)).to(dtype).to(cuda)
    with torch.no_grad():
        want = model(x).clone()
        want_w = [l._weight_ro.clone() for l in model.layers]
        batch = dmx.nn.LiveWeightBatch(model)
        seen = {}
        # (the stamps live for the scope's forward only -- round 6 -- so they are looked at from INSIDE it: before the last layer runs)
        def look(mod, args):
            seen.update({i: (l.__dict__["_live_weight"][0], l.weight_hypernet(l.weight, dtype), l._weight_ro)
                         for i, l in enumerate(model.layers) if "_live_weight" in l.__dict__})

        probe = model.layers[-1].register_forward_pre_hook(look)
        got = model(x)
        probe.remove()
        assert sorted(seen) == list(range(len(model.layers))), "every sibling weight should have been batched"
        for i, w in enumerate(want_w):   # (what Linear._forward asks for: the weight rounded to the input's dtype)
            stamped, asked, ro = seen[i]
            assert asked is stamped and bits_equal(asked, w.to(dtype)) == 0
            assert ro.dtype == w.dtype and bits_equal(ro, w) == 0
        assert bits_equal(got, want) == 0
        # ... and are gone when it ends: nothing stale for a direct call of a submodule, no quantised copy held between forwards
        assert not any("_live_weight" in l.__dict__ or "_live_bias" in l.__dict__ for l in model.layers)
        # a weight changed in place (a version bump, as an optimiser step makes): the next forward re-quantises everything -- the live
        # semantics of the reference -- and a stamp that belongs to the previous version is refused even inside a forward
        stale = seen[1][0]
        model.layers[1].weight.mul_(1.5)
        model.layers[1].__dict__["_live_weight"] = (stale, model.layers[1].weight._version - 1, model.layers[1].weight.data_ptr(), dtype)
        fresh = model.layers[1].weight_hypernet(model.layers[1].weight, dtype)   # NOT the stamped result
        assert "_live_weight" not in model.layers[1].__dict__ and bits_equal(fresh.to(dtype), stale) != 0
        batch.remove()
        want2 = model(x).clone()
        batch = dmx.nn.LiveWeightBatch(model)
        got2 = model(x)
        assert bits_equal(got2, want2) == 0 and bits_equal(got2, want) != 0
        batch.remove()
        assert not any("_live_weight" in l.__dict__ for l in model.layers) and not model._forward_pre_hooks and not model._forward_hooks
        kept = dmx.nn.LiveWeightBatch(model, replan=False)     # the plan of the first forward serves the following ones
        assert bits_equal(model(x), want2) == 0 and kept._plan is not None and bits_equal(model(x), want2) == 0
        kept.remove()
        g = dmx.nn.GraphedForward(model, x)
        assert g.live_batch_scopes == 1
        assert not model._forward_pre_hooks and not model._forward_hooks     # installed for the capture only (round 6)
        assert bits_equal(g(x), want2) == 0
        assert bits_equal(g(x * 0.5), model(x * 0.5)) == 0
        # a constructor that raises leaves no hook behind either
        model.layers[0].input_casts.input_cast.enable_observer()
        with pytest.raises(RuntimeError):
            dmx.nn.GraphedForward(model, x)
        model.layers[0].input_casts.input_cast.disable_observer()
        assert not model._forward_pre_hooks and not model._forward_hooks


def test_live_weight_stamps_do_not_outlive_the_forward(dmx, cuda):
    """a stamp knows the Parameter's storage and version only.  After `root(x)`, a recalibrated scale (updated in place), a
    new sparsifier score or a reconfigured format must show in ANY later call -- of the root, of a submodule, of `_weight_ro` -- and a
    forward that raises must not leave stamps behind either.  Scopes: a budget smaller than the model batches layer by layer."""
    model = _Stack(dmx, [768, 768, 3072, 768], F32).to(cuda).eval()
    _configure_basic(dmx, model)
    hp = dmx.nn.DmxModuleQuantizerCalibrationHyperparams(weight=dmx.nn.DmxQuantizerCalibrationHyperparams(
        observer_cls=dmx.MinMaxObserver, qscheme_to_overload=torch.per_tensor_symmetric, group_size=128, ch_axis=0))
    for l in model.layers:
        l.configure(dict(weight_format=dmx.format.INT8))
        with l.calibrating_quantizers(hp), torch.no_grad():
            l._weight
    x = make("normal", (4, 768), seed=1700).to(cuda)
    with torch.no_grad():
        batch = dmx.nn.LiveWeightBatch(model)
        assert batch.scopes == [model]
        y0 = model(x).clone()
        l1 = model.layers[1]
        before = l1._weight_ro.clone()
        l1.weight_cast.scale.mul_(2.0)          # recalibration updates the buffer in place: the Parameter's version does not move
        after = l1._weight_ro
        assert bits_equal(after, before) != 0 and bits_equal(after, l1.weight_cast(l1.weight)) == 0
        h = model.layers[0](x)
        direct = l1(h).clone()                  # a direct call of the submodule: its own chain, with the new scale
        y1 = model(x)
        assert bits_equal(y1, y0) != 0
        batch.remove()
        assert bits_equal(model(x), y1) == 0      # batched forward == per-module forward after the change
        assert bits_equal(l1(h), direct) == 0
        # a forward that raises: always_call releases the stamps
        batch = dmx.nn.LiveWeightBatch(model)
        def explode(mod, a):
            raise ValueError("boom")

        boom = model.layers[-1].register_forward_pre_hook(explode)
        with pytest.raises(ValueError):
            model(x)
        boom.remove()
        assert not any("_live_weight" in l.__dict__ or "_live_bias" in l.__dict__ for l in model.layers)
        batch.remove()

        # scopes by weight budget: two blocks of two layers each; a budget below the model's weights batches block by block
        class Blocks(torch.nn.Module):
            def __init__(self, a, b):
                super().__init__()
                self.blocks = torch.nn.ModuleList([torch.nn.Sequential(*a), torch.nn.Sequential(*b)])

            def forward(self, t):
                for blk in self.blocks:
                    t = blk(t)
                return t

        m2 = _Stack(dmx, [768, 768, 768, 768, 768], F32).to(cuda).eval()
        _configure_basic(dmx, m2)
        big = Blocks(list(m2.layers[:2]), list(m2.layers[2:]))
        want = big(x).clone()
        per_block = 2 * 768 * 768 * 4
        b2 = dmx.nn.LiveWeightBatch(big, max_bytes=per_block)
        assert b2.scopes == list(big.blocks)
        alive = []
        def look(mod, a):
            alive.append([("_live_weight" in l.__dict__) for l in m2.layers])

        probe = m2.layers[3].register_forward_pre_hook(look)
        assert bits_equal(big(x), want) == 0
        probe.remove()
        assert alive == [[False, False, True, True]]     # the first block's copies were released before the second block's were made
        b2.remove()
        assert dmx.nn.LiveWeightBatch(big, max_bytes=None).scopes == [big]


# ------------------------------------------------------------------------------------------------ x / s through the reciprocal
def _every_exponent(n, dtype, seed):
    """values with every float32 exponent (denormals, the extremes), random and all-ones / all-zeros mantissas, both signs, zeros, Inf, NaN"""
    g = torch.Generator().manual_seed(seed)
    e = torch.randint(0, 255, (n,), generator=g, dtype=torch.int64)
    m = torch.randint(0, 1 << 23, (n,), generator=g, dtype=torch.int64)
    m[::5] = (1 << 23) - 1
    m[1::5] = 0
    sgn = torch.randint(0, 2, (n,), generator=g, dtype=torch.int64)
    bits = ((sgn << 31) | (e << 23) | m).to(torch.int64)
    bits = torch.where(bits >= (1 << 31), bits - (1 << 32), bits).to(torch.int32)
    x = bits.view(torch.float32).clone()
    x[7::97] = 0.0
    x[11::101] = -0.0
    x[13::103] = float("inf")
    x[17::107] = -float("inf")
    x[19::109] = float("nan")
    return x.to(dtype)


@pytest.mark.parametrize("dtype", [BF16, F16, F32])
def test_scale_channels_divide_is_the_ieee_quotient_for_every_input(dmx, cuda, oracle, dtype):
    """SmoothQuant's x / s (smoothquant.py:255-268) comes from the lane's reciprocals since round 5 (common.hpp div_by_recip: Markstein's
    correction makes it RN(x / s) inside a stated operand range, an IEEE division redoes everything else).  Bit for bit against torch's
    CPU division -- IEEE -- on inputs of EVERY exponent, all-ones mantissas, zeros of both signs, Inf, NaN, and scales inside, at the
    edge of and outside the reciprocal's range; same for the fused x / s -> BFP16_64 input path against the oracle composed like the
    reference, and for the transposed layout (scale along dim 0: the flat kernel)."""
    rows, C = 1024, 2048
    x = _every_exponent(rows * C, dtype, 77).reshape(rows, C)
    g = torch.Generator().manual_seed(5)
    s = torch.exp(torch.rand(C, generator=g) * 18.0 - 11.0)     # 1.7e-5 .. 1.1e3
    s[3], s[64], s[65], s[1000], s[2047] = 2.0 ** -20, 2.0 ** 20, 1e-9, 3e7, 1.0
    s[5::64] = torch.tensor(2.0 - 2.0 ** -23)                      # all-ones mantissa
    want = x.float() / s
    got = dmx.ops.scale_channels(x.to(cuda), s.to(cuda), -1, True, out_dtype=F32)
    assert mismatches_nan_aware(got, want) == 0
    if dtype != F32:
        got16 = dmx.ops.scale_channels(x.to(cuda), s.to(cuda), -1, True, out_dtype=dtype)
        assert mismatches_nan_aware(got16.float(), want.to(dtype).float()) == 0
    gotT = dmx.ops.scale_channels(x.t().contiguous().to(cuda), s.to(cuda), 0, True, out_dtype=F32)
    assert mismatches_nan_aware(gotT, want.t().contiguous()) == 0
    # the fused activation path: BFP16_64(x / s) in float32 (finite inputs: a block with an Inf / NaN is all-NaN in both)
    xf = torch.where(torch.isfinite(x.float()), x.float(), torch.zeros(())).to(dtype)
    fused = dmx.ops.input_hypernet(xf.to(cuda), s.to(cuda), 8, 64)
    assert fused is not None and fused.dtype == F32
    assert mismatches_nan_aware(fused, oracle.bfp_cast(xf.float() / s, 8, 64, -1)) == 0


def test_lut_module_refuses_a_capture_that_would_change_its_bits(dmx, cuda):
    """A 16-bit unary module whose policy is the table (`lut_activation = True`, the default) must not silently fall back to the direct
    kernel inside a stream capture that finds no table (eager and captured forwards would differ in last bits): it raises;
    after one eager forward -- or with the speed-only policy "auto" -- the capture goes through."""
    m = dmx.nn.GELU().to(cuda).eval()
    dmx.configure_model(m, *dmx.config_rules.BASIC)
    x = make("normal", (64, 1024), seed=3, dtype=BF16).to(cuda)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="table"):
            with torch.cuda.graph(g, stream=side):
                m(x)
        torch.cuda.synchronize()
        want = m(x).clone()            # builds the table
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            y = m(x)
        g2.replay()
        torch.cuda.synchronize()
        assert bits_equal(y, want) == 0
    torch.cuda.current_stream().wait_stream(side)


# ------------------------------------------------------------------------------------------------ multi-tensor casts on the stream skeleton
@pytest.mark.parametrize("dtype,out_dtype", [(BF16, None), (F32, None), (F16, None), (BF16, F32), (F32, BF16)])
def test_float_qdq_multi_equals_one_call_per_tensor_and_the_oracle(dmx, cuda, oracle, dtype, out_dtype):
    """`dmxq_float_qdq_multi` (round 5: the bias casts of a layer's modules, modeling/nn/core.py:191-203, as one launch): the result of one
    `float_qdq` per tensor, bit for bit, and the oracle's -- 40 tensors (two launches of the 32-tensor argument block) from 1 element to
    600,000 (partial tiles, three tile plans by total size), a ragged one (not a whole vector), a view that starts mid-allocation and an
    empty one ride along on single launches."""
    sizes = [768, 3072, 768, 2304, 8, 16, 1, 5, 4096, 100000, 600000, 1000, 0, 12345] + [768 + 8 * i for i in range(26)]
    xs = [make("heavy", (n,), seed=1200 + i).clamp(-6e4, 6e4).to(dtype) for i, n in enumerate(sizes)]
    dev = [x.to(cuda) for x in xs]
    base = make("heavy", (4099,), seed=1300).to(dtype).to(cuda)
    dev[3] = base[3:3 + 2304]                     # 6- or 12-byte offset: not 16-byte aligned
    xs[3] = dev[3].cpu()
    for man, exp, bias, flush in ((10, 5, 15, True), (3, 4, 7, False), (2, 5, 15, True)):
        got = dmx.ops.float_qdq_multi(dev, man, exp, bias, flush, out_dtype=out_dtype)
        assert len(got) == len(xs)
        for i, (x, d, y) in enumerate(zip(xs, dev, got)):
            assert y.shape == x.shape and y.dtype == (out_dtype or dtype)
            if x.numel() == 0:
                continue
            single = dmx.ops.float_qdq(d, man, exp, bias, flush, out_dtype=out_dtype)
            assert mismatches_nan_aware(y, single) == 0, (i, sizes[i], man, exp)
            if i < 16:
                want = oracle.float_quantize(x.float(), man, exp, bias, flush).to(out_dtype or dtype)
                assert mismatches_nan_aware(y, want) == 0, (i, sizes[i], man, exp)
    # other roundings: every tensor on its own launch, still the single-call results (nearest is the only batched mode)
    got = dmx.ops.float_qdq_multi(dev[:4], 3, 4, 7, False, rounding="down", out_dtype=out_dtype)
    for d, y in zip(dev[:4], got):
        assert mismatches_nan_aware(y, dmx.ops.float_qdq(d, 3, 4, 7, False, rounding="down", out_dtype=out_dtype)) == 0


def test_fixed_qdq_multi_every_tile_plan_and_outer_dims_through_the_c_abi(dmx, cuda, oracle):
    """`dmxq_fixed_qdq_multi` on the stream skeleton (round 5): sets whose TOTAL size selects each tile plan (256 x 1, 256 x 4, the op's
    128 x 16, 256 x 2 beyond 32 MiB), float32 and bf16, [outer, C, inner] descriptors with outer > 1 (batched since round 5), groups that do
    not divide C, one group for the whole tensor -- against one `dmxq_fixed_qdq` call per tensor, bit for bit, and the oracle for the
    small ones."""
    lib, L = dmx._lib, dmx._lib.lib()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dtype, code in ((F32, lib.F32), (BF16, lib.BF16)):
        for shapes, gs in (([(1, 64, 128), (1, 24, 64), (3, 8, 16)], 4),                       # tiny: 256 x 1
                           ([(1, 768, 768), (1, 3072, 768), (2, 100, 256), (1, 30, 64)], 128),  # 256 x 4 (30 rows: one short group ... C % gs != 0)
                           ([(1, 4096, 1024), (1, 1024, 4096), (1, 2048, 1024), (1, 300, 1000)], 128),   # the op's own plan
                           ([(1, 4096, 4096), (1, 4096, 2048), (1, 1, 4096)], 128)):            # beyond 32 MiB; a one-row tensor
            ws, outs, singles, scs, zps, descs = [], [], [], [], [], (lib.AffineDesc * len(shapes))()
            for i, (outer, C, inner) in enumerate(shapes):
                w = (make("normal", (outer, C, inner), seed=1400 + i) * 0.05).to(dtype)
                G = -(-C // gs)
                mn, mx = oracle.group_minmax(w[0].float().reshape(C, inner), 0, gs)
                sc, zp = oracle.qparams(mn, mx, 8, True, True)
                assert sc.numel() == G
                ws.append(w.to(cuda)); scs.append(sc.to(cuda)); zps.append(zp.to(cuda))
                outs.append(torch.empty_like(ws[-1])); singles.append(torch.empty_like(ws[-1]))
                d = descs[i]
                d.in_, d.out, d.scale, d.zero_point, d.outer, d.C, d.inner = ws[-1].data_ptr(), outs[-1].data_ptr(), scs[-1].data_ptr(), zps[-1].data_ptr(), outer, C, inner
            assert L.dmxq_fixed_qdq_multi(descs, len(shapes), code, code, 8, 0, 1, 1, lib.ROUND_NEAREST, gs, 0, sp) == lib.OK
            for i, (outer, C, inner) in enumerate(shapes):
                assert L.dmxq_fixed_qdq(ctypes.c_void_p(ws[i].data_ptr()), ctypes.c_void_p(singles[i].data_ptr()), code, code, outer, C, inner, 8, 0, 1, 1,
                                        lib.ROUND_NEAREST, ctypes.c_void_p(scs[i].data_ptr()), ctypes.c_void_p(zps[i].data_ptr()), gs, 0, sp) == lib.OK
                assert bits_equal(outs[i], singles[i]) == 0, (dtype, shapes[i])
                if outer * C * inner <= 1 << 21:
                    want = torch.stack([oracle.fixed_point_affine_cast(ws[i][o].cpu().float(), 8, 0, True, True, scs[i].cpu(), zps[i].cpu(), ch_axis=0, group_size=gs)
                                        for o in range(outer)]).to(dtype)
                    assert bits_equal(outs[i], want) == 0, (dtype, shapes[i])


def test_live_weight_batch_also_batches_the_bias_casts(dmx, cuda):
    """The bias casts of a layer's modules as ONE `float_qdq_multi` launch per forward (nn.LiveWeightBatch, round 5): stamped results equal
    the modules' own casts, a bias changed in place is re-cast by the next forward, `remove()` drops the stamps."""
    model = _Stack(dmx, [768, 768, 3072, 768], F32, bias=True).to(cuda).eval()
    _configure_basic(dmx, model)
    assert repr(model.layers[0].bias_format) == "BFP[24|8]{1}(SN)"   # BASIC: blocks of one element = float_quantize with 22 mantissa bits
    for l in model.layers[1:]:
        l.configure(dict(bias_format="FP[1|5|10,15](FN)"))   # (a 2-byte minifloat on float32 biases; two groups: [0] stays alone, unbatched)
    model.layers[0].configure(dict(bias_format="BFP[24|8]{1}(SN)"))
    x = make("normal", (4, 768), seed=1500).to(cuda)
    with torch.no_grad():
        want_b = [l.bias_cast(l.bias).clone() for l in model.layers]
        want = model(x).clone()
        batch = dmx.nn.LiveWeightBatch(model)
        seen = {}
        # (stamps live for the forward only -- round 6: looked at from inside it, before the last layer runs)
        def look(mod, args):
            seen.clear()
            seen.update({i: (l.__dict__["_live_bias"][0], l._bias_ro) for i, l in enumerate(model.layers) if "_live_bias" in l.__dict__})

        probe = model.layers[-1].register_forward_pre_hook(look)
        got = model(x)
        assert sorted(seen) == [1, 2]
        for i, b in enumerate(want_b):
            l = model.layers[i]
            if i in seen:
                assert seen[i][1] is seen[i][0] and bits_equal(seen[i][0], b) == 0
            assert bits_equal(l._bias_ro, b) == 0 and bits_equal(b, l.bias) != 0
        assert bits_equal(got, want) == 0
        assert not any("_live_bias" in l.__dict__ for l in model.layers)
        model.layers[1].bias.add_(0.37)
        ref2 = [l.bias_cast(l.bias).clone() for l in model.layers]
        model(x)
        assert sorted(seen) == [1, 2] and all(bits_equal(seen[i][0], ref2[i]) == 0 for i in seen)
        # BFP32_1 on every module (the BASIC configuration): one group of three
        for l in model.layers:
            l.configure(dict(bias_format="BFP[24|8]{1}(SN)"))
        ref3 = [l.bias_cast(l.bias).clone() for l in model.layers]
        model(x)
        assert sorted(seen) == [0, 1, 2] and all(bits_equal(seen[i][0], b) == 0 and bits_equal(b, model.layers[i].bias) != 0 for i, b in enumerate(ref3))
        probe.remove()
        batch.remove()
        assert not any("_live_bias" in l.__dict__ for l in model.layers)


def test_fixed_and_float_casts_of_a_layer_in_one_launch(dmx, cuda, oracle):
    """`dmxq_fixed_float_qdq_multi` (round 5): a layer's INT8-per-row-group weight casts and its float bias casts in ONE launch == the two
    multi-tensor calls == one call per tensor, bit for bit, float32 and bf16; sets the combined form does not take (13 weights; a ragged
    bias) fall back to the two calls with the same results; LiveWeightBatch uses it for an opt-125m style layer."""
    for dtype in (F32, BF16):
        shapes = [(768, 768)] * 4 + [(3072, 768), (768, 3072)]
        ws = [(make("normal", s, seed=1600 + i) * 0.05).to(dtype).to(cuda) for i, s in enumerate(shapes)]
        scs, zps = [], []
        for w in ws:
            mn, mx = oracle.group_minmax(w.cpu().float(), 0, 128)
            sc, zp = oracle.qparams(mn, mx, 8, True, True)
            scs.append(sc.to(cuda)); zps.append(zp.to(cuda))
        bs = [(make("normal", (s[0],), seed=1650 + i) * 0.02).to(dtype).to(cuda) for i, s in enumerate(shapes)]
        wo, bo = dmx.ops.fixed_float_qdq_multi(ws, 8, 0, True, True, scs, zps, 128, bs, 10, 5, 15, True)
        want_w = dmx.ops.fixed_qdq_multi(ws, 8, 0, True, True, scs, zps, group_size=128)
        want_b = dmx.ops.float_qdq_multi(bs, 10, 5, 15, True)
        for i in range(len(ws)):
            assert bits_equal(wo[i], want_w[i]) == 0 and bits_equal(bo[i], want_b[i]) == 0, (dtype, i)
            assert bits_equal(wo[i], dmx.ops.fixed_qdq(ws[i], 8, 0, True, True, scale=scs[i], zero_point=zps[i], ch_axis=0, group_size=128)) == 0
            assert bits_equal(bo[i], dmx.ops.float_qdq(bs[i], 10, 5, 15, True)) == 0
        want = oracle.fixed_point_affine_cast(ws[4].cpu().float(), 8, 0, True, True, scs[4].cpu(), zps[4].cpu(), ch_axis=0, group_size=128).to(dtype)
        assert bits_equal(wo[4], want) == 0
        assert mismatches_nan_aware(bo[4], oracle.float_quantize(bs[4].cpu().float(), 10, 5, 15, True).to(dtype)) == 0
        # 13 weights: more than the combined form takes -> the two calls; a bias of 771 elements (not whole vectors) likewise
        w13, s13, z13 = (ws + ws + ws[:1]), (scs + scs + scs[:1]), (zps + zps + zps[:1])
        wo2, bo2 = dmx.ops.fixed_float_qdq_multi(w13, 8, 0, True, True, s13, z13, 128, bs, 10, 5, 15, True)
        assert all(bits_equal(a, want_w[i % 6]) == 0 for i, a in enumerate(wo2)) and all(bits_equal(a, b) == 0 for a, b in zip(bo2, want_b))
        ragged = bs[:3] + [(make("normal", (771,), seed=1700) * 0.02).to(dtype).to(cuda)]
        wo3, bo3 = dmx.ops.fixed_float_qdq_multi(ws, 8, 0, True, True, scs, zps, 128, ragged, 10, 5, 15, True)
        assert all(bits_equal(a, b) == 0 for a, b in zip(wo3, want_w)) and bits_equal(bo3[3], dmx.ops.float_qdq(ragged[3], 10, 5, 15, True)) == 0


def test_fixed_float_multi_rejects_bad_arguments_before_any_launch(dmx, cuda):
    """an argument error in EITHER list of `dmxq_fixed_float_qdq_multi` is reported before the first launch: the weights' outputs stay untouched"""
    lib, L = dmx._lib, dmx._lib.lib()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    w = torch.ones(256, 64, device=cuda); wo = torch.full_like(w, 7.0)
    sc = torch.full((2,), 0.05, device=cuda); zp = torch.zeros(2, dtype=torch.int64, device=cuda)
    b = torch.ones(64, device=cuda); bo = torch.full_like(b, 7.0)
    fd = (lib.AffineDesc * 1)(); fd[0].in_, fd[0].out, fd[0].scale, fd[0].zero_point, fd[0].outer, fd[0].C, fd[0].inner = w.data_ptr(), wo.data_ptr(), sc.data_ptr(), zp.data_ptr(), 1, 256, 64
    td = (lib.TensorDesc * 1)(); td[0].in_, td[0].out, td[0].outer, td[0].L, td[0].inner = b.data_ptr(), bo.data_ptr(), 1, -64, 1   # negative extent
    assert L.dmxq_fixed_float_qdq_multi(fd, 1, 8, 0, 1, 1, lib.ROUND_NEAREST, 128, td, 1, 10, 5, 15, 1, 0, lib.ROUND_NEAREST, lib.F32, 0, sp) == lib.ERR_BAD_ARG
    td[0].L = 64
    assert L.dmxq_fixed_float_qdq_multi(fd, 1, 8, 0, 1, 1, lib.ROUND_NEAREST, 128, td, 1, 10, 9, 15, 1, 0, lib.ROUND_NEAREST, lib.F32, 0, sp) == lib.ERR_BAD_ARG  # 9 exponent bits
    torch.cuda.synchronize()
    assert bool((wo == 7.0).all()) and bool((bo == 7.0).all())
    assert L.dmxq_fixed_float_qdq_multi(fd, 1, 8, 0, 1, 1, lib.ROUND_NEAREST, 128, td, 1, 10, 5, 15, 1, 0, lib.ROUND_NEAREST, lib.F32, 0, sp) == lib.OK
    torch.cuda.synchronize()
    assert bits_equal(wo, dmx.ops.fixed_qdq(w, 8, 0, True, True, scale=sc, zero_point=zp, ch_axis=0, group_size=128)) == 0 and bits_equal(bo, dmx.ops.float_qdq(b, 10, 5, 15, True)) == 0
