"""-m gpu, round 3: the weight chain of a conv layer as one launch (dmxq_weight_hypernet_strided), the multi-tensor fold's scale
check, DMXQ_ERR_PENDING."""
import ctypes

import pytest
import torch

from _data import bits_equal, make

pytestmark = pytest.mark.gpu
BF16, F32, F16 = torch.bfloat16, torch.float32, torch.float16

# (constructor, weight shape, N:M shorthand along dim 1): Whisper-small conv1 / conv2, LeNet-5 conv2, a ResNet-style 3x3
CONVS = [
    ("whisper conv1 [768,80,3]", lambda nn: nn.Conv1d(80, 768, 3, padding=1), (768, 80, 3), "BTOPK{2:4,1}(U)"),
    ("whisper conv2 [768,768,3]", lambda nn: nn.Conv1d(768, 768, 3, stride=2, padding=1), (768, 768, 3), "BTOPK{4:8,1}(U)"),
    ("lenet conv2 [16,6,5,5]", lambda nn: nn.Conv2d(6, 16, 5), (16, 6, 5, 5), "BTOPK{1:2,1}(U)"),
    ("resnet 3x3 [64,128,3,3]", lambda nn: nn.Conv2d(128, 64, 3, padding=1), (64, 128, 3, 3), "BTOPK{2:4,1}(U)"),
]


@pytest.mark.parametrize("wdt", [F32, BF16, F16])
@pytest.mark.parametrize("tag,ctor,shape,nm", CONVS, ids=[c[0] for c in CONVS])
def test_conv_weight_chain_is_one_launch_and_equals_the_unfused_chain(dmx, cuda, oracle, wdt, tag, ctor, shape, nm):
    """modeling/nn/core.py:178-198 for Conv1d / Conv2d (weight cast, sparsifier and SmoothQuant all along dim 1,
    torch_modules.py:582-585, 674-677): sparsifier -> smoothquant.scale_weight -> weight_cast in ONE launch, bit-identical to the
    three-op chain of this library (which the LeNet / model-shape fixtures pin to the reference), and to the ORACLE composed the way
    the reference composes the steps -- for {mask, no mask} x {scale, no scale} x symmetric / asymmetric formats, ragged last
    blocks (80 = 64 + 16 in-channels, 6 < 64) included."""
    C = shape[1]
    for fmt in ("BFP[8|8]{64}(SN)", "BFP[8|8]{16}(_N)", "BFP[4|8]{2}(SN)"):
        B = int(fmt.split("{")[1].split("}")[0])
        for use_mask in (False, True):
            if use_mask and B % int(nm.split(":")[1].split(",")[0]) != 0:
                continue
            for use_sq in (False, True):
                if not use_mask and not use_sq:
                    continue   # a plain cast is already one launch (and `_fused_weight` says so by returning None)
                m = ctor(dmx.nn).to(wdt)
                w = (make("heavy", shape, seed=31).clamp(-50, 50) * 0.1).to(wdt)
                m.weight.data = w.clone()
                m = m.to(cuda).eval()
                m.configure(dict(weight_format=fmt))
                score = None
                if use_mask:
                    m.configure(dict(weight_sparseness=nm))
                    m.fuse_weight_hypernet = False
                    with torch.no_grad():
                        m._weight                                      # materialise the lazy score
                    score = make("normal", shape, seed=32).abs()
                    score.reshape(-1)[::7] = score.reshape(-1)[1::7][: score.reshape(-1)[::7].numel()]   # ties inside groups
                    m.weight_sparsifier.score.data = score.to(cuda)
                scale = None
                if use_sq:
                    scale = (torch.rand(C, generator=torch.Generator().manual_seed(1)) * 4 + 0.25)
                    m.smoothquant.scale = scale.to(cuda)
                    m.smoothquant.enable()
                with torch.no_grad():
                    m.fuse_weight_hypernet = False
                    chain = m._weight
                    m.fuse_weight_hypernet = True
                    assert m._fused_weight(m.weight) is not None, (tag, fmt, use_mask, use_sq)
                    fused = m._weight
                assert fused.dtype == chain.dtype and bits_equal(fused, chain) == 0, (tag, fmt, use_mask, use_sq)
                # the oracle, composed like the reference: x * mask (promoted dtype) -> (x * scale).to(dtype) -> BFP cast -> .to(dtype)
                f = dmx.Format.from_shorthand(fmt)
                t = w
                if use_mask:
                    K, M = int(nm.split("{")[1].split(":")[0]), int(nm.split(":")[1].split(",")[0])
                    t = oracle.sparsify(t, score, K, M, 1)
                if use_sq:
                    t = (t * scale.view(1, -1, *([1] * (len(shape) - 2)))).to(t.dtype)
                want = oracle.bfp_cast(t, f.precision, f.block_size, 1, f.symmetric).to(t.dtype)
                assert want.dtype == fused.dtype and bits_equal(fused.cpu(), want.contiguous()) == 0, (tag, fmt, use_mask, use_sq, "oracle")
    # a plain BASIC conv (no mask, no scale): nothing to fuse, the chain is the single cast launch
    m = ctor(dmx.nn).to(cuda)
    m.configure(dict(weight_format="BFP[8|8]{64}(SN)"))
    with torch.no_grad():
        assert m._fused_weight(m.weight) is None and m._weight.shape == shape


def test_conv_forward_with_fused_weight_path(dmx, cuda):
    """the module's forward consumes the fused weight (same result as with the chain), Whisper conv2 shape"""
    m = dmx.nn.Conv1d(768, 768, 3, stride=2, padding=1).to(cuda).eval()
    dmx.configure_model(m, *dmx.config_rules.BASIC)
    m.configure(dict(weight_sparseness="BTOPK{4:8,1}(U)"))
    x = make("normal", (1, 768, 300), seed=5).to(cuda)
    with torch.no_grad():
        m.fuse_weight_hypernet = False
        m(x)                                                            # materialises the score
        y_chain = m(x)
        m.fuse_weight_hypernet = True
        assert m._fused_weight(m.weight) is not None
        y_fused = m(x)
    assert torch.equal(y_chain, y_fused)


def test_fold_refuses_an_uncalibrated_group_quantised_weight_like_the_per_module_path(dmx, cuda):
    """`fixed_qdq_multi` took any one-entry scale as per-tensor even with a group_size, so
    `fold_weights_and_biases` silently rounded an UNCALIBRATED group-quantised weight (scale = [1.0]) to integers; the per-module
    path raises ValueError (need N scale entries).  Both raise now; a calibrated model still folds, bit-identically."""
    hp = dmx.nn.DmxModuleQuantizerCalibrationHyperparams(weight=dmx.nn.DmxQuantizerCalibrationHyperparams(
        observer_cls=dmx.MinMaxObserver, qscheme_to_overload=torch.per_tensor_symmetric, group_size=128, ch_axis=0))

    def model():
        seq = torch.nn.Sequential(dmx.nn.Linear(256, 512), dmx.nn.Linear(512, 256)).to(cuda)
        for mod in seq:
            mod.configure(dict(weight_format="XP[8,0](CSN)"))
        return seq

    seq = model()
    for mod in seq:                                   # group_size set on the cast, but never calibrated: scale is still [1.0]
        mod.weight_cast.group_size, mod.weight_cast.ch_axis = 128, 0
    w0 = [mod.weight.detach().clone() for mod in seq]
    with pytest.raises(ValueError):
        with torch.no_grad():
            seq[0]._weight                            # the per-module path
    with pytest.raises(ValueError):
        dmx.nn.fold_weights_and_biases(seq)           # ... and the batched fold
    assert all(torch.equal(mod.weight.detach(), w) for mod, w in zip(seq, w0)), "a refused fold must leave the weights alone"
    with pytest.raises(ValueError):
        dmx.ops.fixed_qdq_multi([w0[0]], 8, 0, True, True, [torch.ones(1, device=cuda)], [torch.zeros(1, dtype=torch.int64, device=cuda)], group_size=128)
    # one group covering every channel is legal with one entry (C <= group_size), and so is a per-tensor scale without a group_size
    small = torch.randn(64, 32, device=cuda)
    a = dmx.ops.fixed_qdq_multi([small], 8, 0, True, True, [torch.full((1,), 0.05, device=cuda)], [torch.zeros(1, dtype=torch.int64, device=cuda)], group_size=128)[0]
    b = dmx.ops.fixed_qdq_multi([small], 8, 0, True, True, [torch.full((1,), 0.05, device=cuda)], [torch.zeros(1, dtype=torch.int64, device=cuda)])[0]
    c = dmx.ops.fixed_qdq(small, 8, 0, True, True, scale=torch.full((1,), 0.05, device=cuda), zero_point=torch.zeros(1, dtype=torch.int64, device=cuda))
    assert torch.equal(a, c) and torch.equal(b, c)
    # calibrated: folds, and equals module-by-module folding
    seq, ref = model(), model()
    for s_mod, r_mod in zip(seq, ref):
        r_mod.load_state_dict(s_mod.state_dict())
        for mod in (s_mod, r_mod):
            with mod.calibrating_quantizers(hp), torch.no_grad():
                mod._weight
    dmx.nn.fold_weights_and_biases(seq)
    for mod in ref:
        mod.fold_weight_and_bias()
    assert all(torch.equal(a.weight, b.weight) for a, b in zip(seq, ref))


def test_pending_foreign_hip_error_is_reported_not_swallowed(dmx, cuda):
    """with another HIP user's error pending on the thread, a call used to return DMXQ_OK without verifying its
    launches.  Now: DMXQ_ERR_PENDING (4), the foreign error left in place for its owner."""
    from dmx_compressor_amd import _lib
    L = _lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    x = torch.randn(64, 64, device=cuda, dtype=BF16)
    y = torch.empty_like(x)
    args = (ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), _lib.BF16, _lib.BF16, 64, 64, 1, 16, 8, 2, 1, 0, None)
    torch.cuda.synchronize()
    assert L.dmxq_bfp_qdq(*args) == 0
    hip.hipGetLastError()                                            # clean slate
    assert hip.hipMemcpy(None, None, 16, 99) != 0                    # somebody else's failing call: sets the thread's last error
    assert hip.hipPeekAtLastError() != 0
    rc = L.dmxq_bfp_qdq(*args)
    assert rc == _lib.ERR_PENDING == 4 and b"pre-existing" in L.dmxq_status_string(rc)
    assert hip.hipPeekAtLastError() != 0                              # still there for its owner
    hip.hipGetLastError()                                             # ... who collects it
    assert L.dmxq_bfp_qdq(*args) == 0
    torch.cuda.synchronize()


def test_graphed_forward_replays_a_configured_layer(dmx, cuda):
    """nn.GraphedForward: an opt-125m-sized block under the BASIC rules captured once and replayed -- same bits as the eager forward,
    new inputs picked up through the static buffer, shape changes refused."""
    nn = dmx.nn

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ln, self.fc1, self.act, self.fc2, self.res = nn.LayerNorm(256), nn.Linear(256, 1024), nn.GELU(), nn.Linear(1024, 256), nn.ResAdd()

        def forward(self, x):
            return self.res(self.fc2(self.act(self.fc1(self.ln(x)))), x)

    torch.manual_seed(0)
    m = Block().to(cuda).eval()
    dmx.configure_model(m, *dmx.config_rules.BASIC)
    x1, x2 = make("normal", (4, 64, 256), seed=1).to(cuda), make("normal", (4, 64, 256), seed=2).to(cuda)
    with torch.no_grad():
        e1, e2 = m(x1), m(x2)
    g = nn.GraphedForward(m, x1)
    y1 = g(x1).clone()
    y2 = g(x2).clone()
    assert torch.equal(y1, e1) and torch.equal(y2, e2) and not torch.equal(y1, y2)
    with pytest.raises(ValueError):
        g(x1[:2])
    # calibration must precede capture: an enabled observer (in-place running min / max) is refused
    m.fc1.input_casts.input_cast.enable_observer()
    with pytest.raises(RuntimeError, match="observer"):
        nn.GraphedForward(m, x1)
    m.fc1.input_casts.input_cast.disable_observer()


def test_round3_entry_points_through_both_bindings(dmx, cuda):
    """the round-3 entry points (fused activation / normalisation modules, the strided weight chain) return identical bits through
    the torch extension and the ctypes binding -- the same C ABI behind both"""
    C, T = dmx.ops.front("ctypes"), dmx.ops.front("torch")
    f16, b16 = dmx.Format.from_shorthand("FP[1|5|10,15](FN)"), dmx.Format.from_shorthand("FP[1|8|7,127](FN)")
    x = (make("normal", (96, 512), seed=1) * 3).to(BF16).to(cuda)
    xf = (make("heavy", (64, 384), seed=2)).clamp(-1e4, 1e4).to(cuda)
    w = torch.randn(512, device=cuda).to(BF16)
    wf = torch.randn(384, device=cuda)
    conv = (make("normal", (32, 24, 3, 3), seed=3) * 0.1).to(cuda)
    score = make("normal", (32, 24, 3, 3), seed=4).abs().to(cuda)
    sq = (torch.rand(24, device=cuda) + 0.5)
    cases = [
        lambda o: o.unary_cast(x, "gelu", f16, f16), lambda o: o.unary_cast(x, "silu", f16, None), lambda o: o.unary_cast(xf, "quick_gelu", b16, f16),
        lambda o: o.unary_cast(xf, "exp", None, f16), lambda o: o.unary_cast(x, "gelu_tanh", None, None),
        lambda o: o.softmax_cast(x, -1, f16, f16), lambda o: o.softmax_cast(xf, -1, f16, f16, input_clamp=-2.0),
        lambda o: o.layernorm_cast(x, 512, w, w, 1e-5, f16, f16), lambda o: o.layernorm_cast(xf, 384, wf, None, 1e-5, f16, b16),
        lambda o: o.rmsnorm_cast(x, 512, w, 1e-6, f16, f16), lambda o: o.rmsnorm_cast(xf, 384, None, None, None, f16),
        lambda o: o.weight_hypernet(conv, 8, 16, True, score, 2, 4, sq, block_dim=1), lambda o: o.weight_hypernet(conv.to(BF16), 4, 8, False, None, 0, 0, sq, block_dim=1),
    ]
    for i, f in enumerate(cases):
        a, b = f(C), f(T)
        assert a is not None and b is not None, i
        assert a.dtype == b.dtype and a.shape == b.shape and bits_equal(a, b) == 0, f"case {i}"
    # not fusable -> None from both (a rounding cast on a 16-bit tensor; softmax over another dim)
    e4m3 = dmx.Format.from_shorthand("FP[1|4|3,7](_N)")
    for o in (C, T):
        assert o.unary_cast(x, "gelu", e4m3, f16) is None and o.softmax_cast(x, 0, f16, f16) is None


def test_remaining_module_types_are_the_torch_op_between_the_casts(dmx, cuda):
    """ReLU6 / Tanh / Dropout / GELU variants / AdaptiveAvgPool2d / BatchNorm2d / GroupNorm / ConvTranspose2d / BAddBMM under the BASIC rules ==
    input cast(s) -> torch's op -> output cast, composed by hand from this library's cast ops (which the fixtures pin to the reference);
    ScaledDotProductAttention (compound) against the same composition of its submodules (modeling/nn/torch_modules.py:108-192)."""
    nn, ops, F = dmx.nn, dmx.ops, torch.nn.functional
    f16 = dmx.format.FLOAT16
    cast = lambda t: f16.cast(t, out_dtype=t.dtype)
    bfp = lambda t, dim=-1: ops.bfp_qdq(t, 8, 64, dim, True, out_dtype=t.dtype)
    x = (make("normal", (4, 16, 12, 12), seed=1) * 3).to(cuda)
    with torch.no_grad():
        for m, f in ((nn.ReLU6(), F.relu6), (nn.Tanh(), torch.tanh), (nn.Dropout(0.3).eval(), lambda t: t), (nn.NewGELU(), nn.NewGELU._f),
                     (nn.FastGELU(), nn.FastGELU._f), (nn.BloomGELU(), nn.BloomGELU._f), (nn.ClippedGELU(-1.0, 2.0), lambda t: torch.clip(F.gelu(t), -1.0, 2.0)),
                     (nn.AdaptiveAvgPool2d(3), lambda t: F.adaptive_avg_pool2d(t, 3))):
            m = m.to(cuda)
            dmx.configure_model(m, *dmx.config_rules.BASIC)
            want = f(cast(x)) if isinstance(m, nn.Dropout) else cast(f(cast(x)))   # (Dropout has no rule in the reference: SAME)
            if isinstance(m, nn.Dropout):
                want = x.clone()
            assert torch.equal(m(x), want), type(m).__name__
        bn = nn.BatchNorm2d(16).to(cuda).eval()
        bn.running_mean.copy_(torch.randn(16)); bn.running_var.copy_(torch.rand(16) + 0.5); bn.weight.data.normal_(1, 0.1); bn.bias.data.normal_(0, 0.1)
        dmx.configure_model(bn, *dmx.config_rules.BASIC)
        assert torch.equal(bn(x), cast(F.batch_norm(cast(x), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.1, bn.eps)))
        gn = nn.GroupNorm(4, 16).to(cuda)
        gn.weight.data.normal_(1, 0.1); gn.bias.data.normal_(0, 0.1)
        dmx.configure_model(gn, *dmx.config_rules.BASIC)
        assert torch.equal(gn(x), cast(F.group_norm(cast(x), 4, gn.weight, gn.bias, gn.eps)))
        ct = nn.ConvTranspose2d(16, 8, 3, stride=2).to(cuda)
        dmx.configure_model(ct, *dmx.config_rules.BASIC)
        w = bfp(ct.weight.detach(), 1)
        b = ops.float_qdq(ct.bias.detach(), 22, 8, 127, False)            # BFP32_1 = float_quantize(man = 22)
        want = cast(F.conv_transpose2d(bfp(x, 1), w, None, ct.stride, ct.padding, (0, 0), 1, ct.dilation) + b.unsqueeze(-1).unsqueeze(-1))
        assert torch.equal(ct(x), want)
        bb = nn.BAddBMM().to(cuda)
        i, b1, b2 = torch.randn(3, 5, 7, device=cuda), torch.randn(3, 5, 64, device=cuda), torch.randn(3, 64, 7, device=cuda)
        bb.configure(dict(input_formats=[dmx.format.SAME, dmx.format.BFP16_64, dmx.format.BFP16_64], output_formats=[f16]))
        assert torch.equal(bb(i, b1, b2), cast(torch.baddbmm(i, bfp(b1), bfp(b2, -2))))
        # compound attention: its five submodules take the rules; result == the same chain spelled out with the configured submodules
        sdpa = nn.ScaledDotProductAttention().to(cuda)
        dmx.configure_model(sdpa, *dmx.config_rules.BASIC)
        q, k, v = (torch.randn(2, 4, 64, 64, device=cuda) for _ in range(3))
        got = sdpa(q, k, v, is_causal=True)
        bias = torch.zeros(64, 64, device=cuda).masked_fill_(torch.ones(64, 64, dtype=torch.bool, device=cuda).tril().logical_not(), -10000.0)
        a = sdpa.actmatmul(q, k.transpose(-2, -1))
        a = sdpa.mul(sdpa.resadd(a, bias), torch.tensor(1 / 8.0, dtype=torch.float16, device=cuda))
        want = sdpa.actmatmul(sdpa.dropout(sdpa.softmax(a)), v)
        assert torch.equal(got, want) and got.shape == (2, 4, 64, 64)
        ref = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        assert (got - ref).abs().max() < 0.05 * ref.abs().max()              # (and it IS attention, up to the BASIC formats' precision)


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
def test_minmax_accumulate_equals_reduce_then_min_max(dmx, cuda, oracle, dtype):
    """dmxq_group_minmax_accumulate: the running min / max of a MinMaxObserver updated in ONE launch == group_minmax followed by
    torch.minimum / torch.maximum with the previous state (observer.py:173-193), exactly, over three batches; per group, per channel,
    per tensor; and the observer itself takes that path from its second observation on (same qparams as the oracle's)."""
    xs = [make("heavy", (256, 96), seed=40 + i, dtype=dtype).clamp(-1e4, 1e4) for i in range(3)]
    for ch_axis, gs in ((0, 32), (0, 1), (-1, 8), (0, 256)):
        G = -(-xs[0].shape[ch_axis] // gs)
        mn = torch.full((G,), float("inf"), device=cuda)
        mx = torch.full((G,), float("-inf"), device=cuda)
        rmn, rmx = mn.clone(), mx.clone()
        for x in xs:
            dmx.ops.group_minmax_accumulate(x.to(cuda), ch_axis, gs, mn, mx)
            a, b = dmx.ops.group_minmax(x.to(cuda), ch_axis, gs)
            rmn, rmx = torch.minimum(rmn, a), torch.maximum(rmx, b)
            assert torch.equal(mn, rmn) and torch.equal(mx, rmx), (ch_axis, gs)
        omn, omx = oracle.group_minmax(torch.cat(xs, dim=1 if ch_axis == 0 else 0), ch_axis, gs)
        assert torch.equal(mn.cpu(), omn) and torch.equal(mx.cpu(), omx)
    for qs, gsz in ((torch.per_tensor_symmetric, None), (torch.per_channel_affine, None), (torch.per_tensor_affine, 64)):
        obs = dmx.MinMaxObserver(qscheme=qs, ch_axis=0).to(cuda)
        for x in xs:
            obs(x.to(cuda), gsz)
        allx = torch.cat(xs, dim=1)
        if gsz:
            omn, omx = oracle.group_minmax(allx, 0, gsz)
        elif qs == torch.per_channel_affine:
            omn, omx = oracle.group_minmax(allx, 0, 1)
        else:
            omn, omx = oracle.group_minmax(allx.reshape(1, -1), 0, 1)
        assert torch.equal(obs.min_val.reshape(-1).cpu(), omn) and torch.equal(obs.max_val.reshape(-1).cpu(), omx), (qs, gsz)
    with pytest.raises(RuntimeError):
        dmx.ops.group_minmax_accumulate(xs[0].to(cuda), 0, 32, torch.zeros(3, device=cuda), torch.zeros(3, device=cuda))


def test_all_same_module_returning_a_view_is_cloned_under_torch_compile(dmx, cuda):
    """in a compiled graph storages have no addresses, and the boundary check used to see only `out is input`; an
    all-SAME module whose _forward returns a VIEW of its input escaped un-cloned, so a caller's in-place op on the result wrote
    through to the input (the reference's Same.cast always clones).  The view relation is followed now."""
    class Slice(dmx.nn.DmxModule):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self._dmx_init()

        def _forward(self, _input):
            return _input[:, :8]

    m = Slice().to(cuda)

    def f(x):
        y = m(x)
        y.add_(1.0)          # must not reach x
        return y

    x = torch.zeros(4, 16, device=cuda)
    eager = f(x)
    assert float(x.abs().max()) == 0.0 and float(eager.min()) == 1.0
    x2 = torch.zeros(4, 16, device=cuda)
    out = torch.compile(f, backend="aot_eager", fullgraph=True)(x2)
    assert float(x2.abs().max()) == 0.0 and float(out.min()) == 1.0


def test_smoothquant_calibration_and_dynamic_forwards_do_not_synchronise(dmx, cuda):
    """The calibration / dynamic SmoothQuant forward (smoothquant.py:518-535: both channel maxima and the scale, every call) reads its
    two scalar buffers (`migration_strength`, `scale_min`) from host mirrors: no device->host copy, so the forward neither drains the
    stream nor breaks a hipGraph capture.  The values used are the buffers' (also after set_migration_strength / load_state_dict)."""
    torch.manual_seed(0)
    m = dmx.nn.Linear(256, 128).to(cuda).eval()
    m.configure(dict(input_formats=["BFP[8|8]{64}(SN)"], weight_format="BFP[8|8]{64}(SN)"))
    x = torch.randn(32, 256, device=cuda)
    m.smoothquant.set_migration_strength(0.8)
    m.smoothquant.set_dynamic(True)
    m.smoothquant.enable()
    with torch.no_grad():
        m(x)                                             # warm-up: allocations, kernel loads
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            y = m(x)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    a, w = x.abs().amax(0), m.weight.detach().abs().amax(0)
    ref = (a.pow(torch.tensor(0.8, device=cuda)) / w.pow(1 - torch.tensor(0.8, device=cuda))).clamp(min=1e-5)
    assert torch.allclose(m.smoothquant.scale, ref, rtol=1e-6, atol=0.0)
    assert torch.isfinite(y).all()
    # a buffer poked directly (or loaded from a state dict: `_load_from_state_dict` calls refresh_flags) is picked up by refresh_flags
    m.smoothquant.migration_strength.fill_(0.3)
    m.smoothquant.refresh_flags()
    assert m.smoothquant._scalar("migration_strength") == float(torch.tensor(0.3, dtype=torch.float32))


def test_keep_layout_casts_a_permuted_view_without_a_copy(dmx, cuda):
    """CastTo.keep_layout (set on ActActMatMul's input casts): a dense transposed / permuted view is cast through the permutation that makes
    it contiguous, the block dimension following it -- same values as casting a contiguous copy, the input's strides on the result; views
    that are not dense permutations (slices) take the copying path."""
    torch.manual_seed(2)
    base = torch.randn(2, 96, 4, 64, device=cuda)
    views = {"[B,S,H,D] -> [B,H,S,D]": base.transpose(1, 2), "... -> k^T [B,H,D,S]": base.transpose(1, 2).transpose(-1, -2),
             "permute(3,0,2,1)": base.permute(3, 0, 2, 1), "sliced (not dense)": base.transpose(1, 2)[:, :, ::2]}
    for fmt in ("BFP[8|8]{64}(SN)", "BFP[8|8]{16}(_N)", "FP[1|5|10,15](FN)"):
        for bd in (-1, -2):
            c = dmx.CastTo(format=fmt, block_dim=bd).to(cuda)
            for name, v in views.items():
                for dtype in (torch.float32, torch.bfloat16):
                    x = v.to(dtype) if dtype != torch.float32 else v
                    if dtype != torch.float32 and "sliced" not in name:   # `.to` of a dense view keeps its strides
                        assert x.stride() == v.stride()
                    c.keep_layout = False
                    want = c(x)
                    c.keep_layout = True
                    got = c(x)
                    assert torch.equal(got, want), (fmt, bd, name, dtype)
                    if "sliced" not in name and not x.is_contiguous():
                        assert got.stride() == x.stride(), (name, got.stride(), x.stride())
    mm = dmx.nn.ActActMatMul().to(cuda).eval()
    assert all(c.keep_layout for c in mm.input_casts.values())
