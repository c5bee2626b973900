"""-m gpu, round 6: exact aliasing (out == in) of the kernels with a deferred exact redo, on inputs that TAKE the redo
(csrc/stream.hpp, csrc/lastdim.hpp redid flagged vectors from a fresh load AFTER the store burst, i.e. from their own
results when out == in)."""
import ctypes

import pytest
import torch

from _data import bits_equal, make, mismatches_nan_aware

pytestmark = pytest.mark.gpu
BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32
vp = ctypes.c_void_p


def _specials(x, seed, every=97):
    """NaN / +-Inf / +-1e38-class values (the largest finite ones of the dtype for fp16) sprinkled over a copy of x: quotients
    that the reciprocal forms do not cover (Inf / NaN / overflow) land in many different vectors of many tiles"""
    t = x.clone().reshape(-1)
    big = 6.0e4 if x.dtype == F16 else 1.0e38
    vals = torch.tensor([float("nan"), float("inf"), float("-inf"), big, -big, 3.0e38 if x.dtype != F16 else 65504.0], dtype=torch.float32)
    idx = torch.arange(seed % every, t.numel(), every)
    t[idx] = vals[torch.arange(idx.numel()) % vals.numel()].to(x.dtype)
    return t.reshape(x.shape)


@pytest.mark.parametrize("dtype", [BF16, F16, F32])
@pytest.mark.parametrize("shape,ch_axis,gs", [
    ((64, 1024), None, None),        # per-tensor scale, 256 x 1 tiles
    ((2048, 1024), None, None),      # 256 x 2 / 256 x 4 tiles
    ((4096, 4096), 0, 128),          # the op's own deep tiles (INT8 group-128: configs[2]'s cast), one group per tile
    ((1024, 1024), 0, 64),           # group runs shorter than a deep tile
    ((512, 768), 0, None),           # per-channel rows (ch_axis = 0): group walker
    ((512, 4096), -1, None),         # per-channel along the contiguous dim: lastdim_kernel
])
def test_fixed_qdq_in_place_with_special_values(dmx, cuda, oracle, dtype, shape, ch_axis, gs):
    """dmxq_fixed_qdq with out == in must give what it gives out of place, and what the oracle gives, on tensors whose NaN / Inf /
    huge elements send vectors through the kernels' exact redo"""
    from dmx_compressor_amd import _lib
    L = _lib.lib()
    s = vp(torch.cuda.current_stream().cuda_stream)
    x = _specials(make("heavy", shape, seed=61, dtype=dtype).clamp(-1e4, 1e4), seed=shape[0])
    C = 1 if ch_axis is None else shape[ch_axis]
    n_sc = 1 if ch_axis is None else -(-C // (gs or 1))
    sc = (make("normal", (n_sc,), seed=62).abs() * 0.05 + 1e-3).float()
    if n_sc > 3:
        sc[1] = 1e-30           # a scale outside the reciprocal form's range: the tile takes the general body
        sc[2] = 3.0e38
    for zero_points in (torch.zeros(n_sc, dtype=torch.int64), (torch.arange(n_sc) % 7 - 3).to(torch.int64)):
        sym = bool((zero_points == 0).all())
        want = oracle.fixed_point_affine_cast(x, 8, 0, True, sym, sc, zero_points, ch_axis=ch_axis, group_size=gs).to(dtype)
        outer, Cc, inner = (1, 1, x.numel()) if ch_axis is None else _lib.split3(x.shape, ch_axis)
        t, scd, zpd = x.to(cuda).contiguous(), sc.to(cuda), zero_points.to(cuda)
        o = torch.empty_like(t)
        code = _lib.dtype_code(dtype)
        args = (code, code, outer, Cc, inner, 8, 0, 1, int(sym), 2, vp(scd.data_ptr()), vp(zpd.data_ptr()), gs or 1, 0, s)
        assert L.dmxq_fixed_qdq(vp(t.data_ptr()), vp(o.data_ptr()), *args) == 0
        assert mismatches_nan_aware(o, want) == 0, ("out of place", shape, ch_axis, gs, sym)
        assert int(torch.isnan(o.float()).sum()) > 0      # (the specials really reach the result)
        assert L.dmxq_fixed_qdq(vp(t.data_ptr()), vp(t.data_ptr()), *args) == 0
        assert bits_equal(t, o) == 0, ("in place differs from out of place", shape, ch_axis, gs, sym)


@pytest.mark.parametrize("rows,L,B", [(512, 4096, 16), (300, 768, 64), (2048, 1024, 16)])
def test_input_hypernet_in_place_f32_with_special_values(dmx, cuda, rows, L, B):
    """dmxq_input_hypernet, float32 in == out (the only dtype pair that can alias): rows that fail the fast form (NaN / Inf / huge
    elements, a tiny and a huge SmoothQuant scale) are redone by the exact form -- from the ORIGINAL elements"""
    from dmx_compressor_amd import _lib
    Lb = _lib.lib()
    s = vp(torch.cuda.current_stream().cuda_stream)
    x = _specials(make("heavy", (rows, L), seed=71).clamp(-1e6, 1e6), seed=rows, every=211)
    sq = (make("normal", (L,), seed=72).abs() + 0.05).float()
    sq[3], sq[17] = 1e-30, 2.0e38
    t, sqd = x.to(cuda), sq.to(cuda)
    o = torch.empty_like(t)
    args = (vp(sqd.data_ptr()),)
    assert Lb.dmxq_input_hypernet(vp(t.data_ptr()), _lib.F32, vp(sqd.data_ptr()), vp(o.data_ptr()), _lib.F32, rows, L, B, 8, 1, s) == 0
    # what the fused kernel must equal (include/dmxq.h): the unfused pair, each of which is oracle-pinned elsewhere
    chain = dmx.ops.bfp_qdq(dmx.ops.scale_channels(t, sqd, -1, True), 8, B)
    assert mismatches_nan_aware(o, chain) == 0
    assert Lb.dmxq_input_hypernet(vp(t.data_ptr()), _lib.F32, vp(sqd.data_ptr()), vp(t.data_ptr()), _lib.F32, rows, L, B, 8, 1, s) == 0
    assert bits_equal(t, o) == 0
    del args


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_scale_channels_in_place_with_special_values(dmx, cuda, dtype):
    """dmxq_scale_channels(divide) in place along the contiguous dim and along rows"""
    from dmx_compressor_amd import _lib
    L = _lib.lib()
    s = vp(torch.cuda.current_stream().cuda_stream)
    x = _specials(make("heavy", (768, 3072), seed=81, dtype=dtype).clamp(-1e6, 1e6), seed=5)
    for ch_axis in (-1, 0):
        C = x.shape[ch_axis]
        sq = (make("normal", (C,), seed=82).abs() + 0.05).float()
        sq[1], sq[2] = 1e-30, 2.0e38
        t, sqd = x.to(cuda), sq.to(cuda)
        o = torch.empty_like(t)
        outer, Cc, inner = _lib.split3(t.shape, ch_axis)
        code = _lib.dtype_code(dtype)
        for divide in (1, 0):
            t.copy_(x)
            assert L.dmxq_scale_channels(vp(t.data_ptr()), vp(o.data_ptr()), code, code, outer, Cc, inner, vp(sqd.data_ptr()), divide, s) == 0
            shp = [1, 1]
            shp[ch_axis] = C
            xf, sf = x.float(), sq.reshape(shp)
            want = (xf / sf if divide else xf * sf).to(dtype)
            assert mismatches_nan_aware(o, want) == 0, (ch_axis, divide)
            assert L.dmxq_scale_channels(vp(t.data_ptr()), vp(t.data_ptr()), code, code, outer, Cc, inner, vp(sqd.data_ptr()), divide, s) == 0
            assert bits_equal(t, o) == 0, (ch_axis, divide)


def test_sparsify_has_the_mask_of_its_initial_score(dmx, cuda, oracle):
    """sparse.py:260-262: the reference's constructor computes the first mask; the mirror computes it when first read (the kernels run
    on the GPU only), on the score's own device, and a forward replaces it"""
    torch.manual_seed(3)
    sp = dmx.sparse.Sparsify((16, 64), sparseness="BTOPK{2:4,-1}(U)")
    m = sp.mask
    assert m is not None and m.device == sp.score.device and bits_equal(m, oracle.nm_mask(sp.score.detach(), 2, 4)) == 0
    assert sp.mask is m                                   # computed once
    sp = dmx.sparse.Sparsify((16, 64), sparseness="BTOPK{4:8,-1}(U)").to(cuda)
    assert bits_equal(sp.mask, oracle.nm_mask(sp.score.detach().cpu(), 4, 8)) == 0
    x = make("normal", (16, 64), seed=9).to(cuda)
    y = sp(x)
    assert bits_equal(y, oracle.sparsify(x.cpu(), sp.score.detach().cpu(), 4, 8)) == 0 and sp.mask is not None
    assert dmx.sparse.Sparsify((4, 4)).mask is None       # DENSE: nothing to mask


# ------------------------------------------------------------------------------------------------ bfp_slab.hip (LDS slabs / column tiles)
SLAB_CASES = [
    # shape, block_dim, B                      what it selects in csrc/bfp_slab.hip
    ((4, 128, 14, 14), 1, 64),               # FLAT: rows of 392 bytes (not whole vectors), one contiguous slab per tile
    ((2, 192, 14, 14), 1, 128),              # FLAT, 32 rows per lane, ragged last block (192 = 128 + 64)
    ((3, 100, 14, 14), 1, 64),               # FLAT with a ragged last block of 36 rows
    ((2, 128, 28, 28), 1, 64),               # SEGMENTED: two column tiles of 392
    ((2, 64, 56, 56), 1, 64),                # SEGMENTED: eight column tiles
    ((2, 48, 28, 28), 1, 16),                # B = 16: two lanes per column pair
    ((2, 40, 28, 28), 1, 8),                 # B = 8: one lane per column pair
    ((2, 96, 28, 28), 1, 32),
    ((1, 512, 18, 18), 1, 256),              # B = 256: 64 rows per lane
    ((2, 3, 224, 224), 1, 64),               # L = 3 < B / 2: one ragged block of three rows, B reduced to 8
    ((2, 64, 10, 10), 1, 64),                # inner = 100
    ((2, 64, 15, 15), 1, 64),                # odd inner: NOT the slab kernel (column kernel, unaligned form) -- same results
    ((2, 12, 200, 64), -2, 64),              # attention operands blocked along the sequence: inner = 64
    ((8, 1500, 64), -2, 64),                 # ... with a ragged last block (1500 = 23 x 64 + 28)
]


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("shape,dim,B", SLAB_CASES)
def test_bfp_slab_kernel_vs_oracle(dmx, cuda, oracle, dtype, shape, dim, B):
    """blocks along a strided dim through the LDS slab kernel: heavy-tailed data with special blocks (all zero, denormal maximum, Inf,
    NaN, huge maximum: the literal path), symmetric and asymmetric, two precisions (single / double rounding builds), in place"""
    x = make("heavy", shape, seed=B + shape[1], dtype=dtype)
    flat = x.reshape(-1)
    n = flat.numel()
    xs = x.clone()
    view = xs.transpose(dim, -1)
    view[0, ..., : min(B, view.shape[-1])] = 0.0                    # an all-zero block per column of the first outer index
    flat = xs.reshape(-1)
    flat[n // 3] = float("inf")
    flat[n // 3 + 7] = float("nan")
    flat[n // 2] = 3.0e38 if dtype == BF16 else 65504.0
    flat[n // 2 + 1] = -(3.0e38 if dtype == BF16 else 65504.0)
    flat[5::997] = flat[5::997] * 0 + (1e-40 if dtype == BF16 else 6e-8)     # denormal elements
    for wl, sym in ((8, True), (8, False), (16, True), (4, False)):
        want = oracle.bfp_cast(xs, wl, B, dim, sym).to(dtype)
        got = dmx.ops.bfp_qdq(xs.to(cuda), wl, B, dim, sym)
        assert mismatches_nan_aware(got, want) == 0, (shape, dim, B, wl, sym)
    # in place through the C ABI
    from dmx_compressor_amd import _lib
    L = _lib.lib()
    t = xs.to(cuda).contiguous()
    outer, Ld, inner = _lib.split3(t.shape, dim)
    code = _lib.dtype_code(dtype)
    assert L.dmxq_bfp_qdq(vp(t.data_ptr()), vp(t.data_ptr()), code, code, outer, Ld, inner, B, 8, 2, 1, 0, vp(torch.cuda.current_stream().cuda_stream)) == 0
    assert mismatches_nan_aware(t, oracle.bfp_cast(xs, 8, B, dim, True).to(dtype)) == 0


# ------------------------------------------------------------------------------------------------ a calibrating forward as a hipGraph
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_graphed_forward_of_a_minmax_calibrating_linear(dmx, cuda, dtype):
    """GraphedForward(calibrating=True): every replay is one MinMax observer step (numerical/cast.py:179-226) on the batch it was given.
    After the same batches, the observer state and the qparams of the graphed copy equal the eagerly calibrated copy's bit for bit
    (input cast: per tensor; weight cast: slabs of 128 rows), and so do the fake-quantised outputs afterwards."""
    nn = dmx.nn

    def make_linear():
        m = nn.Linear(768, 3072, bias=True)
        m.weight.data = (make("normal", (3072, 768), seed=31) * 0.05).to(dtype)
        m.bias.data = (make("normal", (3072,), seed=32) * 0.02).to(dtype)
        m = m.to(cuda).eval()
        m.configure(dict(input_formats=[dmx.format.INT8], weight_format=dmx.format.INT8))
        return m

    hp = nn.DmxModuleQuantizerCalibrationHyperparams(
        inputs={"input_cast": nn.DmxQuantizerCalibrationHyperparams(observer_cls=dmx.MinMaxObserver, qscheme_to_overload=torch.per_tensor_affine)},
        weight=nn.DmxQuantizerCalibrationHyperparams(observer_cls=dmx.MinMaxObserver, qscheme_to_overload=torch.per_tensor_symmetric,
                                                     group_size=128, ch_axis=0))
    batches = [(make("heavy", (4, 64, 768), seed=40 + i).clamp(-50, 50) * (1 + i)).to(dtype).to(cuda) for i in range(4)]
    eager, graphed = make_linear(), make_linear()
    with torch.no_grad():
        with eager.calibrating_quantizers(hp):
            for b in batches:
                eager(b)
        with pytest.raises(RuntimeError):
            graphed.enable_quantizer_calib(True, hp)
            nn.GraphedForward(graphed, batches[0])                # an enabled observer without calibrating=True: refused
        g = nn.GraphedForward(graphed, batches[0], calibrating=True, warmup=2)
        for b in batches:
            g(b)
        torch.cuda.synchronize()
        graphed.enable_quantizer_calib(False, hp)
    for ce, cg in ((eager.input_casts.input_cast, graphed.input_casts.input_cast), (eager.weight_cast, graphed.weight_cast)):
        oe, og = ce.activation_post_process, cg.activation_post_process
        assert oe.min_val.shape == og.min_val.shape
        assert bits_equal(oe.min_val, og.min_val) == 0 and bits_equal(oe.max_val, og.max_val) == 0
        assert bits_equal(ce.scale, cg.scale) == 0 and torch.equal(ce.zero_point, cg.zero_point)
        assert float(oe.max_val.max()) > 0
    x = batches[1]
    with torch.no_grad():
        assert bits_equal(eager(x), graphed(x)) == 0
        hist = nn.DmxModuleQuantizerCalibrationHyperparams(inputs={"input_cast": nn.DmxQuantizerCalibrationHyperparams(observer_cls=dmx.HistogramObserver)})
        graphed.enable_quantizer_calib(True, hist)
        with pytest.raises(RuntimeError, match="HistogramObserver"):
            nn.GraphedForward(graphed, x, calibrating=True)
        graphed.enable_quantizer_calib(False, hist)


# ------------------------------------------------------------------------------------------------ dispatcher-free entry points
def test_direct_entry_points_equal_the_dispatcher_ops(dmx, cuda):
    """csrc/torch_binding.cpp PyInit_dmxq_fast: the same C++ functions as `torch.ops.dmxq.*` without the dispatcher.  Every routed
    name gives the dispatcher op's bits, raises what it raises, and steps aside while torch.compile traces."""
    if dmx.ops.BINDING != "torch":
        pytest.skip("torch binding only")
    from dmx_compressor_amd import _backend_torch as B
    assert B.FAST is not None, "dmxq_torch.so was built without PyInit_dmxq_fast"
    x = make("heavy", (64, 1536), seed=91, dtype=BF16).clamp(-1e4, 1e4).to(cuda)
    xf = x.float()
    sc = (make("normal", (12,), seed=92).abs() * 0.05 + 1e-3).to(cuda)
    zp = torch.zeros(12, dtype=torch.int64, device=cuda)
    f16 = [10, 5, 15, 1]
    w = (make("normal", (1536,), seed=93) * 0.1 + 1).to(BF16).to(cuda)
    table = B.RAW.unary_cast_table(x, 0, 0.0, f16, f16)
    cases = {
        "bfp_qdq_nograd": (x, 8, 64, -1, True, 2, None, 0),
        "float_qdq_nograd": (x, 10, 5, 15, True, False, 2, None, 0),
        "fixed_qdq_nograd": (xf, 8, 0, True, True, 2, sc, zp, -1, 128, None, 0),
        "sbfp_qdq_nograd": (x, 4, 16, 4, 4, 7, True, True, True, -1, None),
        "mxfp_qdq_nograd": (x, 3, 4, 32, -1, None),
        "weight_hypernet": (x, 8, 64, True, torch.rand(64, 1536, device=cuda), 2, 4, None, BF16, -1),
        "input_hypernet": (x, (torch.rand(1536, device=cuda) + 0.5), 8, 64, True),
        "binary_cast": (x, x.flip(0), 0, f16, f16, f16, 0, 0),
        "relu_cast": (x, f16, f16, 0, 0),
        "scale_channels": (x, (torch.rand(1536, device=cuda) + 0.5), -1, True, None),
        "unary_cast": (x, 0, 0.0, f16, f16),
        "lut16_apply": (x, table),
        "softmax_cast": (x, float("-inf"), f16, f16, 0, 0),
        "norm_cast": (x, 1536, w, w, 1e-5, 0, f16, f16, 0, 0),
    }
    assert set(cases) | {"rope_cast"} == set(B._FAST_NAMES)
    for name, args in cases.items():
        routed = getattr(B.RAW, name)
        op = routed.__wrapped__
        got, want = routed(*args), op(*args)
        assert got.dtype == want.dtype and mismatches_nan_aware(got, want) == 0, name
    with pytest.raises(RuntimeError):
        B.RAW.bfp_qdq_nograd(x.cpu(), 8, 64, -1, True, 2, None, 0)            # no CPU fallback, same error class
    with pytest.raises(NotImplementedError):
        B.RAW.input_hypernet(x, torch.ones(7, device=cuda), 8, 64, True)      # not fusable: NotImplementedError as from the dispatcher
    # torch.compile traces the dispatcher op (meta kernel), not the extension function
    cast = dmx.CastTo(format="BFP[8|8]{64}(SN)")
    eager = cast(x)
    compiled = torch.compile(lambda t: cast(t) + 0, fullgraph=True)
    assert mismatches_nan_aware(compiled(x), eager + 0) == 0


def test_bench_line_carries_the_second_tier_and_the_build_stamp():
    """`python bench.py` at N = 1 times configs 3 / 4 / 5's kernels after the headline measurement and puts them into the ONE JSON line
    (`ops`: us / bytes / frac / check per op, each checked against the oracle outside the timed region), with the library's stamp
    (`build`).  Here with `--tier2-only basic` (one op) and no layers' worth of waiting."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--replays", "3", "--nbuf", "4", "--preheat", "50",
                        "--no-cpu-baseline", "--tier2-only", "basic"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    op = line["ops"]["basic.float16_activation_cast_bf16"]
    assert "bit-exact" in op["check"] and 0 < op["frac"] < 1 and op["bytes"] == 4 * 4096 * 4096 and op["us"] > 0
    assert set(line["layers"]) == {"opt125m", "llama", "whisper"} and all("us_per_forward" in v for v in line["layers"].values())
    assert line["build"]["libdmxq_sha256"] and "4/4 slots == oracle" in line["config"]["check"]
    assert line["roofline"]["frac"] > 0 and line["value"] > 0


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_fixed_float_multi_with_an_empty_list_is_the_two_calls_in_both_bindings(dmx, cuda, oracle, binding):
    """the torch binding refused an empty list where the ctypes one fell back to the single-op calls; the front end decides now"""
    F = dmx.ops.front(binding)
    w = (make("normal", (256, 64), seed=3) * 0.05).to(cuda)
    sc = (w.reshape(2, 128, 64).abs().amax(dim=(1, 2)) / 127.0).contiguous()
    zp = torch.zeros(2, dtype=torch.int64, device=cuda)
    b = (make("normal", (64,), seed=4) * 0.02).to(cuda)
    a, f = F.fixed_float_qdq_multi([w], 8, 0, True, True, [sc], [zp], 128, [], 22, 8, 127, False)
    assert f == [] and bits_equal(a[0], oracle.fixed_point_affine_cast(w.cpu(), 8, 0, True, True, sc.cpu(), zp.cpu(), ch_axis=0, group_size=128)) == 0
    a, f = F.fixed_float_qdq_multi([], 8, 0, True, True, [], [], 128, [b], 22, 8, 127, False)
    assert a == [] and bits_equal(f[0], oracle.float_quantize(b.cpu(), 22, 8, 127, False)) == 0
    assert F.fixed_float_qdq_multi([], 8, 0, True, True, [], [], 128, [], 22, 8, 127, False) == ([], [])


def test_graphed_module_captures_one_graph_per_input_signature(dmx, cuda):
    """nn.GraphedModule: the module's call syntax over one hipGraph per (shapes, dtypes): results equal the eager forward bit for bit,
    a second signature gets its own graph, results are clones (they survive the next call), `invalidate()` re-captures after a
    reconfiguration, the least recently used graph is dropped beyond `max_graphs`."""
    nn = dmx.nn

    class Stack(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.act = nn.Linear(256, 512), nn.Linear(512, 256), nn.GELU()

        def forward(self, x):
            return self.b(self.act(self.a(x)))

    m = Stack().to(cuda).to(BF16).eval()
    nn.configure_model(m, *dmx.config_rules.BASIC)
    x1 = make("normal", (8, 64, 256), seed=71, dtype=BF16).to(cuda)
    x2 = make("normal", (3, 256), seed=72, dtype=BF16).to(cuda)
    with torch.no_grad():
        e1, e2 = m(x1).clone(), m(x2).clone()
        fast = nn.GraphedModule(m, max_graphs=2)
        y1 = fast(x1)
        y2 = fast(x2)
        assert bits_equal(y1, e1) == 0 and bits_equal(y2, e2) == 0 and len(fast._graphs) == 2
        y1b = fast(x1 * 0.5)
        assert bits_equal(y1, e1) == 0                      # a clone: not overwritten by the replay
        assert bits_equal(y1b, m(x1 * 0.5)) == 0
        fast(make("normal", (5, 256), seed=73, dtype=BF16).to(cuda))
        assert len(fast._graphs) == 2 and (((3, 256), BF16, x2.device),) not in fast._graphs   # least recently used dropped
        m.a.configure(dict(weight_format="BFP[8|8]{16}(SN)"))
        fast.invalidate()
        assert bits_equal(fast(x1), m(x1)) == 0 and bits_equal(fast(x1), e1) != 0


@pytest.mark.parametrize("chunk", range(3))
def test_bfp_slab_random_cases(dmx, cuda, oracle, chunk):
    """randomised differential cases drawn INSIDE the slab kernel's routing rule (rows that are not whole 128-byte lines, slabs of at
    least 16 KiB): block sizes, ragged block dims, even inner extents, both 16-bit dtypes, precisions, symmetric / asymmetric"""
    import random
    rng = random.Random(6000 + chunk)
    done = 0
    while done < 14:
        B = rng.choice([8, 16, 32, 64, 64, 64, 128, 256])
        inner = 2 * rng.randrange(32, 620)
        if (inner * 2) % 128 == 0 or B * inner * 2 < 16 * 1024 or B * ((inner // 2 + 40)) * 4 > 150 * 1024:
            continue
        L = rng.choice([B, 2 * B, B + rng.randrange(1, B), 3 * B, max(1, B // 2 + 1), 5])
        outer = rng.choice([1, 2, 3])
        if outer * L * inner > 600000:
            continue
        dtype = rng.choice([BF16, F16])
        wl = rng.choice([2, 4, 8, 8, 8, 12, 15, 16, 20])
        sym = rng.random() < 0.7
        x = make(rng.choice(["normal", "heavy", "mixed_nd" if not sym else "mixed", "outlier", "ties"]), (outer, L, inner), seed=chunk * 100 + done, dtype=dtype, block=min(B, 64))
        got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, 1, sym)
        want = oracle.bfp_cast(x, wl, B, 1, sym).to(dtype)
        assert mismatches_nan_aware(got, want) == 0, (chunk, done, outer, L, inner, B, dtype, wl, sym)
        done += 1


def test_plans_scaled_to_another_cu_count_give_the_same_bits(oracle):
    """common.hpp plan_cus: the size classes are stated for 256 CUs and scaled by the device's count (DMXQ_PLAN_CUS overrides, read once
    per process).  Planning for 64 or 304 CUs changes tile geometry only: every result stays bit-exact against the oracle, on sizes
    that cross the plan boundaries (flat rows, per-group affine, multi-tensor sets)."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent("""
        import sys, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests"); sys.path.insert(0, %r + "/oracle")
        import dmx_compressor_amd as dmx, oracle as O
        from _data import make, bits_equal
        dev = torch.device("cuda:0")
        for rows in (96, 700, 1800, 2600, 3072, 4096, 5000):
            x = make("heavy", (rows, 1024), seed=rows, dtype=torch.bfloat16)
            assert bits_equal(dmx.ops.bfp_qdq(x.to(dev), 8, 16), O.bfp_cast(x, 8, 16).to(torch.bfloat16)) == 0, rows
            xf = x.float().clamp(-50, 50)
            sc = (torch.rand(-(-rows // 128)) * 0.05 + 0.01); zp = torch.zeros(sc.numel(), dtype=torch.int64)
            got = dmx.ops.fixed_qdq(xf.to(dev), 8, 0, True, True, scale=sc.to(dev), zero_point=zp.to(dev), ch_axis=0, group_size=128)
            assert bits_equal(got, O.fixed_point_affine_cast(xf, 8, 0, True, True, sc, zp, ch_axis=0, group_size=128)) == 0, rows
        ws = [make("normal", (r, 768), seed=r).to(torch.bfloat16) for r in (768, 3072, 768, 128)]
        outs = dmx.ops.bfp_qdq_multi([w.to(dev) for w in ws], 8, 64)
        assert all(bits_equal(o, O.bfp_cast(w, 8, 64).to(torch.bfloat16)) == 0 for o, w in zip(outs, ws))
        print("OK")
    """ % (root, root, root))
    for cus in ("64", "304"):
        env = dict(os.environ, DMXQ_PLAN_CUS=cus)
        p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "OK" in p.stdout, (cus, p.stderr[-1500:])


def test_init_gate_stress_tests_on_the_formal_fence_build_too():
    """the init gate of the reductions is built two ways -- the default (s_waitcnt + sc1 stores: an argument about this
    hardware) and -DDMXQ_GATE_FENCES=1 (release / acquire fences of the LLVM AMDGPU memory model).  The takeover, filler-kernel,
    multi-stream / capture and concurrent-host-thread tests of rounds 4-5 run in the main suite on the default build; here the SAME tests
    run against lib/libdmxq_gate_fences.so (build.py build_gate_fences_variant), loaded through the ctypes binding."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "dmx-compressor_amd", "lib", "libdmxq_gate_fences.so")
    assert os.path.exists(lib), "libdmxq_gate_fences.so missing: run __graft_entry__.build()"
    env = dict(os.environ, DMXQ_BINDING="ctypes", DMXQ_LIB_PATH=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_round5.py"), os.path.join(root, "tests", "test_gpu_round4.py"),
                        "-m", "gpu", "-q", "-x", "-k", "init_gate", "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert " passed" in p.stdout and "failed" not in p.stdout
