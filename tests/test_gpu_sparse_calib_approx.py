"""-m gpu parity tests: N:M mask (bit-exact), calibration reductions (exact), SmoothQuant scale and the
approximator-slot ops (floating point: tolerance stated per test)."""
import math

import pytest
import torch

from _data import bits_equal, make, mismatches_nan_aware

pytestmark = pytest.mark.gpu


def _scores(kind, shape, seed):
    if kind == "random":
        return make("normal", shape, seed)
    if kind == "tied":       # few distinct values -> many ties inside every group
        g = torch.Generator().manual_seed(seed)
        return torch.randint(0, 3, shape, generator=g).float()
    if kind == "equal":
        return torch.ones(shape)
    if kind == "zeros_signed":
        g = torch.Generator().manual_seed(seed)
        return torch.where(torch.rand(shape, generator=g) < 0.5, torch.tensor(0.0), torch.tensor(-0.0))
    if kind == "nan":
        s = make("normal", shape, seed)
        g = torch.Generator().manual_seed(seed)
        s[torch.rand(shape, generator=g) < 0.2] = float("nan")
        return s
    if kind == "absbf16":
        return make("normal", shape, seed, torch.bfloat16).abs()
    raise ValueError(kind)


@pytest.mark.parametrize("KM", [(2, 4), (4, 8), (2, 8), (1, 4), (3, 4), (1, 2), (5, 16), (3, 12), (8, 32)])
@pytest.mark.parametrize("kind", ["random", "tied", "equal", "zeros_signed", "nan", "absbf16"])
def test_nm_mask_bit_exact(dmx, cuda, oracle, KM, kind):
    K, M = KM
    s = _scores(kind, (64, 6 * M * 4), seed=K * 10 + M)
    got = dmx.ops.nm_mask(s.to(cuda), K, M)
    want = oracle.nm_mask(s, K, M)
    assert got.dtype == s.dtype and bits_equal(got, want) == 0
    assert torch.all(got.reshape(-1, M).float().sum(1) == K)                 # exactly K kept per group


@pytest.mark.parametrize("dim", [1, 0, -2])
def test_nm_mask_block_dim(dmx, cuda, oracle, dim):
    s = _scores("tied", (8, 16, 24), seed=dim + 5)
    assert bits_equal(dmx.ops.nm_mask(s.to(cuda), 4, 8, dim), oracle.nm_mask(s, 4, 8, dim).contiguous()) == 0


@pytest.mark.parametrize("xd,sd", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                   (torch.bfloat16, torch.bfloat16), (torch.float16, torch.float32)])
def test_nm_sparsify_fused_apply(dmx, cuda, oracle, xd, sd):
    x = make("normal", (128, 256), seed=1, dtype=xd)
    s = make("normal", (128, 256), seed=2, dtype=sd)
    got = dmx.ops.nm_sparsify(x.to(cuda), s.to(cuda), 2, 4)
    want = oracle.sparsify(x, s, 2, 4)                                       # torch promotion of x * mask
    assert got.dtype == want.dtype and bits_equal(got, want) == 0
    assert bool((got.cpu() == 0).any()) and bool(torch.signbit(got.cpu()[got.cpu() == 0]).any())  # -0.0 kept


def test_sparsify_module_semantics(dmx, cuda):
    """sparse.py:287-301: score_func result used once, then the stored (random) score; mask recomputed per call."""
    torch.manual_seed(0)
    w = make("normal", (32, 64), seed=3).to(cuda)
    sp = dmx.Sparsify(w.shape, sparseness="BTOPK{2:4,-1}(U)").to(cuda)
    sp.eval()
    y_rand = sp(w)
    assert torch.all((y_rand != 0).reshape(-1, 4).sum(1) <= 2)
    sp.configure(score_func=lambda score, x: x.abs())
    y_mag = sp(w)   # plastic: magnitude pruning, once
    keep = w.abs().reshape(-1, 4).argsort(dim=1, stable=True)[:, 2:]
    ref = torch.zeros_like(w).reshape(-1, 4).scatter(1, keep, w.reshape(-1, 4).gather(1, keep)).reshape(w.shape)
    assert torch.equal(y_mag, ref)
    assert torch.equal(sp(w), y_rand)  # falls back to the stored score
    with pytest.raises(AssertionError):
        dmx.ops.nm_mask(torch.zeros(4, 6, device=cuda), 2, 4)                # sparse.py:166-168


def test_sparsify_gradients(dmx, cuda):
    """Structural check like the reference's tests/test_sparse.py: which gradients exist, and dx = g * mask."""
    w = make("normal", (16, 32), seed=4).to(cuda).requires_grad_()
    sp = dmx.Sparsify(w.shape, sparseness="BTOPK{4:8,-1}(U)", backward_mode="STE").to(cuda).train()
    y = sp(w)
    y.sum().backward()
    assert w.grad is not None and torch.equal(w.grad, sp.mask) and sp.score.grad is None
    sp2 = dmx.Sparsify(w.shape, sparseness="BTOPK{4:8,-1}(U)", backward_mode="joint").to(cuda).train()
    w.grad = None
    sp2(w).sum().backward()
    assert w.grad is not None and sp2.score.grad is not None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_group_minmax_and_qparams(dmx, cuda, oracle, dtype):
    x = make("heavy", (13, 50, 7), seed=6, dtype=dtype)
    for ax in (0, 1, 2):
        for gs in (1, 2, 5, x.shape[ax], 64):
            mn, mx = dmx.ops.group_minmax(x.to(cuda), ax, gs)
            omn, omx = oracle.group_minmax(x, ax, gs)
            assert torch.equal(mn.cpu(), omn) and torch.equal(mx.cpu(), omx), (ax, gs)
            for fmt_sym in (True, False):
                for q_sym in (True, False):
                    qmin = -(2 ** 7) + (1 if fmt_sym else 0)
                    sc, zp = dmx.ops.qparams(mn, mx, qmin, 127, q_sym)
                    osc, ozp = oracle.qparams(omn, omx, 8, fmt_sym, q_sym)
                    assert bits_equal(sc, osc) == 0 and torch.equal(zp.cpu(), ozp)


def test_group_minmax_large_single_group(dmx, cuda):
    x = make("normal", (2048, 4096), seed=7, dtype=torch.bfloat16).to(cuda)
    mn, mx = dmx.ops.group_minmax(x.reshape(1, -1), 0, 1)
    assert float(mn) == float(x.min()) and float(mx) == float(x.max())
    mn, mx = dmx.ops.group_minmax(x, 0, 128)
    xr = x.float().reshape(16, -1)
    assert torch.equal(mn, xr.amin(1)) and torch.equal(mx, xr.amax(1))


def test_castto_calibration_flow_group_quant(dmx, cuda, oracle):
    """numerical/cast.py:308-340 + 179-226: calibrate (observe only) then fake-quantise with per-group scales.
    Pinned against the oracle: scale[g] = amax|W[g-th slab]| / 127, zp = 0, output = fused affine INT8."""
    W = make("normal", (48, 40), seed=8)
    c = dmx.CastTo(format=dmx.format.INT8, ch_axis=0)
    c.enable_calibration(True, dmx.MinMaxObserver, torch.per_tensor_symmetric, group_size=16, ch_axis=0)
    out = c(W.to(cuda))
    assert torch.equal(out.cpu(), W)                                         # observe-only pass is the identity
    c.enable_calibration(False)
    omn, omx = oracle.group_minmax(W, 0, 16)
    osc, ozp = oracle.qparams(omn, omx, 8, True, True)
    assert bits_equal(c.scale, osc) == 0 and torch.equal(c.zero_point.cpu(), ozp)
    got = c(W.to(cuda))
    want = oracle.fixed_point_affine_cast(W, 8, 0, True, True, osc, ozp, ch_axis=0, group_size=16)
    assert bits_equal(got, want) == 0
    # equivalences the reference tests (tests/test_group_quant.py:144-369): g = C <=> per-tensor, g = 1 <=> per-channel
    c1 = dmx.CastTo(format=dmx.format.INT8, ch_axis=0)
    c1.enable_calibration(True, dmx.MinMaxObserver, torch.per_tensor_symmetric, group_size=48, ch_axis=0)
    c1(W.to(cuda)); c1.enable_calibration(False)
    c2 = dmx.CastTo(format=dmx.format.INT8)
    c2.enable_calibration(True, dmx.MinMaxObserver, torch.per_tensor_symmetric)
    c2(W.to(cuda)); c2.enable_calibration(False)
    assert bits_equal(c1(W.to(cuda)), c2(W.to(cuda))) == 0
    c3 = dmx.CastTo(format=dmx.format.INT8, ch_axis=0)
    c3.enable_calibration(True, dmx.MinMaxObserver, torch.per_tensor_symmetric, group_size=1, ch_axis=0)
    c3(W.to(cuda)); c3.enable_calibration(False)
    c4 = dmx.CastTo(format=dmx.format.INT8, ch_axis=0)
    c4.enable_calibration(True, dmx.MinMaxObserver, torch.per_channel_symmetric, ch_axis=0)
    c4(W.to(cuda)); c4.enable_calibration(False)
    assert bits_equal(c3(W.to(cuda)), c4(W.to(cuda))) == 0


def test_channel_maxabs_and_smoothquant_scale(dmx, cuda, oracle):
    a = make("heavy", (4, 33, 96), seed=9, dtype=torch.bfloat16)
    w = make("normal", (80, 96), seed=10)
    am = dmx.ops.channel_maxabs(a.to(cuda), -1)
    wm = dmx.ops.channel_maxabs(w.to(cuda), -1)
    assert torch.equal(am.cpu(), oracle.channel_maxabs(a, -1)) and torch.equal(wm.cpu(), oracle.channel_maxabs(w, -1))
    assert torch.equal(dmx.ops.channel_maxabs(w.to(cuda), 0).cpu(), w.abs().amax(1))
    for alpha in (0.0, 0.5, 0.25, 1.0):
        got = dmx.ops.smoothquant_scale(am, wm, alpha, 1e-5).cpu()
        b = wm.cpu().clamp(min=1e-5)
        # smoothquant.py:309-320; ground truth in float64, rounded once.  Two powf and a division: within 4 ulp of fp32
        # (measured 3-4; torch's own fp32 CPU evaluation measures 2-3 on the same inputs, profiles/r02_accuracy_table.txt)
        from _data import err_in_ulps
        truth = ((am.cpu().double() ** alpha) / (b.double() ** (1.0 - alpha))).clamp(min=1e-5)
        assert err_in_ulps(got, truth, torch.float32) <= 4.0, alpha
    z = dmx.ops.smoothquant_scale(torch.zeros(4, device=cuda), torch.zeros(4, device=cuda), 0.5, 1e-5)
    assert torch.all(z == 1e-5)                                              # clamp at scale_min


# Tolerances of the exact-function ops, in ulps of the OUTPUT format against the float64 truth rounded once
# (tests/_data.py err_in_ulps; measured maxima for the kernels AND for torch's own CPU result, which is what the reference
# evaluates, are in profiles/r02_accuracy_table.txt):
#   16-bit outputs: 1 ulp everywhere ("within 1 ULP of the stated format");
#   fp32 outputs: gelu 2 (erf / tanh + 3 products), softmax 8 (expf <= 1, a row sum of up to 16k fp32 terms, one division;
#   the argument x - max is compensated -- torch's CPU softmax measures 7-24 on the same inputs), layer_norm 3 (two row
#   reductions, rsqrt, fma).  gelu and layer_norm are counted against the magnitude of their CANCELLING terms.
TOL = {"gelu": {torch.float32: 2.0}, "softmax": {torch.float32: 8.0}, "layernorm": {torch.float32: 3.0}}


def _tol(op, dtype):
    return TOL[op].get(dtype, 1.0)


def _ln_truth(x, cols, w, b, eps=1e-5):
    F = torch.nn.functional
    xd = x.double()
    truth = F.layer_norm(xd, (cols,), None if w is None else w.double(), None if b is None else b.double(), eps)
    # cancelling terms of y = (x - mean) rstd w + b: the row mean is a sum of terms of the row's magnitude, so its rounding
    # error -- and with it the error of every (x - mean) -- is relative to max|x| of the ROW, not to the element
    mu, rstd = xd.mean(-1, keepdim=True), (xd.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
    floor = (xd.abs().amax(-1, keepdim=True) + mu.abs()) * rstd * (1.0 if w is None else w.double().abs()) + (0.0 if b is None else b.double().abs())
    return truth, floor


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gelu_softmax_layernorm_exact_function_parity(dmx, cuda, dtype):
    """Approximator slot = the exact torch.nn.functional function (vsimd absent).  Ground truth: float64, rounded once."""
    F = torch.nn.functional
    from _data import err_in_ulps
    x = (make("normal", (64, 1500), seed=11) * 3).to(dtype)
    xd = x.double()
    assert err_in_ulps(dmx.ops.gelu(x.to(cuda)), F.gelu(xd), dtype, floor=xd.abs() / 2) <= _tol("gelu", dtype)
    assert err_in_ulps(dmx.ops.gelu(x.to(cuda), "tanh"), F.gelu(xd, approximate="tanh"), dtype, floor=xd.abs() / 2) <= _tol("gelu", dtype)
    assert err_in_ulps(dmx.ops.softmax(x.to(cuda), -1), F.softmax(xd, -1), dtype) <= _tol("softmax", dtype)
    assert err_in_ulps(dmx.ops.softmax(x.to(cuda), -1, input_clamp=-1.0), F.softmax(xd.clamp(min=-1.0), -1), dtype) <= _tol("softmax", dtype)
    x3 = x.reshape(8, 8, 1500)
    assert err_in_ulps(dmx.ops.softmax(x3.to(cuda), 1), F.softmax(x3.double(), 1), dtype) <= _tol("softmax", dtype)
    w = (make("normal", (1500,), seed=12) * 0.1 + 1).to(dtype)
    b = (make("normal", (1500,), seed=13) * 0.1).to(dtype)
    truth, floor = _ln_truth(x, 1500, w, b)
    assert err_in_ulps(dmx.ops.layernorm(x.to(cuda), (1500,), w.to(cuda), b.to(cuda), 1e-5), truth, dtype, floor=floor) <= _tol("layernorm", dtype)
    big = (make("normal", (3, 20000), seed=14)).to(dtype)                     # longer than the LDS row buffer
    assert err_in_ulps(dmx.ops.softmax(big.to(cuda), -1), F.softmax(big.double(), -1), dtype) <= _tol("softmax", dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_histc_matches_oracle(dmx, cuda, oracle, dtype):
    """dmxq_histc == the oracle's restatement of ATen's histc: ragged sizes (vector body + scalar tail, unaligned
    views), every bins/range shape the observer produces, NaN / inf / out-of-range elements dropped."""
    for i, (n, bins, lo, hi) in enumerate([(1, 2048, -1, 1), (7, 5, 0, 3), (4099, 2048, -3, 4), (100003, 2048, -9, 13),
                                           (1 << 20, 2048, -4, 4), (65536, 8192, -2, 2), (5000, 1, -1, 1), (30000, 2048, 0, 0)]):
        x = make("normal", (n,), seed=70 + i, dtype=dtype) * 3
        if n > 100:
            x[3] = float("nan"); x[5] = float("inf"); x[9] = float("-inf"); x[11] = lo; x[13] = hi
        if lo == hi:
            x = x.nan_to_num(0.0, 1.0, -1.0)
        want = oracle.histc(x, bins, lo, hi)
        assert bits_equal(dmx.ops.histc(x.to(cuda), bins, lo, hi), want) == 0, (n, bins, lo, hi)
        if n > 16:  # a view that starts off the 16-byte grid
            assert bits_equal(dmx.ops.histc(x.to(cuda)[1:], bins, lo - 1, hi + 1), oracle.histc(x[1:], bins, lo - 1, hi + 1)) == 0
    const = torch.full((1000,), 2.5, dtype=dtype)
    assert bits_equal(dmx.ops.histc(const.to(cuda), 16), oracle.histc(const, 16, 0, 0)) == 0
    with pytest.raises(NotImplementedError):
        dmx.ops.histc(const.to(cuda), 1 << 14, -1, 1)   # more bins than the LDS histogram holds
    with pytest.raises(dmx.DmxqError):
        dmx.ops.histc(const.to(cuda), 16, 3, 1)


def test_histogram_observer_group_equals_tensor(dmx, cuda):
    """tests/test_group_quant.py:152-188 (HistogramObserver leg): one group spanning the whole ch_axis == per tensor."""
    W = make("normal", (24, 40), seed=5).to(cuda)
    for fmt in ("XP[8,0](CSN)", "XP[4,0](CSN)"):
        for qs in (torch.per_tensor_affine, torch.per_tensor_symmetric):
            a, b = dmx.CastTo(format=fmt), dmx.CastTo(format=fmt)
            a.enable_calibration(True, dmx.HistogramObserver, qs, group_size=W.shape[-1], ch_axis=-1)
            b.enable_calibration(True, dmx.HistogramObserver, qs)
            a(W); b(W)
            a.enable_calibration(False); b.enable_calibration(False)
            assert torch.allclose(a(W), b(W), rtol=0.0, atol=1e-8)
    with pytest.raises(NotImplementedError):
        dmx.HistogramObserver(qscheme=torch.per_channel_affine)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_softmax_layernorm_every_row_kernel_shape(dmx, cuda, dtype):
    """Row lengths that select every register-resident kernel shape (approx.hip: 64 / 32 lanes per row, 1..16 vectors
    per lane, 8-byte vectors for 16-bit rows that are only a multiple of 4, workgroup-per-row for long rows) and the
    LDS / global fallbacks (odd lengths, mixed dtypes), with ragged row counts so that the clamped out-of-range rows
    and lanes are exercised.  Same tolerances as above (float64 truth)."""
    F = torch.nn.functional
    from _data import err_in_ulps
    for n, cols in enumerate((4, 8, 64, 252, 256, 264, 768, 1024, 1280, 1500, 1536, 2048, 2304, 2560, 3072, 4096, 5120,
                              6144, 7168, 8192, 12288, 16384, 777, 1501)):
        rows = (1, 7, 33, 130)[n % 4]
        x = (make("normal", (rows, cols), seed=100 + n) * 2).to(dtype)
        if rows > 2:
            x[1, : cols // 2] = float("-inf")      # a masked row: exp(-inf) = 0
        assert err_in_ulps(dmx.ops.softmax(x.to(cuda), -1), F.softmax(x.double(), -1), dtype) <= _tol("softmax", dtype), ("softmax", cols, rows)
        if rows > 2:
            y = x.clone(); y[2] = float("-inf")    # a fully masked row is NaN in torch too
            assert err_in_ulps(dmx.ops.softmax(y.to(cuda), -1), F.softmax(y.double(), -1), dtype) <= _tol("softmax", dtype), ("softmax -inf row", cols)
        x = (make("normal", (rows, cols), seed=200 + n) * 2 + 0.5).to(dtype)
        w = (make("normal", (cols,), seed=12) * 0.1 + 1).to(dtype)
        b = (make("normal", (cols,), seed=13) * 0.1).to(dtype)
        for ww, bb in ((w, b), (w, None), (None, None)):
            got = dmx.ops.layernorm(x.to(cuda), (cols,), None if ww is None else ww.to(cuda), None if bb is None else bb.to(cuda), 1e-5)
            truth, floor = _ln_truth(x, cols, ww, bb)
            assert err_in_ulps(got, truth, dtype, floor=floor) <= _tol("layernorm", dtype), ("layernorm", cols, rows, ww is not None, bb is not None)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_topk_mask_matches_oracle(dmx, cuda, oracle, dtype):
    """Global TOPK{density} (sparse.py:109-123) by radix select: every density edge, sizes around the chunk / workgroup
    boundaries, heavy ties at the threshold (16-bit scores, integer scores, constant tensors: the index-ordered tie
    path), -0 / +0, NaN (largest), inf; mask and fused x * mask."""
    for n, (shape, density) in enumerate([((1,), 0.5), ((7,), 0.5), ((2048,), 0.25), ((2049,), 0.75), ((333, 77), 0.5),
                                          ((64, 4096), 0.5), ((1000, 1000), 0.1), ((5, 1 << 17), 0.9), ((4097,), 0.0),
                                          ((4097,), 1.0), ((123, 457), 0.999), ((123, 457), 0.001)]):
        s = make("normal", shape, seed=300 + n, dtype=dtype)
        want = oracle.topk_mask(s, density)
        got = dmx.ops.topk_mask(s.to(cuda), density)
        assert bits_equal(got, want) == 0, (shape, density)
        assert int(want.float().sum()) == s.numel() - int(s.numel() * (1.0 - density))
    # ties everywhere
    for n, s in enumerate([torch.zeros(5000), torch.ones(3, 4099), (make("normal", (70000,), seed=5) * 3).round(),
                           torch.tensor([0.0, -0.0] * 3000), make("normal", (1 << 18,), seed=6).to(torch.bfloat16).float()]):
        s = s.to(dtype)
        if n == 2:
            s[17] = float("nan"); s[900] = float("inf"); s[901] = float("-inf"); s[55555] = float("nan")
        for density in (0.3, 0.5, 0.97):
            assert bits_equal(dmx.ops.topk_mask(s.to(cuda), density), oracle.topk_mask(s, density)) == 0, (n, density)
    s = make("normal", (257, 1025), seed=9, dtype=dtype)
    x = make("heavy", (257, 1025), seed=10, dtype=dtype)
    y, m = dmx.ops.topk_sparsify(x.to(cuda), s.to(cuda), 0.5, return_mask=True)
    assert bits_equal(m, oracle.topk_mask(s, 0.5)) == 0 and mismatches_nan_aware(y, x * oracle.topk_mask(s, 0.5)) == 0
    # module level: Sparsify with the TOPK shorthand, inference (fused) and training (mask * x with autograd)
    sp = dmx.Sparsify(s.shape, "TOPK{0.5}(U)").to(cuda)
    sp.score.data = s.to(cuda)
    with torch.no_grad():
        assert mismatches_nan_aware(sp(x.to(cuda)), x * oracle.topk_mask(s, 0.5)) == 0
    xg = x.to(cuda).float().requires_grad_(True)
    sp.score.data = s.to(cuda).float()
    out = sp(xg)
    out.sum().backward()
    assert torch.equal(xg.grad.cpu(), oracle.topk_mask(s.float(), 0.5))


def test_bernoulli_mask(dmx, cuda, oracle):
    """BERN (sparse.py:201-242): the counter-based draws are reproducible and bit-identical to the oracle's; against the
    reference (torch.bernoulli on the global generator) parity is statistical: mean of the mask = mean of the scores."""
    p = (make("normal", (1 << 20,), seed=77) * 0.2 + 0.5).clamp(0, 1)
    p[:3] = torch.tensor([0.0, 1.0, 0.5])
    got = dmx.ops.bernoulli_mask(p.to(cuda), seed=1234)
    assert bits_equal(got, oracle.bernoulli_mask(p, 1234)) == 0
    assert got[0] == 0 and got[1] == 1
    assert abs(float(got.mean()) - float(p.mean())) < 3e-3
    assert not torch.equal(got, dmx.ops.bernoulli_mask(p.to(cuda), seed=1235))
    sp = dmx.Sparsify(p.shape, "BERN").to(cuda)
    sp.score.data = p.to(cuda)
    x = torch.ones_like(p).to(cuda)
    with torch.no_grad():
        y = sp(x)
    assert abs(float(y.mean()) - float(p.mean())) < 3e-3 and set(y.unique().tolist()) <= {0.0, 1.0}
    sp.score.data = (p * 3).to(cuda)
    with pytest.raises(AssertionError):  # scores outside [0, 1] (sparse.py:211-213)
        sp(x)


@pytest.mark.parametrize("sparseness", ["TOPK{0.5}(U)", "TOPK{0.5}(M)", "BTOPK{4:8,-1}(U)", "BTOPK{4:8,-1}(M)", "BTOPK{2:8,-1}(U)", "BERN"])
@pytest.mark.parametrize("backward_mode", ["STE", "supermask", "joint"])
def test_sparsify_gradient_routing(dmx, cuda, sparseness, backward_mode):
    """The reference's tests/test_sparse.py::test_sparsify: which of (input, score) receives a gradient in each
    backward mode, for every sparseness class; plus the values: d(x * mask)/dx = mask, d/dscore = x (identity through
    the mask function)."""
    for shape in ((64, 128), (4, 16, 8, 8), (8, 32, 32)):
        sp = dmx.Sparsify(shape, sparseness, backward_mode).to(cuda)
        x = torch.randn(shape, requires_grad=True, device=cuda)
        y = sp(x)
        y.backward(torch.ones_like(y))
        if backward_mode == "STE":
            assert isinstance(x.grad, torch.Tensor) and sp.score.grad is None
            assert torch.equal(x.grad, sp.mask.to(x.grad.dtype))
        elif backward_mode == "supermask":
            assert x.grad is None and isinstance(sp.score.grad, torch.Tensor)
            assert torch.equal(sp.score.grad, x.detach())
        else:
            assert isinstance(x.grad, torch.Tensor) and isinstance(sp.score.grad, torch.Tensor)
        assert torch.equal(y.detach(), x.detach() * sp.mask.detach())


def test_sparsify_reconfiguration(dmx):
    """tests/test_sparse.py::test_transformation: sparseness / backward mode / score function can be swapped in place."""
    sp = dmx.Sparsify((8, 16))
    assert repr(sp.sparseness) == "DENSE" and sp.backward_mode == "STE"
    sp.configure(sparseness="BERN", backward_mode="supermask", score_func=lambda score, input: score)
    assert repr(sp.sparseness) == "BERN" and sp.backward_mode == "supermask" and sp.plastic
    sp.sparseness = dmx.Sparseness.from_shorthand("DENSE")
    sp.backward_mode = "STE"
    assert repr(sp.sparseness) == "DENSE"
    sp.configure(sparseness="TOPK{0.5}(U)", backward_mode="joint", score_func=lambda score, input: torch.abs(input))
    assert repr(sp.sparseness) == "TOPK{0.5}(U)" and sp.backward_mode == "joint"
    sp.sparseness = dmx.Sparseness.from_shorthand("BTOPK{4:8,-1}(U)")
    assert repr(sp.sparseness) == "BTOPK{4:8,-1}(U)"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gelu_tails_and_special_values(dmx, cuda, dtype):
    """GELU (erf and tanh forms; closed-form fast paths for 16-bit outputs): the negative tail where 1 + erf cancels,
    saturation, signed zeros, inf / NaN exactly as torch."""
    F = torch.nn.functional
    eps = torch.finfo(dtype).eps
    x = torch.cat([torch.linspace(-14, 14, 4001), torch.tensor([0.0, -0.0, 1e-30, -1e-30, 20.0, -20.0, 100.0, -100.0, 1e4, -1e4,
                                                               float("inf"), float("-inf"), float("nan")])]).to(dtype)
    for approx in ("none", "tanh"):
        got = dmx.ops.gelu(x.to(cuda), approx).cpu().float()
        ref = F.gelu(x.float(), approximate=approx).to(dtype).float()
        if approx == "none":  # torch's vectorised CPU erf form returns NaN at +inf; the mathematical value is +inf
            keep = x.float() != float("inf")
            got, ref, xs = got[keep], ref[keep], x.float()[keep]
        else:
            xs = x.float()
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), approx
        assert torch.equal(got[~fin & ~torch.isnan(ref)], ref[~fin & ~torch.isnan(ref)]), approx
        assert int(((got[fin] - ref[fin]).abs() > eps * ref[fin].abs() + 2e-6).sum()) == 0, approx
        z = (xs == 0) | (xs.abs() >= 20)       # exact results incl. the sign of zero
        z &= fin
        assert torch.equal(got[z].view(torch.int32), ref[z].view(torch.int32)), approx


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_softmax_on_a_view_that_starts_mid_allocation(dmx, cuda, dtype):
    """Rows at element alignment only (a view one element into its buffer): the unaligned-access form of the row kernel."""
    F = torch.nn.functional
    eps = torch.finfo(dtype).eps
    for rows, cols in ((9, 256), (5, 1500), (3, 197), (2, 4096)):
        base = (make("normal", (rows * cols + 1,), seed=cols, dtype=dtype) * 2)
        xg = base.to(cuda)[1:].view(rows, cols)
        got = dmx.ops.softmax(xg, -1).cpu().float()
        ref = F.softmax(base[1:].view(rows, cols).float(), -1).to(dtype).float()
        assert int(((got - ref).abs() > eps * ref.abs() + 2e-6).sum()) == 0, (rows, cols)
