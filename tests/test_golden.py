"""Golden-vector tests.  tests/golden/*.npz hold inputs and the outputs of the REAL reference (generated in the
build container by oracle/gen_golden.py, which runs d-matrix-ai/dmx-compressor's own CastTo / Sparsify /
observer / SmoothQuant code).  Two consumers:
  * -m "not gpu": the CPU oracle must reproduce every stored output bit for bit  -> the oracle is pinned;
  * -m gpu      : the HIP kernels (through the C ABI) must reproduce them too     -> parity with the reference.
"""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
T_BITS = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.float16: torch.int16}


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def tensor(arr, dtype):
    a = np.ascontiguousarray(arr)
    t = torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int16))
    return t.view(dtype)


def mism(got, want_bits, dtype):
    """bit mismatches, any NaN == any NaN"""
    want = tensor(want_bits, dtype)
    got = got.detach().cpu().contiguous()
    assert got.dtype == dtype and got.shape == want.shape, (got.dtype, got.shape, want.shape)
    both_nan = torch.isnan(got.float()) & torch.isnan(want.float())
    return int(((got.view(T_BITS[dtype]) != want.view(T_BITS[dtype])) & ~both_nan).sum())


class OracleBackend:
    """runs a case on the CPU oracle"""

    def __init__(self, O):
        self.O = O
        self.dev = torch.device("cpu")

    def bfp(self, x, wl, B, dim, sym, out_dtype):
        return self.O.bfp_cast(x, wl, B, dim, sym).to(out_dtype).contiguous()

    def fmt(self, x, sh, out_dtype):
        import dmx_compressor_amd as d
        f = d.Format.from_shorthand(sh)
        if isinstance(f, d.BlockFloatingPoint):
            y = self.O.bfp_cast(x, f.precision, f.block_size)
        elif isinstance(f, d.FloatingPoint):
            y = self.O.floating_point_cast(x, f.mantissa, f.exponent, f.bias, f.flush_subnormal, f.unsigned)
        else:
            y = self.O.fixed_point_cast(x, f.precision, f.fraction, f.clamp, f.symmetric)
        return y.to(out_dtype)

    def affine(self, x, f, sc, zp, ch_axis, gs):
        return self.O.fixed_point_affine_cast(x, f.precision, f.fraction, f.clamp, f.symmetric, sc, zp, ch_axis=ch_axis, group_size=gs)

    def nm(self, s, K, M, dim):
        return self.O.nm_mask(s, K, M, dim).contiguous()

    def sparsify(self, x, s, K, M):
        return self.O.sparsify(x, s, K, M)

    def maxabs(self, x, ax):
        return self.O.channel_maxabs(x, ax)

    def minmax_qparams(self, x, ax, gs, per_channel, f, sym_q):
        if gs:
            mn, mx = self.O.group_minmax(x, ax, gs)
        elif per_channel:
            mn, mx = self.O.group_minmax(x, ax, 1)
        else:
            mn, mx = self.O.group_minmax(x.reshape(1, -1), 0, 1)
        return self.O.qparams(mn, mx, f.precision, f.symmetric, sym_q)


class HipBackend:
    """runs a case on the GPU through the host mirror -> ctypes -> C ABI"""

    def __init__(self, d, dev):
        self.d, self.dev = d, dev

    def bfp(self, x, wl, B, dim, sym, out_dtype):
        return self.d.ops.bfp_qdq(x.to(self.dev), wl, B, dim, sym, out_dtype=out_dtype)

    def fmt(self, x, sh, out_dtype):
        return self.d.CastTo(format=sh)(x.to(self.dev))

    def affine(self, x, f, sc, zp, ch_axis, gs):
        return self.d.ops.fixed_qdq(x.to(self.dev), f.precision, f.fraction, f.clamp, f.symmetric, scale=sc, zero_point=zp,
                                    ch_axis=ch_axis, group_size=gs)

    def nm(self, s, K, M, dim):
        return self.d.ops.nm_mask(s.to(self.dev), K, M, dim)

    def sparsify(self, x, s, K, M):
        return self.d.ops.nm_sparsify(x.to(self.dev), s.to(self.dev), K, M)

    def maxabs(self, x, ax):
        return self.d.ops.channel_maxabs(x.to(self.dev), ax)

    def minmax_qparams(self, x, ax, gs, per_channel, f, sym_q):
        obs = self.d.MinMaxObserver(dtype=f, qscheme={(True, True): torch.per_channel_symmetric, (True, False): torch.per_channel_affine,
                                                     (False, True): torch.per_tensor_symmetric, (False, False): torch.per_tensor_affine}[(per_channel, sym_q)],
                                    ch_axis=ax)
        obs(x.to(self.dev), gs)
        return obs.calculate_qparams()


def backends():
    return [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def backend(request, oracle, dmx):
    if request.param == "oracle":
        return OracleBackend(oracle)
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return HipBackend(dmx, torch.device("cuda:0"))


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_bfp_fixtures(backend):
    g = load("bfp.npz")
    n = 0
    for key in g.files:
        if not key.startswith("y_"):
            continue
        _, dt, wl, B, sym = key.split("_")
        x = tensor(g[("x_" if sym == "S" else "xa_") + dt], DT[dt])
        got = backend.bfp(x, int(wl), int(B), -1, sym == "S", DT[dt])
        assert mism(got, g[key], DT[dt]) == 0, key
        n += 1
    assert n == 3 * 4 * (5 * 2 - 1)
    xr = tensor(g["ragged_x"], torch.float32)
    for B in (16, 24, 64):
        assert mism(backend.bfp(xr, 8, B, -1, True, torch.float32), g[f"ragged_y_{B}"], torch.float32) == 0, B
    xc = tensor(g["conv_x"], torch.bfloat16)
    for dim in (-1, -2, 1, 0):
        for B in (16, 64):
            assert mism(backend.bfp(xc, 8, B, dim, True, torch.bfloat16), g[f"conv_y_{dim}_{B}"], torch.bfloat16) == 0, (dim, B)
    adv = tensor(g["adv_x"], torch.float32)
    assert mism(backend.bfp(adv, 8, 16, -1, True, torch.float32), g["adv_y_S"], torch.float32) == 0
    assert mism(backend.bfp(adv, 8, 16, -1, False, torch.float32), g["adv_y_A"], torch.float32) == 0


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_float_and_fixed_fixtures(backend, dmx):
    g = load("elementwise.npz")
    for i, sh in enumerate([str(s) for s in g["float_sh"]]):
        for tag in ("f32", "bf16"):
            x = tensor(g[f"x_{tag}"], DT[tag])
            assert mism(backend.fmt(x, sh, DT[tag]), g[f"float{i}_{tag}"], DT[tag]) == 0, (sh, tag)
    x = tensor(g["x_f32"], torch.float32)
    for i, sh in enumerate([str(s) for s in g["fixed_sh"]]):
        assert mism(backend.fmt(x, sh, torch.float32), g[f"fixed{i}_f32"], torch.float32) == 0, sh
    # reference KAT (tests/test_group_quant.py:49-63) as stored by the generator
    xk, yk = tensor(g["kat_group_x"], torch.float32), tensor(g["kat_group_y"], torch.float32)
    assert torch.allclose(yk, torch.tensor([[0, 1], [3, 7], [6, 8], [10, 14], [0.1, 0.7]]), rtol=0.0, atol=1e-6)
    f = dmx.format.INT4
    sc, zp = backend.minmax_qparams(xk, 0, 2, False, f, True)
    assert mism(backend.affine(xk, f, sc, zp, 0, 2), g["kat_group_y"], torch.float32) == 0


AFFINE = [("tensor_sym", False, True, -1, None), ("tensor_aff", False, False, -1, None), ("chan0_sym", True, True, 0, None),
          ("chan1_aff", True, False, 1, None), ("group16_sym", False, True, 0, 16), ("group7_sym_ragged", False, True, 0, 7),
          ("group5_aff_axis1", False, False, 1, 5)]


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_affine_group_quant_fixtures(backend, dmx):
    """MinMax observer -> (scale, zero_point) -> fused affine fixed-point cast, all three stages vs the reference."""
    g = load("elementwise.npz")
    W = tensor(g["aff_w"], torch.float32)
    for sh in ("XP[8,0](CSN)", "XP[4,0](CSN)", "XP[8,0](C_N)"):
        f = dmx.Format.from_shorthand(sh)
        for name, per_channel, sym_q, ax, gs in AFFINE:
            key = f"aff_{sh}_{name}"
            sc, zp = backend.minmax_qparams(W, ax, gs, per_channel, f, sym_q)
            assert mism(sc.reshape(-1), g[key + "_scale"], torch.float32) == 0, key
            assert np.array_equal(zp.cpu().numpy().reshape(-1), g[key + "_zp"]), key
            got = backend.affine(W, f, sc, zp, ax if (gs or per_channel) else None, gs)
            assert mism(got, g[key + "_y"], torch.float32) == 0, key


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_composite_block_format_fixtures(backend, dmx, oracle):
    """SBFP (weight-storage format) and MXFP through the reference's CastTo: format.py:453-479, 545-564."""
    g = load("composite.npz")
    hip = isinstance(backend, HipBackend)

    def run(sh, x, dim=-1):
        f = dmx.Format.from_shorthand(sh)
        if hip:
            return dmx.CastTo(format=f, block_dim=dim)(x.to(backend.dev))
        if isinstance(f, dmx.ScaledBlockFloatingPoint):
            bf, sf = f.block_format, f.scaler_format
            y = oracle.sbfp_cast(x, bf.precision, f.block_size, sf.mantissa, sf.exponent, sf.bias, sf.flush_subnormal,
                                 bf.clamp, bf.symmetric, dim)
        else:
            y = oracle.mxfp_cast(x, f.element_format.mantissa, f.element_format.exponent, f.block_size, dim)
        return y.to(x.dtype).contiguous()

    for i, sh in enumerate([str(s) for s in g["sbfp_sh"]]):
        for k, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f32z", torch.float32)):
            assert mism(run(sh, tensor(g[f"sx_{k}"], dt)), g[f"sbfp{i}_{k}"], dt) == 0, (sh, k)
    sh0 = str(g["sbfp_sh"][0])
    assert mism(run(sh0, tensor(g["sx_ragged"], torch.float32)), g["sbfp_ragged"], torch.float32) == 0
    assert mism(run(sh0, tensor(g["sx_f32"], torch.float32), 0), g["sbfp_dim0"], torch.float32) == 0
    for i, sh in enumerate([str(s) for s in g["mxfp_sh"]]):
        for k, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            assert mism(run(sh, tensor(g[f"mx_{k}"], dt)), g[f"mxfp{i}_{k}"], dt) == 0, (sh, k)


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_boundaries_mxfp_maxima_just_below_powers_of_two(backend, dmx, oracle):
    """MXFP.cast (format.py:545-564) scales a block by 2^floor(log2 max) with a FLOAT32 log2 (:551-553), which rounds to the
    integer v for a maximum 2^v (1 - j 2^-24) with small j -- the scale is then twice the "exact" one.  1534 such blocks
    (59 exponents x 26 distances j = 1 .. 96; an exact floor would get 448 of them wrong) cast by the reference itself
    (oracle/gen_golden_r3.py, which also checks the closed rule of oracle_floor_log2f against torch.log2 for every float32
    exponent): oracle and kernel must reproduce every bit."""
    g = load("boundaries.npz")
    x = tensor(g["mx_x"], torch.float32)
    hip = isinstance(backend, HipBackend)
    for i, sh in enumerate([str(s) for s in g["mx_sh"]]):
        f = dmx.Format.from_shorthand(sh)
        if hip:
            y = dmx.CastTo(format=f)(x.to(backend.dev))
        else:
            y = oracle.mxfp_cast(x, f.element_format.mantissa, f.element_format.exponent, f.block_size)
        assert mism(y, g[f"mx_y{i}"], torch.float32) == 0, sh
    if hip:  # the same blocks inside longer rows (several blocks per lane group), and as bf16 (the exponent-field path: never rounds up)
        xl = x.reshape(-1, 8 * 26)
        f = dmx.Format.from_shorthand("MXFP8[E4M3]{8}")
        assert mism(dmx.CastTo(format=f)(xl.to(backend.dev)).reshape(x.shape), g["mx_y0"], torch.float32) == 0
        xb = x.to(torch.bfloat16)
        want = oracle.mxfp_cast(xb, 3, 4, 8).to(torch.bfloat16)
        got = dmx.CastTo(format=f)(xb.to(backend.dev)).cpu()
        assert int(((got.view(torch.int16) != want.view(torch.int16)) & ~(torch.isnan(got.float()) & torch.isnan(want.float()))).sum()) == 0


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_boundaries_asymmetric_bfp_with_planted_nonfinite_and_denormal_blocks(backend, dmx):
    """`BFP[p|8]{16}(_N)` (format.py:304-372) on tensors with planted Inf / NaN / denormal-maximum blocks, against the reference's
    own output.  The reference's post-pass rebuilds a whole [rows, 16] chunk from integers iff ANY row of the
    chunk holds an edge code (format.py:362-370): a block whose maximum is Inf / NaN then becomes `ldexp(int(NaN), ..)` garbage
    instead of NaN, and a -0.0 result (which only a block with a DENORMAL maximum can produce) becomes +0.0.  Blocks are independent
    here.  Asserted: (1) bit-equality with the reference on every element outside those two sets -- in particular
    on every finite block, edge codes and all-zero blocks included; (2) inside a poisoned block this library returns exactly what
    its SYMMETRIC format returns for that block; (3) in denormal-maximum blocks only the sign of zeros may differ; (4) the
    differing set is exactly as large as recorded when the fixture was made (64 = 4 blocks x 16; 15 zero signs, float32 only)."""
    g = load("boundaries.npz")
    for nm, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x = tensor(g[f"asym_x_{nm}"], dt)
        xf = x.float().reshape(48, 4, 16)
        poisoned = ~torch.isfinite(xf).all(-1, keepdim=True).expand_as(xf).reshape(48, 64)
        den = (xf.abs().amax(-1, keepdim=True) < 2.0 ** -126).expand_as(xf).reshape(48, 64)
        assert int(poisoned.sum()) == 64
        for i, sh in enumerate([str(s) for s in g["asym_sh"]]):
            f = dmx.Format.from_shorthand(sh)
            got = backend.bfp(x, f.precision, f.block_size, -1, False, dt).cpu()
            sym = backend.bfp(x, f.precision, f.block_size, -1, True, dt).cpu()
            want = tensor(g[f"asym_y{i}_{nm}"], dt)
            it = T_BITS[dt]
            nan2 = torch.isnan(got.float()) & torch.isnan(want.float())
            diff = (got.view(it) != want.view(it)) & ~nan2
            zero_sign = den & (got.float() == 0) & (want.float() == 0)
            assert int((diff & ~poisoned & ~zero_sign).sum()) == 0, (sh, nm)                       # (1)
            same_as_sym = (got.view(it) == sym.view(it)) | (torch.isnan(got.float()) & torch.isnan(sym.float()))
            assert bool(same_as_sym[poisoned].all()), (sh, nm)                                      # (2)
            assert bool(((got.float() == want.float()) | nan2 | poisoned)[den].all()), (sh, nm)     # (3)
            assert int((diff & poisoned).sum()) == 64 and int((diff & zero_sign).sum()) == (15 if dt == torch.float32 else 0), (sh, nm)  # (4)


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_nm_mask_fixtures(backend):
    g = load("nm_mask.npz")
    n = 0
    for key in g.files:
        if not key.startswith("m_"):
            continue
        body, K, M, dim = key[2:].rsplit("_", 3)
        sdt = torch.bfloat16 if body == "absbf16" else torch.float32
        s = tensor(g[f"s_{body}"], sdt)
        assert mism(backend.nm(s, int(K), int(M), int(dim)), g[key], sdt) == 0, key
        n += 1
    assert n == 6 * 6 * 2
    x, s = tensor(g["sp_x"], torch.bfloat16), tensor(g["s_random"], torch.float32)
    assert mism(backend.sparsify(x, s, 2, 4), g["sp_y"], torch.float32) == 0


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_topk_fixtures(backend, dmx, oracle):
    """TOPK{density} masks recorded from the reference's TopK.get_mask (inputs without ties: its argsort is unstable)
    and Sparsify.forward with the TOPK shorthand."""
    g = load("topk.npz")
    hip = isinstance(backend, HipBackend)
    for n in range(int(g["n"])):
        s = tensor(g[f"s{n}"], torch.float32)
        for density in (0.5, 0.25, 0.9, 0.01, 1.0, 0.0):
            got = dmx.ops.topk_mask(s.to(backend.dev), density) if hip else oracle.topk_mask(s, density)
            assert mism(got, g[f"m{n}_{density}"], torch.float32) == 0, (n, density)
    x, s = tensor(g["sp_x"], torch.bfloat16), tensor(g["s0"], torch.float32)
    y = dmx.ops.topk_sparsify(x.to(backend.dev), s.to(backend.dev), 0.5) if hip else x * oracle.topk_mask(s, 0.5)
    assert mism(y, g["sp_y"], torch.float32) == 0


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_smoothquant_fixtures(backend, dmx):
    g = load("smoothquant.npz")
    a, w = tensor(g["a"], torch.float32), tensor(g["w"], torch.float32)
    am, wm = backend.maxabs(a, -1), backend.maxabs(w, -1)
    assert mism(am, g["a_maxabs"], torch.float32) == 0 and mism(wm, g["w_maxabs"], torch.float32) == 0
    if isinstance(backend, HipBackend):
        for alpha in (0.0, 0.25, 0.5, 1.0):
            got = dmx.ops.smoothquant_scale(am, wm, alpha, 1e-5).cpu()
            want = tensor(g[f"scale_{alpha}"], torch.float32)
            # floating point (two powf + a divide on different libms): within 4 ulp of fp32
            assert torch.allclose(got, want, rtol=4.8e-7, atol=0.0), alpha


@pytest.mark.parametrize("backend", backends(), indirect=True)
def test_histc_fixtures(backend, dmx, oracle):
    """torch.histc outputs recorded from ATen (the reference's call, observer.py:470-472/489-491)."""
    g = load("histogram.npz")
    for i in range(int(g["n_histc"])):
        x = tensor(g[f"histc{i}_x"], torch.float32)
        bins, lo, hi = (int(v) for v in g[f"histc{i}_cfg"])
        if isinstance(backend, HipBackend):
            got = dmx.ops.histc(x.to(backend.dev), bins, lo, hi)
        else:
            got = oracle.histc(x, bins, lo, hi)
        assert mism(got, g[f"histc{i}_out"], torch.float32) == 0, (i, bins, lo, hi)


@pytest.mark.gpu
def test_histogram_observer_fixtures(dmx, cuda):
    """HistogramObserver state after each batch of a sequence (histogram, running range, scale, zero point) and
    CastTo calibration through it, per tensor and per group -- recorded from the reference's own classes."""
    g = load("histogram.npz")
    qs = {"affine": torch.per_tensor_affine, "symmetric": torch.per_tensor_symmetric}
    for s in range(int(g["n_seq"])):
        obs = dmx.HistogramObserver(dtype=dmx.Format.from_shorthand(str(g[f"seq{s}_fmt"])), qscheme=qs[str(g[f"seq{s}_qs"])])
        for b in range(int(g["n_batch"])):
            obs(tensor(g[f"seq{s}_x{b}"], torch.float32).to(cuda))
            assert mism(obs.histogram, g[f"seq{s}_hist{b}"], torch.float32) == 0, (s, b)
            assert mism(torch.stack([obs.min_val, obs.max_val]), g[f"seq{s}_range{b}"], torch.float32) == 0, (s, b)
            scale, zp = obs.calculate_qparams()
            assert mism(scale.reshape(1), g[f"seq{s}_scale{b}"], torch.float32) == 0, (s, b)
            assert int(zp.reshape(-1)[0]) == int(g[f"seq{s}_zp{b}"][0]), (s, b)
    w = tensor(g["w"], torch.float32)
    for name, kw in (("tensor", dict()), ("group16", dict(group_size=16, ch_axis=-1)), ("group64", dict(group_size=64, ch_axis=-1))):
        c = dmx.CastTo(format="XP[8,0](CSN)")
        c.enable_calibration(True, dmx.HistogramObserver, torch.per_tensor_symmetric, **kw)
        c(w.to(cuda))
        c.enable_calibration(False)
        assert mism(c.scale.float().reshape(-1), g[f"cast_{name}_scale"], torch.float32) == 0, name
        assert np.array_equal(c.zero_point.reshape(-1).cpu().numpy().astype(np.int64), g[f"cast_{name}_zp"]), name
        assert mism(c(w.to(cuda)).float(), g[f"cast_{name}_out"], torch.float32) == 0, name


def test_vocabulary_matches_reference(dmx):
    g = load("vocabulary.npz")
    names, reprs = [str(s) for s in g["format_names"]], [str(s) for s in g["format_reprs"]]
    mine = vars(dmx.format)
    assert sorted(mine) == names
    for n, r in zip(names, reprs):
        assert repr(mine[n]) == r, n
    for n, r in zip([str(s) for s in g["sparse_names"]], [str(s) for s in g["sparse_reprs"]]):
        assert repr(getattr(dmx.sparseness, n)) == r, n


def test_native_asym_oracle_matches_the_reference_extension(oracle):
    """oracle.block_quantize_native(symmetric=False) == the reference's compiled block_quantize_*(a, wl, dim, False)
    (quant_cpu.cpp:247-253) on the committed fixture (oracle/gen_golden_r2.py)."""
    z = np.load(os.path.join(GOLD, "native_asym.npz"))
    shapes = {0: (16, 64), -1: (8, 32), 1: (4, 16, 8), 2: (4, 16, 8)}
    n_diff = 0
    for i in range(int(z["n"])):
        dim = int(z[f"dim{i}"])
        x = tensor(z[f"x{i}"], torch.float32).reshape(shapes[dim])
        for wl in (4, 8, 12):
            for rnd in ("nearest", "down", "up"):
                want = tensor(z[f"y{i}_{wl}_{rnd}"], torch.float32).reshape(x.shape)
                got = oracle.block_quantize_native(x, wl, dim, False, rnd)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (i, wl, rnd)
                n_diff += int(not torch.equal(oracle.block_quantize_native(x, wl, dim, True, rnd).view(torch.int32), want.view(torch.int32)))
    assert n_diff > 50   # the fixture really exercises the branch


def test_experimental_silu_oracle_matches_the_reference_module(oracle):
    z = np.load(os.path.join(GOLD, "approx.npz"))
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        x = tensor(z[f"x_{name}"], dt)
        for i in range(4):
            got = oracle.silu_experimental(x, float(z[f"scale_{i}"]))
            want = tensor(z[f"raw_{name}_{i}"], torch.float16)
            both_nan = torch.isnan(got) & torch.isnan(want)
            assert bool(((got.view(torch.int16) == want.view(torch.int16)) | both_nan).all())


def test_packed_bfp_oracle_is_consistent_with_the_cast_oracle(oracle):
    """codes * 2^(exps - 127 - (p - 2)) reproduces the oracle's Q->DQ wherever the block maximum is a normal number"""
    from _data import make
    x = make("mixed", (16, 200), seed=3, dtype=torch.float32, block=16)
    for wl, B, sym in ((8, 16, True), (8, 64, False), (4, 24, True)):
        mant, exps = oracle.bfp_pack(x, wl, B, sym)
        q = oracle.bfp_cast(x, wl, B, -1, sym)
        e = exps.repeat_interleave(B, dim=-1)[..., : x.shape[-1]].to(torch.int32)
        y = torch.ldexp(mant.float(), e - 127 - (wl - 2))
        ok = (e > 0) & (e < 255)
        assert torch.equal(y[ok].view(torch.int32), q[ok].view(torch.int32))


# ------------------------------------------------------------------------------------------------ BASELINE.json config 2 itself
def test_chunked_generator_equals_make():
    """tests/_data.py make_chunked (bench.py's input generator) == make, bit for bit, incl. a ragged last chunk"""
    from _data import bits_equal, make, make_chunked

    for kind, shape, seed, dt in (("heavy", (37, 129), 3, torch.bfloat16), ("normal", (5, 1000), 1, torch.float32),
                                  ("heavy", (1 << 10, 1 << 10), 1019, torch.float16)):
        assert bits_equal(make_chunked(kind, shape, seed, dt, workers=3, chunk=1000), make(kind, shape, seed, dt)) == 0


def test_c2_oracle_reproduces_the_reference_digests(oracle):
    """tests/golden/c2_digests.json (oracle/gen_golden_r4.py): SHA-256 of the reference's CastTo("BFP[8|8]{16}(SN)") output on the
    counter-generated 4096 x 4096 bf16 tensors, 20 rotation slots for each of the eight ranks of the scaling run (round 6; ranks 0 / 1 since
    round 4).  The oracle must reproduce them (three of the 164 here: seconds, not minutes)."""
    import json

    from _data import make_chunked, sha256_bits

    g = json.load(open(os.path.join(GOLD, "c2_digests.json")))
    assert g["format"] == "BFP[8|8]{16}(SN)" and g["shape"] == [4096, 4096] and len(g["slots"]) == 160 and len(g["kinds"]) == 4
    assert sorted(int(k) for k in g["slots"]) == [1000 * r + s for r in range(8) for s in range(20)]
    for key, kind, seed in (("kinds", "normal", 0), ("slots", "heavy", 1007), ("slots", "heavy", 7019)):
        e = g[key][kind if key == "kinds" else str(seed)]
        x = make_chunked(kind, (4096, 4096), seed, torch.bfloat16)
        assert sha256_bits(x) == e["input_sha256"]
        assert sha256_bits(oracle.bfp_cast(x, 8, 16).to(torch.bfloat16)) == e["output_sha256"]


def test_stochastic_bfp_draws_are_unbiased_and_pairwise_uncorrelated(oracle):
    """The BFP cast's stochastic stream (round 5: oracle.c bfp_rnd == csrc/common.hpp bfp_rnd, one hash per 8 elements expanded per pair;
    the reference's is an unseeded global mt19937, so parity with IT is statistical only -- SURVEY Appendix C).  What a stochastic
    rounding needs of its draws: P(round up) == the dropped fraction for every element position of the group of 8 (unbiased), and
    the decisions of neighbouring elements -- the two halves of one word, two words of one hash -- uncorrelated."""
    import torch

    n_blocks, B, wl = 1 << 16, 16, 8
    # block maximum 1.0 -> exponent 0, quantum 2^-6; every other element sits a fraction f of a quantum above a code
    fracs = torch.tensor([0.1, 0.25, 0.5, 0.7, 0.9, 0.33, 0.05, 0.95, 0.6, 0.4, 0.8, 0.2, 0.15, 0.85, 0.45])
    q = 2.0 ** -6
    x = torch.empty(n_blocks, B)
    x[:, 0] = 1.0
    x[:, 1:] = (3.0 + fracs) * q
    up = torch.zeros(B - 1)
    ups = []
    for seed in (1, 2, 3):
        y = oracle.bfp_cast(x, wl, B, -1, rounding="stochastic", seed=seed)
        d = ((y[:, 1:] / q).round() - 3.0)                    # 0 = rounded down, 1 = up
        assert bool(((d == 0) | (d == 1)).all())
        ups.append(d)
        up += d.mean(0)
    up /= 3
    # 3 x 65536 draws per position: standard error of a frequency <= 0.5 / sqrt(196608) = 0.0011
    assert float((up - fracs).abs().max()) < 0.006, (up - fracs)
    d = torch.cat(ups, 0)
    c = torch.corrcoef(d.t())
    off = c - torch.eye(B - 1)
    assert float(off.abs().max()) < 0.02, float(off.abs().max())   # (independent draws: |r| ~ 1 / sqrt(196608) = 0.002)


def test_llama_shard_digests_are_reproduced_by_the_oracle(oracle):
    """tests/golden/llama_shard_digests.json (oracle/gen_golden_r6.py): SHA-256 of the reference's Sparsify(2:4) -> CastTo(BFP16_64) and
    CastTo(BFP16_16) outputs per eighth of the rows of the seven Llama-3-8B layer weights bench.py's llama-shard workload generates.
    One piece of one small weight here: generated as a SHARD (start offset), cast by the oracle, compared with the committed digests."""
    import json

    from _data import make_chunked, sha256_bits

    g = json.load(open(os.path.join(GOLD, "llama_shard_digests.json")))
    assert sorted(g["tensors"]) == sorted(["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"])
    assert all(len(t["pieces"]) == 8 for t in g["tensors"].values())
    t = g["tensors"]["k_proj"]
    k, piece, cols = 5, t["rows"] // 8, t["cols"]
    w = make_chunked("heavy", (piece, cols), t["w_seed"], torch.bfloat16, start=k * piece * cols)
    sc = make_chunked("uniform", (piece, cols), t["score_seed"], torch.bfloat16, start=k * piece * cols)
    e = t["pieces"][k]
    assert e["rows"] == [k * piece, (k + 1) * piece] and sha256_bits(w) == e["w_sha256"] and sha256_bits(sc) == e["score_sha256"]
    assert sha256_bits(oracle.bfp_cast(oracle.sparsify(w, sc, 2, 4), 8, 64).to(torch.bfloat16)) == e["hypernet_sha256"]
    assert sha256_bits(oracle.bfp_cast(w, 8, 16).to(torch.bfloat16)) == e["bfp_sha256"]
