"""-m gpu: every launch-geometry branch of the size-dependent plans computes the same values.

Round 3 made the tile geometry of the flat-stream kernels a function of the tensor size (csrc/stream.hpp launch_stream: 256 x 1 up
to 2 MiB, 256 x 2 up to 12 MiB, 256 x 4 up to 20 MiB, the op's own geometry up to 32 MiB -- 64 x 16, 128 x 16, 64 x 8, 512 x 4, 512 x 16 ...
-- and 256 x 2 beyond), gave the per-channel kernel along the contiguous dim 4 / 8 / 16 rows per lane by size (csrc/elementwise.hip
lastdim_plan), the norm workgroup kernel 8 rows per pass for 26-32 MiB tensors and softmax one pass per workgroup (csrc/approx.hip).
The oracle-parity tests run at small sizes (the 256 x 1 branch) and at 4096 x 4096; this file closes the gap with a size-independent
property of these ops: they are per element (or per row / per block / per quantisation group), so the result on a big tensor must
equal, bit for bit, the results on its row slabs -- each small enough to take the 256 x 1 branch the oracle tests pin.  Sizes are chosen to hit every
branch, including a tensor one row past the 32 MiB boundary and row counts that leave partial tiles."""
import pytest
import torch

from _data import mismatches_nan_aware, bits_equal, make

pytestmark = pytest.mark.gpu
BF16, F32, F16 = torch.bfloat16, torch.float32, torch.float16

# rows of a [rows, 4096] 16-bit tensor (float32: [rows, 2048], the same bytes): 2.3 MiB (256 x 2), 12.5 MiB (256 x 4), 20.3 MiB and 32 MiB
# (the op's own geometry), 32.8 MiB and 48 MiB (256 x 2 again)
ROWS = [300, 1600, 2600, 4096, 4200, 6144]
SLAB = 128  # rows per reference slab: 1 MiB -> the 256 x 1 branch; a multiple of every group / block size used below


def _input(rows, dtype, seed):
    cols = 2048 if dtype == F32 else 4096
    x = make("heavy", (rows, cols), seed=seed, dtype=torch.float32).clamp(-3e4, 3e4)
    x.view(-1)[:: 4099] = 0.0
    x.view(-1)[5:: 8191] = -0.0
    return x.to(dtype)


def _slabs(fn, x, per_slab_args=None):
    outs = []
    for i, r0 in enumerate(range(0, x.shape[0], SLAB)):
        xs = x[r0:r0 + SLAB].contiguous()
        outs.append(fn(xs, i) if per_slab_args else fn(xs))
    return torch.cat(outs, 0)


def _check(tag, whole, parts):
    bad = bits_equal(whole, parts)
    assert bad == 0, f"{tag}: {bad} elements differ between the whole-tensor launch and its 128-row slabs"


@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("dtype", [BF16, F16, F32], ids=["bf16", "f16", "f32"])
def test_per_element_casts_and_functions(dmx, cuda, rows, dtype):
    ops = dmx.ops
    x = _input(rows, dtype, seed=rows).to(cuda)
    fp16 = dmx.Format.from_shorthand("FP[1|5|10,15](FN)")
    cases = {
        "float_qdq E4M3": lambda t: ops.float_qdq(t, 3, 4, 7, False),
        "float_qdq FP16(FN)": lambda t: ops.float_qdq(t, 10, 5, 15, True),
        "float_qdq E5M2 stochastic-free up": lambda t: ops.float_qdq(t, 2, 5, 15, False, rounding="up"),
        "fixed_qdq INT8": lambda t: ops.fixed_qdq(t, 8, 0),
        "fixed_qdq XP[8,4] unclamped": lambda t: ops.fixed_qdq(t, 8, 4, clamp=False),
        "gelu": lambda t: ops.gelu(t),
        "gelu tanh": lambda t: ops.gelu(t, approximate="tanh"),
        "silu": lambda t: ops.silu(t),
        "quick_gelu": lambda t: ops.quick_gelu(t),
        "unary_cast gelu FLOAT16": lambda t: ops.unary_cast(t, "gelu", fp16, fp16),
        "unary_cast silu FLOAT16": lambda t: ops.unary_cast(t, "silu", fp16, fp16),
        "unary_cast quick_gelu FLOAT16": lambda t: ops.unary_cast(t, "quick_gelu", fp16, fp16),
        "relu_cast FLOAT16": lambda t: ops.relu_cast(t, fp16, fp16),
    }
    for tag, fn in cases.items():
        whole = fn(x)
        assert whole is not None, tag
        _check(f"{tag} {dtype} rows={rows}", whole, _slabs(fn, x))


@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("dtype", [BF16, F16, F32], ids=["bf16", "f16", "f32"])
def test_every_size_class_directly_against_the_oracle(dmx, cuda, oracle, rows, dtype):
    """the mid-size launch geometries of the non-BFP ops were pinned only through the slab identity above (HIP vs
    HIP).  Here every size class of float_qdq, fixed_qdq (plain and affine: per group of rows, per row, per column), scale_channels,
    SBFP and MXFP is compared with the CPU oracle itself, bit for bit, on the whole tensor."""
    ops = dmx.ops
    xh = _input(rows, dtype, seed=3 * rows + 1)
    x = xh.to(cuda)
    cols = xh.shape[1]

    def eq(tag, got, want):
        bad = bits_equal(got, want.to(dtype).contiguous())
        assert bad == 0, f"{tag} {dtype} rows={rows}: {bad} elements differ from the oracle"

    eq("float_qdq E4M3", ops.float_qdq(x, 3, 4, 7, False), oracle.floating_point_cast(xh, 3, 4, 7, False))
    eq("float_qdq FP16(FN)", ops.float_qdq(x, 10, 5, 15, True), oracle.floating_point_cast(xh, 10, 5, 15, True))
    eq("float_qdq BF16(FN)", ops.float_qdq(x, 7, 8, 127, True), oracle.floating_point_cast(xh, 7, 8, 127, True))
    eq("fixed_qdq INT8", ops.fixed_qdq(x, 8, 0), oracle.fixed_point_cast(xh, 8, 0, True, True))
    eq("fixed_qdq XP[8,4] unclamped", ops.fixed_qdq(x, 8, 4, clamp=False), oracle.fixed_point_cast(xh, 8, 4, False, True))
    g = torch.Generator().manual_seed(rows)
    n_groups = -(-rows // 128)
    sc_g, zp_g = torch.rand(n_groups, generator=g) * 0.2 + 0.01, torch.randint(-5, 6, (n_groups,), generator=g)
    sc_r, zp_r = torch.rand(rows, generator=g) * 0.2 + 0.01, torch.randint(-5, 6, (rows,), generator=g)
    sc_c, zp_c = torch.rand(cols, generator=g) * 0.2 + 0.01, torch.randint(-5, 6, (cols,), generator=g)
    sc_c[3], sc_c[cols - 1] = 3e-7, 2.5e6            # outside the reciprocal form's range: the IEEE-division redo
    eq("INT8 affine, groups of 128 rows", ops.fixed_qdq(x, 8, 0, scale=sc_g.to(cuda), zero_point=zp_g.to(cuda), ch_axis=0, group_size=128),
       oracle.fixed_point_affine_cast(xh, 8, 0, True, True, sc_g, zp_g, ch_axis=0, group_size=128))
    eq("INT8 affine per row", ops.fixed_qdq(x, 8, 0, scale=sc_r.to(cuda), zero_point=zp_r.to(cuda), ch_axis=0),
       oracle.fixed_point_affine_cast(xh, 8, 0, True, True, sc_r, zp_r, ch_axis=0))
    eq("INT8 affine per column", ops.fixed_qdq(x, 8, 0, scale=sc_c.to(cuda), zero_point=zp_c.to(cuda), ch_axis=1),
       oracle.fixed_point_affine_cast(xh, 8, 0, True, True, sc_c, zp_c, ch_axis=1))
    eq("INT4 asymmetric affine per tensor", ops.fixed_qdq(x, 4, 0, True, False, scale=sc_c[:1].to(cuda), zero_point=zp_c[:1].to(cuda)),
       oracle.fixed_point_affine_cast(xh, 4, 0, True, False, sc_c[:1], zp_c[:1]))
    # SmoothQuant's x / s and w * s (numerical/smoothquant.py:255-283): IEEE float32 operations rounded once to the tensor dtype
    eq("scale_channels divide", ops.scale_channels(x, sc_c.to(cuda), 1, True, out_dtype=dtype), xh.float() / sc_c)
    eq("scale_channels multiply", ops.scale_channels(x, sc_c.to(cuda), 1, False, out_dtype=dtype), xh.float() * sc_c)
    if dtype != F16:
        eq("MXFP8[E4M3]{32}", ops.mxfp_qdq(x, 3, 4, 32), oracle.mxfp_cast(xh, 3, 4, 32))
        eq("MXFP4[E2M1]{32}", ops.mxfp_qdq(x, 1, 2, 32), oracle.mxfp_cast(xh, 1, 2, 32))
        eq("SBFP12_16", ops.sbfp_qdq(x, 4, 16, 4, 4, 7), oracle.sbfp_cast(xh, 4, 16, 4, 4, 7))
    eq("BFP[8|8]{64} asym", ops.bfp_qdq(x, 8, 64, symmetric=False), oracle.bfp_cast(xh, 8, 64, -1, False))


@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "f32"])
def test_affine_integer_casts(dmx, cuda, rows, dtype):
    """INT8 with a scale per group of 128 rows (stream kernel, one scale per tile), per row (stream kernel, channel walker), per
    column (lastdim kernel: 4 / 8 / 16 rows per lane by size) and SmoothQuant's per-column divide / multiply."""
    ops = dmx.ops
    x = _input(rows, dtype, seed=7 * rows).to(cuda)
    cols = x.shape[1]
    g = torch.Generator().manual_seed(rows)
    n_groups = -(-rows // 128)
    sc_g = (torch.rand(n_groups, generator=g) * 0.2 + 0.01).to(cuda)
    zp_g = torch.randint(-5, 6, (n_groups,), generator=g).to(cuda)
    sc_r = (torch.rand(rows, generator=g) * 0.2 + 0.01).to(cuda)
    zp_r = torch.randint(-5, 6, (rows,), generator=g).to(cuda)
    sc_c = (torch.rand(cols, generator=g) * 0.2 + 0.01)
    sc_c[3] = 3e-7            # outside the reciprocal form's range: the lane redoes its vector with the IEEE division
    sc_c[cols - 1] = 2.5e6
    sc_c = sc_c.to(cuda)
    zp_c = torch.randint(-5, 6, (cols,), generator=g).to(cuda)

    whole = ops.fixed_qdq(x, 8, 0, scale=sc_g, zero_point=zp_g, ch_axis=0, group_size=128)
    parts = _slabs(lambda t, i: ops.fixed_qdq(t, 8, 0, scale=sc_g[i:i + 1], zero_point=zp_g[i:i + 1], ch_axis=0, group_size=128), x, True)
    _check(f"INT8 group 128 {dtype} rows={rows}", whole, parts)

    whole = ops.fixed_qdq(x, 8, 0, scale=sc_r, zero_point=zp_r, ch_axis=0)
    parts = _slabs(lambda t, i: ops.fixed_qdq(t, 8, 0, scale=sc_r[i * SLAB:(i + 1) * SLAB], zero_point=zp_r[i * SLAB:(i + 1) * SLAB], ch_axis=0), x, True)
    _check(f"INT8 per row {dtype} rows={rows}", whole, parts)

    whole = ops.fixed_qdq(x, 8, 0, scale=sc_c, zero_point=zp_c, ch_axis=1)
    _check(f"INT8 per column {dtype} rows={rows}", whole, _slabs(lambda t: ops.fixed_qdq(t, 8, 0, scale=sc_c, zero_point=zp_c, ch_axis=1), x))
    # ... against the per-element definition as well (cast.py:278-296 on the GPU's own torch ops; IEEE division): the lastdim kernel is new
    xf = x.float()
    q = torch.clamp(torch.round((xf / sc_c + zp_c) + 0.5 - 0.5), -127, 127)   # (symmetric=True: [-127, 127])
    # (sim_helper's (a + 0.5) - 0.5 pre-step equals torch.round's half-even except where the fp32 add rounds; compare loosely here,
    #  the bit-exact statement is the slab identity above plus the small-size oracle tests)
    ref = ((q - zp_c) * sc_c).to(dtype)
    assert bool(((whole.float() - ref.float()).abs() <= sc_c * 1.6).all())

    for divide in (True, False):
        whole = ops.scale_channels(x, sc_c, 1, divide)
        _check(f"scale_channels divide={divide} {dtype} rows={rows}", whole, _slabs(lambda t: ops.scale_channels(t, sc_c, 1, divide), x))


@pytest.mark.parametrize("rows", ROWS)
def test_block_formats(dmx, cuda, rows):
    ops = dmx.ops
    x = _input(rows, BF16, seed=11 * rows).to(cuda)
    cases = {
        "MXFP8[E4M3]{32}": lambda t: ops.mxfp_qdq(t, 3, 4, 32),
        "MXFP4[E2M1]{32}": lambda t: ops.mxfp_qdq(t, 1, 2, 32),
        "SBFP12_16": lambda t: ops.sbfp_qdq(t, 4, 16, 4, 4, 7, False),
        "BFP[8|8]{16}": lambda t: ops.bfp_qdq(t, 8, 16),
        "BFP[8|8]{64} asym": lambda t: ops.bfp_qdq(t, 8, 64, symmetric=False),
        "BFP[8|8]{64} along rows": lambda t: ops.bfp_qdq(t, 8, 64, block_dim=0),
    }
    for tag, fn in cases.items():
        _check(f"{tag} rows={rows}", fn(x), _slabs(fn, x))


@pytest.mark.parametrize("rows", [40, 300, 1600, 4096, 4200])
@pytest.mark.parametrize("dtype", [BF16, F16, F32], ids=["bf16", "f16", "f32"])
def test_dense_smoothquant_weight_path(dmx, cuda, rows, dtype):
    """w * s[column] -> BFP in one launch (csrc/hypernet.hip HnLastOp on the lastdim kernel: 4 / 8 / 16 rows per lane by size) == the two
    library ops it fuses (scale_channels, then bfp_qdq: both pinned to the oracle / the reference elsewhere), and == its own row slabs;
    symmetric and asymmetric codes, block sizes with 2, 8 and 64 lanes per block, a widening output."""
    ops = dmx.ops
    x = _input(rows, dtype, seed=17 * rows).to(cuda)
    cols = x.shape[1]
    sc = (torch.rand(cols, generator=torch.Generator().manual_seed(rows)) * 4 + 0.25).to(cuda)
    for B, sym, out_dtype in ((64, True, None), (16, False, None), (512 if dtype != F32 else 256, True, None), (64, True, F32)):
        if out_dtype == dtype:
            continue
        fn = lambda t: ops.weight_hypernet(t, 8, B, sym, sq_scale=sc, out_dtype=out_dtype)
        whole = fn(x)
        assert whole is not None, (B, sym, out_dtype)
        scaled = ops.scale_channels(x, sc, 1, False)             # w * s, rounded to the weight dtype
        chain = ops.bfp_qdq(scaled, 8, B, symmetric=sym)
        if out_dtype is not None:
            chain = chain.to(out_dtype)
        _check(f"dense SQ + BFP{B} sym={sym} {dtype}->{out_dtype} rows={rows} vs the two-op chain", whole, chain)
        _check(f"dense SQ + BFP{B} sym={sym} {dtype}->{out_dtype} rows={rows} vs slabs", whole, _slabs(fn, x))


@pytest.mark.parametrize("rows", [1600, 3584, 4096, 4200])
@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "f32"])
def test_row_functions(dmx, cuda, rows, dtype):
    """RMSNorm / LayerNorm rows of 4096 16-bit elements (workgroup-per-row kernel: 2 rows per iteration on a persistent grid, 8 rows per
    pass at 26-32 MiB), rows of 768 (wave kernel) and softmax rows of 1500 (one pass per workgroup), plain and as fused modules."""
    ops = dmx.ops
    fp16 = dmx.Format.from_shorthand("FP[1|5|10,15](FN)")
    x = (_input(rows, dtype, seed=13 * rows).float().clamp(-30, 30)).to(dtype).to(cuda)
    cols = x.shape[1]
    w = (torch.rand(cols, generator=torch.Generator().manual_seed(1)) + 0.5).to(dtype).to(cuda)
    b = (torch.rand(cols, generator=torch.Generator().manual_seed(2)) - 0.5).to(dtype).to(cuda)
    cases = {
        "rmsnorm": lambda t: ops.rmsnorm(t, cols, w, 1e-6),
        "layernorm": lambda t: ops.layernorm(t, cols, w, b),
        "rmsnorm_cast": lambda t: ops.rmsnorm_cast(t, cols, w, 1e-6, fp16, fp16),
        "layernorm_cast": lambda t: ops.layernorm_cast(t, cols, w, b, 1e-5, fp16, fp16),
        "softmax": lambda t: ops.softmax(t),
    }
    for tag, fn in cases.items():
        whole = fn(x)
        assert whole is not None, tag
        _check(f"{tag} {dtype} {rows}x{cols}", whole, _slabs(fn, x))
    # short rows: the same elements viewed as rows of 768 / 1500
    n = x.numel()
    for c, names in ((768, ("layernorm", "layernorm_cast")), (1500, ("softmax", "softmax_cast"))):
        r = n // c
        y = x.view(-1)[: r * c].view(r, c)
        wc, bc = w[:c].contiguous(), b[:c].contiguous()
        fns = {
            "layernorm": lambda t: ops.layernorm(t, c, wc, bc),
            "layernorm_cast": lambda t: ops.layernorm_cast(t, c, wc, bc, 1e-5, fp16, fp16),
            "softmax": lambda t: ops.softmax(t),
            "softmax_cast": lambda t: ops.softmax_cast(t, -1, fp16, fp16),
        }
        for nm in names:
            whole = fns[nm](y)
            assert whole is not None, nm
            parts = torch.cat([fns[nm](y[r0:r0 + 512].contiguous()) for r0 in range(0, r, 512)], 0)
            _check(f"{nm} {dtype} {r}x{c}", whole, parts)


def test_per_channel_kernel_equals_the_channel_walker_on_random_shapes(dmx, cuda):
    """The per-channel-along-the-last-dim kernel (csrc/lastdim.hpp: column strips, several rows side by side in a workgroup when rows are
    short, 4 / 8 / 16 rows per lane, a clamped tail workgroup) against the flat-stream kernel's channel walker, which the same call takes
    when the scale table is not 16-byte aligned (elementwise.hip pick_mode): 60 random [rows, C] shapes, C any multiple of the lane
    vector, bf16 / fp16 / float32, INT8 with zero points, INT4 symmetric, and SmoothQuant's divide / multiply -- bit for bit."""
    ops = dmx.ops
    g = torch.Generator().manual_seed(2024)
    for it in range(60):
        dtype = (BF16, F16, F32)[it % 3]
        epl = 4 if dtype == F32 else 8
        C = int(torch.randint(1, 700, (1,), generator=g)) * epl
        rows = int(torch.randint(1, 3000, (1,), generator=g))
        x = (make("heavy", (rows, C), seed=it, dtype=torch.float32).clamp(-1e4, 1e4)).to(dtype).to(cuda)
        buf = (torch.rand(C + 4, generator=g) * 0.3 + 0.01).to(cuda)
        zbuf = torch.randint(-7, 8, (C + 2,), generator=g).to(cuda)
        sc_al, sc_un = buf[4:].clone(), buf[1:C + 1]        # same values; the second view starts 4 bytes into the allocation
        sc_un.copy_(sc_al)
        zp_al, zp_un = zbuf[2:].clone(), zbuf[1:C + 1]      # int64: 8 bytes into the allocation
        zp_un.copy_(zp_al)
        assert sc_al.data_ptr() % 16 == 0 and sc_un.data_ptr() % 16 != 0
        for prec in (8, 4):
            a = ops.fixed_qdq(x, prec, 0, scale=sc_al, zero_point=zp_al, ch_axis=1)
            b = ops.fixed_qdq(x, prec, 0, scale=sc_un, zero_point=zp_un, ch_axis=1)
            _check(f"INT{prec} per column {dtype} [{rows}, {C}]", a, b)
        for divide in (True, False):
            a = ops.scale_channels(x, sc_al, 1, divide)
            b = ops.scale_channels(x, sc_un, 1, divide)
            _check(f"scale_channels divide={divide} {dtype} [{rows}, {C}]", a, b)
        if dtype != F32:
            a = ops.scale_channels(x, sc_al, 1, True, out_dtype=F32)
            b = ops.scale_channels(x, sc_un, 1, True, out_dtype=F32)
            _check(f"scale_channels -> float32 {dtype} [{rows}, {C}]", a, b)


@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "f32"])
def test_dense_smoothquant_weight_path_on_narrow_rows(dmx, cuda, dtype):
    """Rows shorter than a workgroup (several rows side by side, lanes left over when 256 is not a multiple of the row's vectors) and
    blocks of 2 .. 32 lanes: w * s -> BFP on the lastdim kernel == scale_channels then bfp_qdq, bit for bit."""
    ops = dmx.ops
    for cols, Bs in ((768, (16, 64, 256)), (1536, (64, 512)), (192, (8, 64)), (64, (16, 64)), (4160, (64,))):
        for rows in (1, 7, 333, 2050):
            x = make("heavy", (rows, cols), seed=rows + cols, dtype=torch.float32).clamp(-3e4, 3e4).to(dtype).to(cuda)
            sc = (torch.rand(cols, generator=torch.Generator().manual_seed(cols)) * 4 + 0.25).to(cuda)
            for B in Bs:
                if dtype == F32 and B > 256:
                    continue
                for sym in (True, False):
                    whole = ops.weight_hypernet(x, 8, B, sym, sq_scale=sc)
                    assert whole is not None, (cols, B)
                    chain = ops.bfp_qdq(ops.scale_channels(x, sc, 1, False), 8, B, symmetric=sym)
                    _check(f"dense SQ + BFP{B} sym={sym} {dtype} [{rows}, {cols}]", whole, chain)


@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "f32"])
def test_packed_bfp_geometry_branches(dmx, cuda, dtype):
    """bfp_pack / bfp_unpack choose their tile geometry and load / store width by size (csrc/bfp_pack.hip: two or four vectors per lane,
    16-byte code stores through a neighbour-lane exchange when the vector count is even, exponents through LDS; unpack: 8-byte loads x 4,
    16-byte loads x 8 up to 32 MiB of output, x 2 beyond, 8-byte loads x 8 when the vector count is odd): codes, exponents and unpacked
    values of the whole tensor equal those of its 128-row slabs (small: the branch the oracle test pins), and the round trip equals the cast."""
    ops = dmx.ops
    cols = 4096
    for rows, L, B in [(300, cols, 16), (1600, cols, 64), (4096, cols, 16), (4200, cols, 64), (1537, 4104, 8), (130, 4104, 8), (300, cols, 512), (2000, cols, 512),
                       (300, cols, 256)]:
        x = make("heavy", (rows, L), seed=rows, dtype=torch.float32).clamp(-3e4, 3e4)
        x.view(-1)[:: 4099] = 0.0
        x[1, :B] = 0.0
        x = x.to(dtype).to(cuda)
        m, e = ops.bfp_pack(x, 8, B, True)
        ms, es = [], []
        for r0 in range(0, rows, SLAB):
            a, b = ops.bfp_pack(x[r0:r0 + SLAB].clone(), 8, B, True)
            ms.append(a)
            es.append(b)
        assert torch.equal(m, torch.cat(ms, 0)) and torch.equal(e, torch.cat(es, 0)), (rows, L, B)
        for od in (BF16, F32):
            y = ops.bfp_unpack(m, e, 8, B, od)
            ys = torch.cat([ops.bfp_unpack(m[r0:r0 + SLAB].clone(), e[r0:r0 + SLAB].clone(), 8, B, od) for r0 in range(0, rows, SLAB)], 0)
            _check(f"bfp_unpack {rows}x{L} B={B} -> {od}", y, ys)
        ok = ((e > 0) & (e < 255)).repeat_interleave(B, dim=-1)
        y, q = ops.bfp_unpack(m, e, 8, B, torch.float32), ops.bfp_qdq(x, 8, B, out_dtype=torch.float32)
        assert torch.equal(y[ok].view(torch.int32), q[ok].view(torch.int32)), (rows, L, B)


def test_float32_nm_mask_kernel_equals_the_general_vector_kernel(dmx, cuda):
    """float32 score -> float32 mask takes its own kernel (csrc/nm_mask.hip nm_mask_f32_kernel: one 16-byte vector per lane and slot, M = 8
    groups assembled by a neighbour-lane exchange); the same score with a bfloat16 mask takes the general vector kernel the oracle tests pin:
    the masks must agree, with ties, +-0, NaN and Inf in the groups, on sizes with partial last tiles."""
    ops = dmx.ops
    for rows, cols in [(300, 4096), (4096, 4096), (5, 48), (1, 16), (1030, 1040)]:
        s = make("heavy", (rows, cols), seed=rows + cols, dtype=torch.float32)
        s.view(-1)[::7] = 0.0
        s.view(-1)[3::11] = -0.0
        s.view(-1)[5::13] = 1.5            # ties
        s.view(-1)[1::97] = float("nan")
        s.view(-1)[2::101] = float("inf")
        s.view(-1)[4::103] = -float("inf")
        s = s.to(cuda)
        for K, M in [(1, 2), (2, 4), (1, 4), (3, 4), (4, 8), (2, 8), (7, 8), (8, 8)]:
            m32 = ops.nm_mask(s, K, M)
            m16 = ops.nm_mask(s, K, M, mask_dtype=torch.bfloat16)
            assert m32.dtype == torch.float32 and torch.equal(m32, m16.float()), (rows, cols, K, M)
            assert int(m32.sum()) == s.numel() // M * K


@pytest.mark.parametrize("rows", [300, 1600, 4096, 4200])
@pytest.mark.parametrize("dtype", [BF16, F16], ids=["bf16", "f16"])
def test_widening_outputs_take_the_same_values_on_every_geometry(dmx, cuda, rows, dtype):
    """16-bit in, float32 out: the flat-stream and per-channel kernels give a lane 4 elements (8-byte loads, ONE 16-byte store: every
    store instruction covers whole lines) instead of 8 -- csrc/stream.hpp / lastdim.hpp IVB.  Whole tensor == its 128-row slabs (the
    small branch the oracle tests pin, tests/test_gpu_elementwise.py), and == the same-dtype result where that is exact."""
    ops = dmx.ops
    x = _input(rows, dtype, seed=rows + 1).to(cuda)
    C = x.shape[1]
    sc = (torch.rand(C, device=cuda) * 0.05 + 0.01)
    zp = torch.randint(-3, 4, (C,), device=cuda, dtype=torch.int64)
    sq = torch.rand(C, device=cuda) + 0.5
    cases = {
        "float_qdq E4M3 -> f32": lambda t: ops.float_qdq(t, 3, 4, 7, False, out_dtype=F32),
        "float_qdq FP16(FN) -> f32": lambda t: ops.float_qdq(t, 10, 5, 15, True, out_dtype=F32),
        "fixed_qdq INT8 -> f32": lambda t: ops.fixed_qdq(t, 8, 0, out_dtype=F32),
        "fixed_qdq INT8 per channel (last dim) -> f32": lambda t: ops.fixed_qdq(t, 8, 0, scale=sc, zero_point=zp, ch_axis=-1, out_dtype=F32),
        "scale_channels x / s -> f32": lambda t: ops.scale_channels(t, sq, -1, True, out_dtype=F32),
        "scale_channels x * s -> f32": lambda t: ops.scale_channels(t, sq, -1, False, out_dtype=F32),
    }
    for tag, fn in cases.items():
        whole = fn(x)
        assert whole.dtype == F32
        _check(f"{tag} {dtype} rows={rows}", whole, _slabs(fn, x))
    # the casts' values are representable in the input dtype here: the float32 output is the same-dtype output widened
    assert bits_equal(ops.float_qdq(x, 10, 5, 15, True, out_dtype=F32), ops.float_qdq(x, 10, 5, 15, True).float()) == 0
    assert bits_equal(ops.scale_channels(x, sq, -1, True, out_dtype=F32), x.float() / sq) == 0


@pytest.mark.parametrize("rows", [600, 1400, 2000, 2500, 3072, 4096, 4200, 4500, 4700])
@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "f32"])
def test_hot_kernel_tile_plans(dmx, cuda, oracle, rows, dtype):
    """every size class of csrc/common.hpp rows_plan (512x1, 128x2, 512x4, 128x8, 512x16, round 4: 512x17 and 512x18 in one round, 512x2; float32 tensors reach the classes at half
    the rows) for the hot BFP kernel, its widening / down / asymmetric builds and the range-only FLOAT16 cast that shares the plan:
    whole tensor == 128-row slabs, and (round 5) every one of them directly against the CPU ORACLE as well -- the slab identity alone
    compares the library with itself."""
    ops = dmx.ops
    xh = _input(rows, dtype, seed=7 * rows)
    x = xh.to(cuda)
    cases = {
        "BFP[8|8]{16}": (lambda t: ops.bfp_qdq(t, 8, 16), lambda t: oracle.bfp_cast(t, 8, 16, -1).to(dtype)),
        "BFP[8|8]{64}": (lambda t: ops.bfp_qdq(t, 8, 64), lambda t: oracle.bfp_cast(t, 8, 64, -1).to(dtype)),
        "BFP[8|8]{64} asym": (lambda t: ops.bfp_qdq(t, 8, 64, symmetric=False), lambda t: oracle.bfp_cast(t, 8, 64, -1, symmetric=False).to(dtype)),
        "BFP[8|8]{32} down": (lambda t: ops.bfp_qdq(t, 8, 32, rounding="down"), lambda t: oracle.bfp_cast(t, 8, 32, -1, rounding="down").to(dtype)),
        "BFP[8|8]{64} -> f32": (lambda t: ops.bfp_qdq(t, 8, 64, out_dtype=F32), lambda t: oracle.bfp_cast(t, 8, 64, -1)),
        "FLOAT16 cast": (lambda t: ops.float_qdq(t, 10, 5, 15, True), lambda t: oracle.float_quantize(t, 10, 5, 15, True).to(dtype)),
    }
    for tag, (fn, ref) in cases.items():
        whole = fn(x)
        _check(f"{tag} {dtype} rows={rows}", whole, _slabs(fn, x))
        bad = bits_equal(whole, ref(xh))
        assert bad == 0, f"{tag} {dtype} rows={rows}: {bad} elements differ from the oracle"


@pytest.mark.parametrize("rows", [2561, 2700, 2816, 2817, 3100, 3400, 3700, 3950, 4097, 4100, 4353, 4608, 4609, 4864, 4865, 5000, 5120, 5121])
@pytest.mark.parametrize("dtype", [BF16, F16], ids=["bf16", "f16"])
def test_exact_depth_one_round_plans_against_the_oracle(dmx, cuda, oracle, rows, dtype):
    """Round 4: 20-40 MiB of a 16-bit tensor run as ONE round of <= 256 workgroups whose depth is exactly what that takes -- 11 .. 20
    vectors per lane (csrc/common.hpp rows_plan; 19 and 20 through the compact kernel, results in place of the raw vectors) -- and the last, partial tile of any tensor runs on the
    same schedule with predicated loads and stores (bfp_rows_tile_partial; it used to run vector by vector).  Both sides of class
    boundaries, nearly empty and nearly full last tiles, directly against the CPU oracle, for both 16-bit dtypes (the float16 builds
    are distinct code objects: round 4 skipped 12 of their sizes).  Round 5: the builds that do NOT take these plans -- asymmetric,
    widening (16-bit -> float32), rounding down / up (compile-time modes since round 5, on the 512 x 4 ... 512 x 16 plans) and
    stochastic -- are compared with the ORACLE once per depth class as well (round 4: the slab identity, HIP against HIP)."""
    ops = dmx.ops
    xh = _input(rows, dtype, seed=11 * rows)
    x = xh.to(cuda)
    for B in (16, 128):
        bad = bits_equal(ops.bfp_qdq(x, 8, B), oracle.bfp_cast(xh, 8, B, -1).to(dtype))
        assert bad == 0, f"BFP[8|8]{{{B}}} {dtype} rows={rows}: {bad} elements differ from the oracle"
    if rows in (2700, 2817, 3100, 3400, 3700, 3950, 4100, 4353, 4609, 5000):   # one size per depth class 11 .. 20
        bad = bits_equal(ops.bfp_qdq(x, 4, 32), oracle.bfp_cast(xh, 4, 32, -1).to(dtype))
        assert bad == 0, f"BFP[4|8]{{32}} {dtype} rows={rows}: {bad} elements differ from the oracle"
        others = {
            "asym": (lambda t: ops.bfp_qdq(t, 8, 64, symmetric=False), lambda t: oracle.bfp_cast(t, 8, 64, -1, symmetric=False).to(dtype)),
            "-> f32": (lambda t: ops.bfp_qdq(t, 8, 64, out_dtype=F32), lambda t: oracle.bfp_cast(t, 8, 64, -1)),
            "down": (lambda t: ops.bfp_qdq(t, 8, 32, rounding="down"), lambda t: oracle.bfp_cast(t, 8, 32, -1, rounding="down").to(dtype)),
            "up": (lambda t: ops.bfp_qdq(t, 8, 16, rounding="up"), lambda t: oracle.bfp_cast(t, 8, 16, -1, rounding="up").to(dtype)),
            "stochastic": (lambda t: ops.bfp_qdq(t, 8, 16, rounding="stochastic", seed=rows),
                           lambda t: oracle.bfp_cast(t, 8, 16, -1, rounding="stochastic", seed=rows).to(dtype)),
        }
        for tag, (fn, ref) in others.items():
            bad = bits_equal(fn(x), ref(xh))
            assert bad == 0, f"{tag} {dtype} rows={rows}: {bad} elements differ from the oracle"


@pytest.mark.parametrize("rows", [2700, 3001, 3400, 3840])
def test_exact_depth_plans_float32_against_the_oracle(dmx, cuda, oracle, rows):
    """the exact-depth plans of the float32 -> float32 build (11 .. 16 vectors of 4 elements per lane): [rows, 2048] float32, whole tiles
    and a partial last one, BFP16 and a 16-bit-mantissa format (double rounding build)."""
    xh = _input(rows, F32, seed=13 * rows)
    x = xh.to(cuda)
    for wl, B in ((8, 16), (8, 64), (16, 32)):
        bad = bits_equal(dmx.ops.bfp_qdq(x, wl, B), oracle.bfp_cast(xh, wl, B, -1))
        assert bad == 0, f"BFP[{wl}|8]{{{B}}} float32 rows={rows}: {bad} elements differ from the oracle"


def test_partial_last_tile_special_values_and_the_literal_redo(dmx, cuda, oracle):
    """The predicated partial tile with blocks that cannot take the magic-add path (denormal block maxima, NaN) INSIDE it and just
    before it: the rare literal redo re-reads predicated too."""
    rows = 4300                                           # 512 x 17 tiles: 252 full + one that is 94 % full
    xh = _input(rows, BF16, seed=5)
    flat = xh.view(-1)
    n = flat.numel()
    tail0 = (n // (512 * 17 * 8)) * (512 * 17 * 8)        # first element of the partial tile
    for off in (tail0 + 3, tail0 + 8 * 512 * 5 + 64, n - 20, tail0 - 40):
        flat[off - off % 16: off - off % 16 + 16] = torch.tensor([1e-39] * 16).to(BF16)   # a block of bf16 denormals
    flat[tail0 + 8 * 512 * 9 + 160] = float("nan")
    flat[n - 1] = float("-inf")
    x = xh.to(cuda)
    for B in (16, 64):
        got, want = dmx.ops.bfp_qdq(x, 8, B), oracle.bfp_cast(xh, 8, B, -1).to(BF16)
        # (NaN-aware: which NaN a poisoned block's elements become when narrowed to 16 bits is pinned in test_gpu_bfp.py)
        assert mismatches_nan_aware(got, want) == 0 and int(torch.isnan(got).sum()) == 2 * B
    # the compact kernel (19 vectors per lane at 4800 rows): the literal redo inside FULL tiles and in the partial one
    xh = _input(4800, BF16, seed=6)
    flat = xh.view(-1)
    n = flat.numel()
    for off in (48, 8 * 512 * 19 * 3 + 8 * 512 * 7 + 320, n - 64, n // 2):
        flat[off - off % 16: off - off % 16 + 16] = torch.tensor([1e-39] * 16).to(BF16)
    flat[8 * 512 * 19 * 100 + 5] = float("nan")
    flat[n - 3] = float("inf")
    x = xh.to(cuda)
    for B in (16, 64):
        got, want = dmx.ops.bfp_qdq(x, 8, B), oracle.bfp_cast(xh, 8, B, -1).to(BF16)
        assert mismatches_nan_aware(got, want) == 0 and int(torch.isnan(got).sum()) == 2 * B
