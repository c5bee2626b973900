"""-m gpu: SBFP / MXFP single-kernel casts vs the CPU oracle (bit-exact), over layouts and dtypes."""
import pytest
import torch

from _data import bits_equal, make

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(4, 16, 4, 4, 7), (4, 16, 4, 4, 12), (8, 64, 2, 5, 15), (6, 32, 3, 4, 7), (4, 24, 4, 4, 7)])
def test_sbfp(dmx, cuda, oracle, dtype, cfg):
    p, B, man, exp, bias = cfg
    for shape, dim in (((64, 512), -1), ((7, 400), -1), ((4, 64, 48), 1), ((3, 40, 8), -2), ((2, 6, 5, 5), 1)):
        x = make("mixed_nd", shape, seed=p + B, dtype=dtype, block=8)
        x.view(-1)[: min(64, x.numel())] = 0                          # leading all-zero block(s): passed through
        got = dmx.ops.sbfp_qdq(x.to(cuda), p, B, man, exp, bias, True, True, True, dim)
        want = oracle.sbfp_cast(x, p, B, man, exp, bias, True, True, True, dim).to(dtype)
        assert bits_equal(got, want.contiguous()) == 0, (shape, dim)
    x = make("heavy", (32, 256), seed=1, dtype=dtype)
    got = dmx.ops.sbfp_qdq(x.to(cuda), p, B, man, exp, bias, out_dtype=torch.float32)
    assert bits_equal(got, oracle.sbfp_cast(x, p, B, man, exp, bias).contiguous()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(3, 4, 32), (2, 5, 64), (3, 2, 32), (2, 3, 128), (1, 2, 32), (3, 4, 20)])
def test_mxfp(dmx, cuda, oracle, dtype, cfg):
    man, exp, B = cfg
    for shape, dim in (((64, 512), -1), ((7, 400), -1), ((4, 64, 48), 1), ((3, 40, 8), -2)):
        x = make("heavy", shape, seed=man + exp + B, dtype=dtype)
        got = dmx.ops.mxfp_qdq(x.to(cuda), man, exp, B, dim)
        want = oracle.mxfp_cast(x, man, exp, B, dim).to(dtype)
        assert bits_equal(got, want.contiguous()) == 0, (shape, dim)
    z = torch.zeros(4, 64, dtype=dtype)
    assert torch.equal(dmx.ops.mxfp_qdq(z.to(cuda), man, exp, B).cpu(), z)  # zero block stays zero (reference: NaN)


def test_format_objects_and_weight_storage_rule(dmx, cuda, oracle):
    w = make("normal", (48, 64), seed=5)
    m = dmx.nn.Linear(64, 48)
    m.weight.data = w.clone()
    m = m.to(cuda)
    dmx.configure_model(m, *dmx.config_rules.SBFP_WEIGHT_STORAGE)         # weight_storage_format = SBFP12_16
    assert bits_equal(m._weight, oracle.sbfp_cast(w, 4, 16, 4, 4, 7).contiguous()) == 0
    f = dmx.format.MXFP8_E4M3K32
    assert bits_equal(f.cast(w.to(cuda)), oracle.mxfp_cast(w, 3, 4, 32).contiguous()) == 0
    assert repr(f) == "MXFP8[E4M3]{32}" and f.bytes_per_elem == 1 + (9 / 8) / 32   # reference formula (format.py:566-571)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [(8, 16, True), (8, 64, True), (8, 64, False), (6, 32, True), (4, 128, False), (8, 24, True)])
def test_packed_bfp_roundtrip_equals_fake_quant(dmx, cuda, oracle, dtype, cfg):
    """unpack(pack(x)) == Q->DQ(x) bit for bit (round trip property), codes within the p-bit range,
    exponents = biased exponent of each block's maximum."""
    wl, B, sym = cfg
    for shape in ((64, 512), (7, 400), (3, 5, 96)):
        x = make("mixed_nd", shape, seed=wl + B, dtype=dtype, block=8).to(cuda)
        mant, exps = dmx.ops.bfp_pack(x, wl, B, sym)
        assert mant.dtype == torch.int8 and exps.dtype == torch.uint8 and mant.shape == x.shape
        lim = 2 ** (wl - 1) - (1 if sym else 0)
        assert int(mant.min()) >= -lim and int(mant.max()) <= 2 ** (wl - 1) - 1
        back = dmx.ops.bfp_unpack(mant, exps, wl, B, out_dtype=dtype)
        qdq = dmx.ops.bfp_qdq(x, wl, B, -1, sym)
        normal = (x.float().reshape(-1, x.shape[-1]).abs().amax(1) >= 0)  # every row here has normal maxima or zeros
        assert normal.all() and bits_equal(back, qdq) == 0 or _only_zero_blocks_differ(x, back, qdq, B)
        L = x.shape[-1]
        pad = (-L) % B
        xm = torch.nn.functional.pad(x.float().abs(), (0, pad)).reshape(*x.shape[:-1], -1, B).amax(-1)
        want_e = ((xm.view(torch.int32) >> 23) & 0xFF).to(torch.uint8)
        want_e = torch.where(xm < 2.0 ** -126, torch.zeros_like(want_e), want_e)
        assert torch.equal(exps, want_e)


def _only_zero_blocks_differ(x, back, qdq, B):
    """-0.0 vs +0.0 in all-zero blocks is the only allowed difference (packed zeros carry no sign)"""
    d = (back.float().view(torch.int32) != qdq.float().view(torch.int32))
    return bool(((back == 0) & (qdq == 0))[d].all())
