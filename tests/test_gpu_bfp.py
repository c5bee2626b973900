"""-m gpu parity tests: HIP BFP Q->DQ (through the C ABI) vs the CPU oracle, bit-exact.

Reference behaviour under test: numerical/format.py:304-343 BlockFloatingPoint.cast + quant_cpu.cpp:239-311,
CastTo's dtype contract numerical/cast.py:262,306.  Tolerance: NONE (bit patterns must match, incl. -0.0).
"""
import pytest
import torch

from _data import bits_equal, make

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16, torch.float32]


def _run(dmx, cuda, O, x, wl, B, dim=-1, sym=True, rounding="nearest", out_dtype=None, seed=0):
    got = dmx.ops.bfp_qdq(x.to(cuda), wl, B, dim, sym, rounding, out_dtype=out_dtype, seed=seed)
    want = O.bfp_cast(x, wl, B, dim, sym, rounding, seed).to(out_dtype or x.dtype)
    assert got.shape == x.shape and got.dtype == (out_dtype or x.dtype) and got.is_contiguous()
    return bits_equal(got, want.contiguous())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("wl", [4, 6, 8, 16])
@pytest.mark.parametrize("B", [16, 32, 64, 128])
@pytest.mark.parametrize("sym", [True, False])
def test_rows_mixed_inputs(dmx, cuda, oracle, dtype, wl, B, sym):
    x = make("mixed", (64, 512), seed=wl * 131 + B, dtype=dtype, block=B)
    assert _run(dmx, cuda, oracle, x, wl, B, sym=sym) == 0


@pytest.mark.parametrize("kind", ["normal", "heavy", "outlier", "ties", "zeros", "denormal"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_headline_format_each_input_kind(dmx, cuda, oracle, kind, dtype):
    # BFP[8|8]{16}(SN) = "BFP16 group 16", the north-star format
    x = make(kind, (128, 1024), seed=3, dtype=dtype, block=16)
    assert _run(dmx, cuda, oracle, x, 8, 16) == 0


@pytest.mark.parametrize("din,dout", [(torch.bfloat16, torch.float32), (torch.float16, torch.float32),
                                      (torch.float32, torch.bfloat16), (torch.float32, torch.float16)])
def test_mixed_io_dtypes(dmx, cuda, oracle, din, dout):
    x = make("heavy", (32, 256), seed=9, dtype=din)
    assert _run(dmx, cuda, oracle, x, 8, 64, out_dtype=dout) == 0


@pytest.mark.parametrize("B", [1, 2, 3, 4, 8, 24, 40, 256, 512])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_odd_block_sizes_and_ragged_tails(dmx, cuda, oracle, B, dtype):
    # torch.split semantics: last block shorter (format.py:324-326); LeNet fc1 in=400, 120, 84
    for shape in [(4, 40), (7, 400), (3, 120), (5, 84), (2, 1000)]:
        x = make("heavy", shape, seed=B, dtype=dtype)
        assert _run(dmx, cuda, oracle, x, 8, B) == 0, (shape, B)


@pytest.mark.parametrize("dim", [1, -2, 0, 2])
@pytest.mark.parametrize("B", [16, 64, 5])
def test_block_dim_layouts(dmx, cuda, oracle, dim, B):
    # conv activations / weights block along dim 1, attention multipliers along -2
    for shape in [(2, 32, 5, 5), (3, 6, 5, 5), (2, 1, 32, 32), (4, 64, 48)]:
        if dim >= len(shape):
            continue
        x = make("normal", shape, seed=dim + 10, dtype=torch.bfloat16)
        assert _run(dmx, cuda, oracle, x, 8, B, dim=dim) == 0, (shape, dim, B)
        assert _run(dmx, cuda, oracle, x.float(), 6, B, dim=dim, sym=False) == 0, (shape, dim, B)


@pytest.mark.parametrize("rounding", ["down", "up", "stochastic"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_other_rounding_modes(dmx, cuda, oracle, rounding, dtype):
    x = make("heavy", (64, 256), seed=5, dtype=dtype)
    assert _run(dmx, cuda, oracle, x, 8, 16, rounding=rounding, seed=1234) == 0
    assert _run(dmx, cuda, oracle, x, 6, 24, rounding=rounding, seed=99) == 0          # generic kernel
    assert _run(dmx, cuda, oracle, x.reshape(8, 8, 256), 8, 4, dim=1, rounding=rounding, seed=7) == 0


def test_fp32_double_rounding_class(dmx, cuda, oracle):
    """fp32 inputs within 2^-22 relative of a rounding tie: the reference rounds twice (x + 6*2^e, then the
    mantissa); a single-rounding implementation differs on these (SURVEY.md §7 'Hard parts')."""
    g = torch.Generator().manual_seed(0)
    B = 16
    base = torch.randint(-127, 127, (4096, B), generator=g).float() + 0.5      # exact ties at quantum 1
    eps = (torch.randint(-3, 4, (4096, B), generator=g).float()) * 2.0 ** -17   # nudges below fp32 ulp of 6*64
    x = base + eps
    x[:, 0] = 100.0  # block max exponent e = 6 -> quantum 2^0
    assert _run(dmx, cuda, oracle, x, 8, B) == 0


def test_full_size_headline_config(dmx, cuda, oracle):
    """BASELINE.json config 2: 4096x4096 bf16, BFP[8|8]{16}(SN)."""
    x = make("heavy", (4096, 4096), seed=0, dtype=torch.bfloat16)
    assert _run(dmx, cuda, oracle, x, 8, 16) == 0


def test_properties_at_full_size(dmx, cuda):
    """Size-independent properties on the 4096x4096 bf16 tensor (no oracle involved)."""
    x = make("heavy", (4096, 4096), seed=1, dtype=torch.bfloat16).to(cuda)
    q = dmx.ops.bfp_qdq(x, 8, 16)
    assert bits_equal(dmx.ops.bfp_qdq(q, 8, 16), q) == 0                      # idempotent
    qn = dmx.ops.bfp_qdq(-x, 8, 16)
    assert torch.equal(qn, -q)                                               # odd symmetry (symmetric format)
    # shard invariance: rows are independent -> quantising 8 row shards == quantising the whole (§8e)
    shards = torch.cat([dmx.ops.bfp_qdq(s, 8, 16) for s in x.chunk(8, dim=0)])
    assert bits_equal(shards, q) == 0
    # every block has at most 2^8-1 distinct codes on a common quantum: max/quantum <= 127
    qb = q.float().reshape(-1, 16)
    m = qb.abs().amax(dim=1, keepdim=True)
    quantum = torch.where(m > 0, torch.exp2(torch.floor(torch.log2(m)) - 6), torch.ones_like(m))
    codes = qb / quantum
    assert torch.all(codes == codes.round()) and float(codes.abs().max()) <= 127


def test_inplace_and_noncontiguous(dmx, cuda, oracle):
    x = make("normal", (64, 96), seed=2, dtype=torch.bfloat16)
    xt = x.t()  # non-contiguous view
    got = dmx.ops.bfp_qdq(xt.to(cuda), 8, 16)
    want = oracle.bfp_cast(xt, 8, 16).to(torch.bfloat16)
    assert bits_equal(got, want.contiguous()) == 0


def test_empty_and_scalar(dmx, cuda):
    assert dmx.ops.bfp_qdq(torch.empty(0, 16, device=cuda), 8, 16).shape == (0, 16)
    assert dmx.ops.bfp_qdq(torch.empty(4, 0, device=cuda), 8, 16).shape == (4, 0)
    assert float(dmx.ops.bfp_qdq(torch.tensor(1.2345, device=cuda), 8, 16)) == 1.234375


def test_error_behaviour(dmx, cuda):
    with pytest.raises(dmx.DmxqError):
        dmx.ops.bfp_qdq(torch.randn(4, 16), 8, 16)                          # CPU tensor: loud, no fallback
    with pytest.raises(NotImplementedError):
        dmx.ops.bfp_qdq(torch.randn(4, 16, device=cuda), 24, 16)            # reference UB region
    with pytest.raises(TypeError):
        dmx.ops.bfp_qdq(torch.zeros(4, 16, device=cuda, dtype=torch.int32), 8, 16)
