"""-m gpu, round 4: the weight chain of a whole layer in one launch (dmxq_weight_hypernet_multi), the calibration reductions'
new kernels, the reference-side patch applied to the mirror's classes."""
import ctypes

import pytest
import torch

from _data import bits_equal, make

pytestmark = pytest.mark.gpu
BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32

# Llama-3-8B decoder layer (bench.py LLAMA_LAYER), [out_features, in_features]
LLAMA = [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)]


def _shards(world, rank, shapes=LLAMA):
    from dmx_compressor_amd.parallel import row_shards

    out = []
    for rows, cols in shapes:
        s, e = row_shards(rows, world)[rank]
        out.append((e - s, cols))
    return out


# ------------------------------------------------------------------------------------------------ dmxq_weight_hypernet_multi
@pytest.mark.parametrize("world", [8, 4])
def test_hypernet_multi_equals_per_tensor_calls_and_the_oracle_on_llama_shards(dmx, cuda, oracle, world):
    """One launch over a rank's seven Llama-3-8B weight shards (2:4 mask -> BFP16_64, bf16 weight and score) == one
    dmxq_weight_hypernet call per shard == the oracle composed like the reference (sparse.py:287-301 -> format.py:304-343)."""
    shapes = _shards(world, world - 1)
    ws = [make("normal", s, seed=11 + i, dtype=BF16) * 0.02 for i, s in enumerate(shapes)]
    ss = [make("normal", s, seed=91 + i, dtype=BF16).abs() for i, s in enumerate(shapes)]
    wd, sd = [w.to(cuda) for w in ws], [s.to(cuda) for s in ss]
    got = dmx.ops.weight_hypernet_multi(wd, 8, 64, True, sd, 2, 4)
    assert got is not None and len(got) == len(ws)
    for w, s, g, wh, sh in zip(wd, sd, got, ws, ss):
        one = dmx.ops.weight_hypernet(w, 8, 64, True, s, 2, 4)
        assert g.dtype == one.dtype == BF16 and bits_equal(g, one) == 0
        assert bits_equal(g, oracle.bfp_cast(oracle.sparsify(wh, sh, 2, 4), 8, 64).to(BF16)) == 0


@pytest.mark.parametrize("wdt,sdt,odt", [(BF16, F32, BF16), (BF16, F32, None), (F16, F16, None), (F32, F32, None), (F16, F32, F16)])
@pytest.mark.parametrize("M,K", [(4, 2), (8, 4), (2, 1), (0, 0)])
@pytest.mark.parametrize("with_scale", [False, True])
def test_hypernet_multi_every_dtype_triple_mask_and_scale(dmx, cuda, oracle, wdt, sdt, odt, M, K, with_scale):
    """mixed sizes incl. tensors smaller than one tile, a tile-boundary straddler and an empty one; asymmetric + B = 16 too"""
    shapes = [(3, 64), (128, 192), (0, 64), (1000, 128), (257, 320), (16, 8192)]
    ws = [(make("heavy", s, seed=5 + i, dtype=F32) * 0.1).to(wdt).to(cuda) for i, s in enumerate(shapes)]
    ss = [make("normal", s, seed=55 + i, dtype=F32).abs().to(sdt).to(cuda) for i, s in enumerate(shapes)] if M else None
    qs = [(torch.rand(s[1], generator=torch.Generator().manual_seed(i)) * 3 + 0.1).to(cuda) for i, s in enumerate(shapes)] if with_scale else None
    for B, sym in ((64, True), (16, False)):
        got = dmx.ops.weight_hypernet_multi(ws, 8, B, sym, ss, K, M, qs, out_dtype=odt)
        assert got is not None
        for i, w in enumerate(ws):
            if w.numel() == 0:
                assert got[i].numel() == 0
                continue
            one = dmx.ops.weight_hypernet(w, 8, B, sym, ss[i] if M else None, K, M, qs[i] if with_scale else None, out_dtype=odt)
            assert one is not None and got[i].dtype == one.dtype and bits_equal(got[i], one) == 0, (i, B, sym)
            # ... and the oracle chain with the reference's dtype flow: mask multiply in the promoted dtype, scale_weight's `.to(wgt.dtype)`
            x = w.cpu()
            if M:
                x = oracle.sparsify(x, ss[i].cpu(), K, M)
            if with_scale:
                x = (x.float() * qs[i].cpu()).to(x.dtype)
            want = oracle.bfp_cast(x, 8, B, -1, sym).to(x.dtype).to(got[i].dtype)
            assert bits_equal(got[i], want) == 0, (i, B, sym)


def test_hypernet_multi_more_than_one_launch_and_unfusable_sets(dmx, cuda):
    """70 weights = three launches of <= 32; a set with one unfusable member (L % B != 0) returns None and launches nothing"""
    ws = [make("normal", (8 + i, 128), seed=i, dtype=BF16).to(cuda) for i in range(70)]
    got = dmx.ops.weight_hypernet_multi(ws, 8, 64)
    for w, g in zip(ws, got):
        assert bits_equal(g, dmx.ops.bfp_qdq(w, 8, 64)) == 0
    bad = ws[:3] + [make("normal", (4, 96), seed=1, dtype=BF16).to(cuda)]
    assert dmx.ops.weight_hypernet_multi(bad, 8, 64) is None
    assert dmx.ops.weight_hypernet_multi([], 8, 64) == []


def test_hypernet_multi_c_abi_validation(dmx, cuda):
    from dmx_compressor_amd import _lib as lib

    L = lib.lib()
    w = torch.zeros(4, 64, dtype=BF16, device=cuda)
    o = torch.empty_like(w)
    d = (lib.HypernetDesc * 2)()
    for e in d:
        e.w, e.score, e.sq_scale, e.out, e.rows, e.L = w.data_ptr(), None, None, o.data_ptr(), 4, 64
    sp = lib.stream_of(w)
    f = L.dmxq_weight_hypernet_multi
    assert f(d, 2, lib.BF16, 0, 0, 0, lib.BF16, 64, 8, 1, sp) == lib.OK
    assert f(d, -1, lib.BF16, 0, 0, 0, lib.BF16, 64, 8, 1, sp) == lib.ERR_BAD_ARG
    assert f(None, 2, lib.BF16, 0, 0, 0, lib.BF16, 64, 8, 1, sp) == lib.ERR_BAD_ARG
    assert f(d, 2, lib.BF16, lib.BF16, 2, 4, lib.BF16, 64, 8, 1, sp) == lib.ERR_BAD_ARG        # mask without scores
    assert f(d, 2, lib.BF16, 0, 0, 0, lib.BF16, 48, 8, 1, sp) == lib.ERR_UNSUPPORTED            # block size not 2^k
    assert f(d, 2, lib.BF16, 0, 0, 0, lib.F16, 64, 8, 1, sp) == lib.ERR_UNSUPPORTED             # dtype triple not instantiated
    d[1].sq_scale = torch.ones(64, device=cuda).data_ptr()
    assert f(d, 2, lib.BF16, 0, 0, 0, lib.BF16, 64, 8, 1, sp) == lib.ERR_BAD_ARG                # scale on some tensors only
    assert f(d, 0, lib.BF16, 0, 0, 0, lib.BF16, 64, 8, 1, sp) == lib.OK
    torch.cuda.synchronize()
