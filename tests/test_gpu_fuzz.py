"""Randomised differential test: every quantisation / mask entry point against the oracle on randomly drawn shapes,
dtypes, layouts and parameters (fixed seed, so a failure is reproducible by its case number).  Complements the
hand-picked cases of the other files: the dispatcher picks among ~10 kernels per op by shape and alignment, and this
walks across those boundaries blindly."""
import random

import pytest
import torch

from _data import bits_equal, make, mismatches_nan_aware

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16, torch.float16]
KINDS = ["normal", "heavy", "mixed", "outlier", "zeros", "ties"]


def _shape(rng, max_elems=40000):
    nd = rng.choice([1, 2, 2, 2, 3, 4])
    while True:
        shape = tuple(rng.choice([1, 2, 3, 5, 7, 8, 13, 16, 24, 31, 32, 48, 64, 100, 128, 200, 256, 500, 1000, 1500]) for _ in range(nd))
        n = 1
        for s in shape:
            n *= s
        if n <= max_elems:
            return shape


@pytest.mark.parametrize("chunk", range(8))
def test_bfp_random_cases(dmx, cuda, oracle, chunk):
    rng = random.Random(1000 + chunk)
    for case in range(40):
        shape = _shape(rng)
        dtype = rng.choice(DT)
        dim = rng.randrange(-len(shape), len(shape))
        B = rng.choice([1, 2, 3, 8, 12, 16, 16, 32, 64, 64, 128, 256, 512])
        wl = rng.choice([2, 3, 4, 6, 8, 8, 8, 11, 12, 15, 16, 21, 22])
        sym = rng.random() < 0.7
        rounding = rng.choice(["nearest"] * 5 + ["down", "up", "stochastic"])
        kind = rng.choice(KINDS) if sym else rng.choice(["normal", "mixed_nd", "heavy"])
        out_dtype = rng.choice([None, None, None, torch.float32 if dtype != torch.float32 else torch.bfloat16])
        x = make(kind, shape, seed=chunk * 100 + case, dtype=dtype, block=max(2, min(B, 64)))
        off = rng.choice([0, 0, 1])  # a view that starts one element into its allocation
        if off:
            base = torch.empty(x.numel() + 1, dtype=dtype)
            base[1:] = x.reshape(-1)
            xg = base.to(cuda)[1:].view(shape)
        else:
            xg = x.to(cuda)
        tag = (chunk, case, shape, dtype, dim, B, wl, sym, rounding, kind, out_dtype, off)
        got = dmx.ops.bfp_qdq(xg, wl, B, dim, sym, rounding, out_dtype=out_dtype, seed=7)
        want = oracle.bfp_cast(x, wl, B, dim, sym, rounding, 7).to(out_dtype or dtype).contiguous()
        assert mismatches_nan_aware(got, want) == 0, tag


@pytest.mark.parametrize("chunk", range(4))
def test_elementwise_random_cases(dmx, cuda, oracle, chunk):
    rng = random.Random(2000 + chunk)
    for case in range(40):
        shape = _shape(rng)
        dtype = rng.choice(DT)
        x = make(rng.choice(["normal", "heavy", "ties"]), shape, seed=chunk * 100 + case, dtype=dtype) * rng.choice([1.0, 10.0, 100.0])
        off = rng.choice([0, 0, 1, 3])  # a view that starts `off` elements into its allocation (unaligned 16-byte accesses)
        if off:
            base = torch.empty(x.numel() + off, dtype=dtype)
            base[off:] = x.reshape(-1)
            xg = base.to(cuda)[off:].view(shape)
        else:
            xg = x.to(cuda)
        tag = (chunk, case, shape, dtype, off)
        which = rng.choice(["float", "fixed", "affine"])
        if which == "float":
            man, exp = rng.choice([(10, 5), (7, 8), (3, 4), (2, 5), (1, 2), (0, 8), (3, 2), (22, 8)])
            bias = rng.choice([(1 << (exp - 1)) - 1, 7, 1, 15]) if exp < 8 else 127
            flush = rng.random() < 0.5
            rounding = rng.choice(["nearest"] * 4 + ["down", "up", "stochastic"])
            got = dmx.ops.float_qdq(xg, man, exp, bias, flush, False, rounding, seed=3)
            want = oracle.floating_point_cast(x, man, exp, bias, flush, False, rounding, 3).to(dtype)
            assert mismatches_nan_aware(got, want) == 0, tag + (which, man, exp, bias, flush, rounding)
        elif which == "fixed":
            p, f = rng.choice([(8, 0), (4, 0), (8, 4), (16, 8), (12, -2), (2, 0), (24, 10)])
            clamp, sym = rng.random() < 0.8, rng.random() < 0.5
            rounding = rng.choice(["nearest"] * 4 + ["stochastic"])
            got = dmx.ops.fixed_qdq(xg, p, f, clamp, sym, rounding, seed=5)
            want = oracle.fixed_point_cast(x, p, f, clamp, sym, rounding, 5).to(dtype)
            assert mismatches_nan_aware(got, want) == 0, tag + (which, p, f, clamp, sym, rounding)
        else:
            p = rng.choice([8, 4])
            sym = rng.random() < 0.5
            ax = rng.randrange(-len(shape), len(shape))
            C = shape[ax]
            mode = rng.choice(["tensor", "channel", "group"])
            gs = None if mode != "group" else rng.choice([1, 2, 3, 16, C, C + 5])
            ng = 1 if mode == "tensor" else (C if mode == "channel" else -(-C // gs))
            sc = (torch.rand(ng) * 0.2 + 0.01)
            zp = torch.randint(-20, 20, (ng,)) if not sym else torch.zeros(ng, dtype=torch.int64)
            ch = None if mode == "tensor" else ax
            got = dmx.ops.fixed_qdq(xg, p, 0, True, True, scale=sc.to(cuda), zero_point=zp.to(cuda), ch_axis=ch, group_size=gs)
            want = oracle.fixed_point_affine_cast(x, p, 0, True, True, sc, zp, ch_axis=ch, group_size=gs).to(dtype)
            assert mismatches_nan_aware(got, want) == 0, tag + (which, p, mode, gs, ax)


@pytest.mark.parametrize("chunk", range(3))
def test_mask_and_reduction_random_cases(dmx, cuda, oracle, chunk):
    rng = random.Random(3000 + chunk)
    for case in range(30):
        shape = _shape(rng)
        dtype = rng.choice(DT)
        s = make(rng.choice(["normal", "ties", "zeros"]), shape, seed=chunk * 100 + case, dtype=dtype)
        tag = (chunk, case, shape, dtype)
        M = rng.choice([2, 4, 8, 16])
        dims = [d for d in range(len(shape)) if shape[d] % M == 0]
        if dims:
            d = rng.choice(dims)
            K = rng.randrange(1, M + 1)
            assert bits_equal(dmx.ops.nm_mask(s.to(cuda), K, M, d), oracle.nm_mask(s, K, M, d).contiguous()) == 0, tag + ("nm", K, M, d)
        density = rng.choice([0.0, 0.1, 0.5, 0.5, 0.75, 1.0])
        assert bits_equal(dmx.ops.topk_mask(s.to(cuda), density), oracle.topk_mask(s, density)) == 0, tag + ("topk", density)
        ax = rng.randrange(-len(shape), len(shape))
        gs = rng.choice([1, 2, 16, shape[ax], shape[ax] + 3])
        mn, mx = dmx.ops.group_minmax(s.to(cuda), ax, gs)
        omn, omx = oracle.group_minmax(s, ax, gs)
        assert bits_equal(mn, omn) == 0 and bits_equal(mx, omx) == 0, tag + ("minmax", ax, gs)
        assert bits_equal(dmx.ops.channel_maxabs(s.to(cuda), ax), oracle.channel_maxabs(s, ax)) == 0, tag + ("maxabs", ax)
        fmt = rng.choice(["sbfp", "mxfp"])
        B = rng.choice([8, 16, 32, 64])
        d = rng.randrange(-len(shape), len(shape))
        x = make("heavy", shape, seed=case, dtype=dtype)
        if fmt == "sbfp":
            got = dmx.ops.sbfp_qdq(x.to(cuda), 4, B, 4, 4, 7, True, block_dim=d)
            want = oracle.sbfp_cast(x, 4, B, 4, 4, 7, True, block_dim=d).to(dtype)
        else:
            man, exp = rng.choice([(3, 4), (2, 5), (1, 2), (3, 2)])
            got = dmx.ops.mxfp_qdq(x.to(cuda), man, exp, B, block_dim=d)
            want = oracle.mxfp_cast(x, man, exp, B, block_dim=d).to(dtype)
        assert mismatches_nan_aware(got, want.contiguous()) == 0, tag + (fmt, B, d)
