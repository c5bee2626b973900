"""N > 1 path on CPU: world_size-2 gloo processes.  The shard plan, the harness gather and the one real exchange
(all_reduce MAX of per-channel maxima) are exercised with the ORACLE standing in for the GPU kernels — the
property under test is shard invariance: shard -> op -> concat == op on the whole tensor."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle as O
    from _data import make
    from dmx_compressor_amd import parallel as P

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        # 1. BFP along the last dim: any row split is block aligned
        x = make("heavy", (37, 64), seed=1)
        mine = P.my_rows(x, rank, world)
        full = P.gather_rows(O.bfp_cast(mine, 8, 16), x.shape[0], world)
        ok &= torch.equal(full, O.bfp_cast(x, 8, 16))
        # 2. blocks ALONG dim 0 (block_dim = 0, B = 8): boundaries must be multiples of B
        y = make("normal", (40, 12), seed=2)
        mine = P.my_rows(y, rank, world, multiple=8)
        full = P.gather_rows(O.bfp_cast(mine, 8, 8, 0).contiguous(), y.shape[0], world, multiple=8)
        ok &= torch.equal(full, O.bfp_cast(y, 8, 8, 0).contiguous())
        # 3. group-quant slabs of 4 rows along ch_axis 0 stay inside a shard: per-group min/max need no exchange
        mn, mx = O.group_minmax(P.my_rows(y, rank, world, multiple=4), 0, 4)
        gm = P.gather_rows(torch.stack([mn, mx], 1), 10, world)
        wn, wx = O.group_minmax(y, 0, 4)
        ok &= torch.equal(gm, torch.stack([wn, wx], 1))
        # 4. SmoothQuant weight maxabs per INPUT channel is a reduction over rows -> one all_reduce(MAX)
        w = make("normal", (33, 24), seed=3)
        part = O.channel_maxabs(P.my_rows(w, rank, world), -1) if P.my_rows(w, rank, world).shape[0] else torch.zeros(24)
        ok &= torch.equal(P.allreduce_max_(part), O.channel_maxabs(w, -1))
        # 5. bench-style timing reduction: max over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        ok &= float(P.allreduce_max_(t)) == float(world)
        # 6. the same exchange THROUGH THE PRODUCT's host code (smoothquant.py / observer.py `set_process_group`): the module calls
        #    parallel.allreduce_max_ / allreduce_min_ itself.  No GPU here, so the module's kernel front end is swapped for the oracle
        #    (test infrastructure standing in for the launches; the GPU twin is tests/test_gpu_multi_gpu.py).
        from types import SimpleNamespace

        import dmx_compressor_amd as dmx
        import dmx_compressor_amd.observer as obs_mod
        import dmx_compressor_amd.smoothquant as sq_mod

        def sq_scale(a, b, alpha, smin):
            return torch.clamp(a.pow(alpha) / torch.clamp(b, min=smin).pow(1.0 - alpha), min=smin)

        sq_mod.ops = SimpleNamespace(channel_maxabs=O.channel_maxabs, smoothquant_scale=sq_scale)
        obs_mod.ops = SimpleNamespace(group_minmax=O.group_minmax)
        inp = make("heavy", (19, 24), seed=4)
        shard = P.my_rows(w, rank, world)
        sq = dmx.ActivationWeightSmoothQuant(ch_axis=-1, win_ch_axis=-1)
        sq.set_process_group(P.WORLD, weight=True, input=False)
        sq(inp, shard)
        ok &= torch.equal(sq.weight_maxabs, O.channel_maxabs(w, -1))
        ok &= torch.equal(sq.scale, sq_scale(O.channel_maxabs(inp, -1), O.channel_maxabs(w, -1), 0.5, 1e-5))
        sq2 = dmx.ActivationWeightSmoothQuant(ch_axis=-1, win_ch_axis=-1)   # tokens sharded too
        sq2.set_process_group(P.WORLD, weight=True, input=True)
        sq2(P.my_rows(inp, rank, world), shard)
        ok &= torch.equal(sq2.scale, sq.scale)
        sq3 = dmx.ActivationWeightSmoothQuant(ch_axis=-1, win_ch_axis=-1)   # no group: the shard's own maxima, another scale
        sq3(inp, shard)
        ok &= not torch.equal(sq3.scale, sq.scale)
        mm = dmx.MinMaxObserver(qscheme=torch.per_tensor_symmetric)
        mm.set_process_group(P.WORLD)
        mm(shard)
        ok &= float(mm.min_val) == float(w.min()) and float(mm.max_val) == float(w.max())
        pc = dmx.MinMaxObserver(qscheme=torch.per_channel_symmetric, ch_axis=-1)  # per input channel over row shards: same groups on every rank
        pc.set_process_group(P.WORLD)
        pc(shard)
        ok &= torch.equal(pc.min_val, w.amin(0)) and torch.equal(pc.max_val, w.amax(0))
        empty = dmx.MinMaxObserver(qscheme=torch.per_channel_symmetric, ch_axis=-1)   # a rank with an EMPTY shard still takes part
        empty.set_process_group(P.WORLD)
        empty(w[:5] if rank == 0 else w[:0])
        ok &= torch.equal(empty.min_val, w[:5].amin(0)) and torch.equal(empty.max_val, w[:5].amax(0))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_row_shards_plan():
    sys.path.insert(0, ROOT)
    from dmx_compressor_amd.parallel import row_shards

    assert row_shards(4096, 8) == [(i * 512, (i + 1) * 512) for i in range(8)]
    assert row_shards(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert row_shards(40, 2, multiple=16) == [(0, 32), (32, 40)]          # ragged last unit stays whole
    assert row_shards(3, 8) == [(0, 1), (1, 2), (2, 3)] + [(3, 3)] * 5      # more ranks than rows
    assert row_shards(14336, 8, multiple=128) == [(i * 1792, (i + 1) * 1792) for i in range(8)]
    for n, w, m in ((4097, 8, 16), (1, 1, 1), (0, 4, 2), (100, 7, 3)):
        sh = row_shards(n, w, m)
        assert sh[0][0] == 0 and sh[-1][1] == n and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        assert all(s % m == 0 for s, _ in sh)
    with pytest.raises(ValueError):
        row_shards(8, 0)


@pytest.mark.timeout(180)
def test_shard_invariance_world2_gloo():
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
