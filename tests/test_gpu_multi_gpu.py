"""-m gpu; the 2-rank cases need >= 2 GPUs (skipped on the 1-GPU box), the self-launch cases run on one: the N > 1 path of bench.py as the driver launches it -- one process
per GPU under torch.distributed.run, RCCL ("nccl") for the barrier, the max-over-ranks of the timing and the harness-only
all_gather of the output shards, which rank 0 compares bit for bit with its own whole-tensor result (SURVEY.md §8e:
shard -> op -> concat == op on the whole).  The CPU twin with gloo is tests/test_parallel_gloo.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port, n=2, plain=False):
    """plain: `python bench.py --gpus N ...` (bench.py starts its own ranks); else the driver's torch.distributed.run form"""
    launcher = [] if plain else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                                 "127.0.0.1", "--master-port", str(port)]
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "6", "--warmup", "2",
                                         "--replays", "3", "--no-cpu-baseline", "--no-tier2"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=500 if n > 2 else 300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,scaling", [([], "weak"), (["--workload", "llama-shard", "--op", "bfp", "--layers", "1"], "strong"),
                                           (["--workload", "llama-shard", "--layers", "1"], "strong")])
@pytest.mark.parametrize("plain", [False, True])
def test_bench_two_ranks_shards_and_gathers(extra, scaling, plain):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    line = _run(extra, 29500 + len(extra), plain=plain)
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == scaling
    assert "all_gather of 2 row shards" in line["config"]["check"] and "bit-exact" in line["config"]["check"]
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1.0


@pytest.mark.parametrize("extra,scaling", [([], "weak"), (["--workload", "llama-shard", "--op", "bfp", "--layers", "1"], "strong"),
                                           (["--workload", "llama-shard", "--layers", "1"], "strong")])
def test_bench_two_ranks_share_one_gpu_over_gloo(extra, scaling):
    """The N > 1 path on REAL kernels with the lease's single GPU: world = 2, both ranks on cuda:0, the harness transport swapped
    for gloo (`--dist-backend gloo`; RCCL refuses two ranks per device).  Everything else is the N = 2 run: row shards, K launches
    per rank through the C ABI, max over ranks of every region, all_gather of the output shards, rank 0's whole-tensor bit
    compare, every rank's slots against the oracle.  Only the RCCL transport itself is left to the 2-GPU test above."""
    line = _run(extra + ["--dist-backend", "gloo", "--nbuf", "4"], 29620 + len(extra), n=2, plain=True)
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == scaling
    assert "all_gather of 2 row shards" in line["config"]["check"] and "bit-exact" in line["config"]["check"]
    assert "oracle" in line["config"]["check"] and "gloo" in line["config"]["dist_backend"]
    if not extra:
        assert line["config"]["per_gpu_elements_per_step"] == 4096 * 4096
        assert "4/4 slots == oracle" in line["config"]["check"] and "4/4 slots == SHA-256 of the reference" in line["config"]["check"]
    else:
        assert "56/56 row pieces" in line["config"]["check"], line["config"]["check"]
    assert len(line["config"]["per_rank"]["value"]) == 2
    assert line["value"] > 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra,scaling", [([], "weak"), (["--workload", "llama-shard", "--op", "hypernet", "--layers", "1"], "strong")])
def test_bench_eight_ranks_share_one_gpu_over_gloo(extra, scaling):
    """The N = 8 rehearsal (round 5): world = 8 on the lease's one GPU over gloo -- the driver's 8-GPU run must not be the first time
    world = 8 executes.  `parallel.row_shards(., 8)`, K launches per rank through the C ABI (c2: one [4096, 4096] shard per rank;
    llama-shard: a rank's seven weight shards in ONE multi-tensor launch), max over ranks of every region, all_gather of the eight
    output shards, rank 0's whole-tensor bit compare, every rank's slots against the oracle.  Not a scaling figure: eight ranks time-share
    one GPU (`config.dist_backend` says so); what is exercised is everything BUT the RCCL transport."""
    line = _run(extra + ["--dist-backend", "gloo", "--nbuf", "4"], 29700 + len(extra), n=8, plain=True)
    assert line["n_gpus"] == 8 and line["steps"] == 6 and line["scaling"] == scaling
    assert "all_gather of 8 row shards" in line["config"]["check"] and "bit-exact" in line["config"]["check"]
    assert "oracle" in line["config"]["check"] and "gloo" in line["config"]["dist_backend"]
    if not extra:
        assert line["config"]["per_gpu_elements_per_step"] == 4096 * 4096
        assert "4/4 slots == oracle" in line["config"]["check"]
        # round 6: the reference's digests exist for the slots of ALL eight ranks; the reported count is the minimum over ranks
        assert "4/4 slots == SHA-256 of the reference" in line["config"]["check"], line["config"]["check"]
    else:
        # ... and for every eighth of the seven weights' rows (tests/golden/llama_shard_digests.json): 7 weights x 8 pieces over the 8 ranks
        assert "56/56 row pieces" in line["config"]["check"] and "SHA-256 of the reference's Sparsify -> CastTo" in line["config"]["check"], line["config"]["check"]
    pr = line["config"]["per_rank"]
    assert len(pr["value"]) == 8 and 0 < pr["value_min"] <= pr["value_max"]
    assert line["value"] > 0
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):   # (kept for profiles/: the line of the rehearsal, labelled by its own dist_backend field)
        with open(os.path.join(out, "bench_line_world8_gloo" + ("_llama_hypernet" if extra else "") + ".json"), "w") as f:
            json.dump(line, f)


@pytest.mark.parametrize("extra", [[], ["--workload", "llama-shard", "--layers", "1"]])
def test_bench_starts_its_own_ranks(extra):
    """The self-launch path on ONE GPU: `bench.py --gpus 1 --spawn` runs its single rank as a torch.distributed.run child
    (RCCL initialised, barrier / all_reduce / gather code path of N > 1 taken with world = 1) and relays one JSON line."""
    line = _run(extra + ["--spawn"], 0, n=1, plain=True)
    assert line["n_gpus"] == 1 and line["steps"] == 6
    assert "bit-exact" in line["config"]["check"]
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1.0


def test_bench_refuses_more_ranks_than_gpus():
    n = torch.cuda.device_count() + 1   # (with the default RCCL transport; `--dist-backend gloo` lets ranks share a GPU)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert p.returncode == 2 and "GPU(s)" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


# ------------------------------------------------------------------------------------------------------------------------------
# The one real exchange of the path IN THE PRODUCT: SmoothQuant calibration of a ROW-SHARDED Linear.  Two ranks share the
# lease's GPU over gloo (the pattern above); each holds half the rows of Whisper-small's fc1 weight [3072, 768] and calibrates on the
# same activations with `smoothquant.set_process_group(parallel.WORLD)`: the per-input-channel weight maxima are completed by ONE
# all_reduce(MAX), so both ranks end with the whole tensor's scale -- the bits a single process computes from the whole weight, within
# 4 ulp of the REFERENCE's committed scale (tests/golden/model_scales.npz) -- and the scaled + cast weight shards, gathered, are the
# reference's whole-tensor `_weight` (SHA-256 in tests/golden/model_shapes.json).  Reference: numerical/smoothquant.py:285-321.
def _sq_shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    import torch.distributed as dist

    import dmx_compressor_amd as dmx
    from _model_shapes import _configure, digest, make_api, stage_input
    from dmx_compressor_amd import parallel as P

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        api = make_api(dmx.nn, dmx)
        gold = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(gold, "model_shapes.json")) as f:
            expected = json.load(f)
        z = np.load(os.path.join(gold, "model_scales.npz"))
        ref_scale = torch.from_numpy(np.ascontiguousarray(z["whisper/fc1"]).view(np.int32).copy()).view(torch.float32)
        dt, fin, fout = torch.float32, 768, 3072
        W, b = stage_input("fc1/weight", (fout, fin), dt, 0.03), stage_input("fc1/bias", (fout,), dt, 0.02)
        x = stage_input("fc1/x", (1, 1500, fin), dt, 2.0, "heavy").clamp(-1e3, 1e3)
        x[..., :: max(fin // 12, 1)] *= 20.0
        sq_hp = api.ModuleSQHP(migration_strength=0.5, fuse_to_weight=False)

        def calibrated(rows, group):
            m = dmx.nn.Linear(fin, rows.stop - rows.start, bias=True)
            m.weight.data, m.bias.data = W[rows].clone(), b[rows].clone()
            m = m.to(dev)
            _configure([m], api.config_rules.BASIC)
            if group is not None:
                m.smoothquant.set_process_group(group, weight=True, input=False)
            with m.calibrating_smoothquant(sq_hp), torch.no_grad():
                m(x.to(dev))
            return m

        s, e = P.row_shards(fout, world)[rank]
        m = calibrated(slice(s, e), P.WORLD)
        own = m.smoothquant.scale.detach().float().cpu().reshape(-1).clone()
        res = {"rank": rank}
        # (a) every rank holds the whole tensor's scale: identical across ranks, identical to an unsharded calibration
        both = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(both, own)
        res["same_on_all_ranks"] = all(torch.equal(both[0].view(torch.int32), t.view(torch.int32)) for t in both)
        whole = calibrated(slice(0, fout), None).smoothquant.scale.detach().float().cpu().reshape(-1)
        res["equals_unsharded"] = torch.equal(own.view(torch.int32), whole.view(torch.int32))
        shard_only = calibrated(slice(s, e), None).smoothquant.scale.detach().float().cpu().reshape(-1)
        res["exchange_matters"] = not torch.equal(own.view(torch.int32), shard_only.view(torch.int32))   # per-shard maxima give another scale
        res["rel_vs_reference"] = ((own - ref_scale).abs() / ref_scale.abs()).max().item()
        # (b) with the reference's scale as the stage input (the harness convention: two powf differ by an ulp between libms), the
        #     gathered shards of the weight hypernet are the reference's whole-tensor `_weight`
        with torch.no_grad():
            m.smoothquant.scale.data.copy_(ref_scale.to(dev).reshape(m.smoothquant.scale.shape))
            wq = m._weight.detach().contiguous()
            xin = m.smoothquant.scale_input(x.to(dev))
            cin, _, _ = m.input_casts(xin)
        full = P.gather_rows(wq, fout, world)
        res["w_digest_ok"] = digest(full) == expected["config5_whisper_small_encoder_layer/fc1/w"]["sha256"]
        res["sq_in_digest_ok"] = digest(xin) == expected["config5_whisper_small_encoder_layer/fc1/sq_in"]["sha256"]
        res["in_digest_ok"] = digest(cin) == expected["config5_whisper_small_encoder_layer/fc1/in"]["sha256"]
        # (c) the per-tensor MinMax observer over a row-sharded tensor: running min / max completed over the group
        obs = dmx.MinMaxObserver(qscheme=torch.per_tensor_symmetric).to(dev)
        obs.set_process_group(P.WORLD)
        obs(W[s:e].to(dev))
        res["minmax_ok"] = float(obs.min_val) == float(W.min()) and float(obs.max_val) == float(W.max())
        torch.cuda.synchronize()
        q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_smoothquant_calibration_of_a_row_sharded_linear_two_ranks_one_gpu():
    import torch.multiprocessing as mp

    world, port = 2, 29800 + (os.getpid() % 150)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sq_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r["same_on_all_ranks"] and r["equals_unsharded"] and r["exchange_matters"], r
        assert r["rel_vs_reference"] <= 4 * 2.0 ** -23, r
        assert r["w_digest_ok"] and r["sq_in_digest_ok"] and r["in_digest_ok"] and r["minmax_ok"], r
