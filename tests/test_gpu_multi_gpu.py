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
                                         "--replays", "3", "--no-cpu-baseline"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
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
    assert line["value"] > 0


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
