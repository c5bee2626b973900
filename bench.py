#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: Gelements/s of fused Q->DQ, BFP[8|8]{16}(SN) ("BFP16,
group 16"), on a 4096x4096 bf16 tensor, and % of the MI355X HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|llama-shard] [--replays R]
  python bench.py --workload layer --model opt125m|llama|whisper    one configured model layer end to end (tools/bench_layer.py)
  python bench.py --gpus 2|4|8 ...   starts its own N ranks: with no WORLD_SIZE in the environment the parent (which never
                                     touches the GPU) runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
                                     --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>` as a CHILD process,
                                     forwards rank 0's one JSON line and exits with the child's code
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
                                     the driver's form: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment

Workloads (one process per GPU, no data-path collective: SURVEY.md §8e)
  c2 (default)   one "step" = one pass of the hot path (`dmxq_bfp_qdq`, ONE kernel launch through the C ABI) over this
                 rank's [4096, 4096] row shard of a global [N*4096, 4096] bf16 tensor (parallel.row_shards), already
                 resident in HBM.  At N = 1 that is BASELINE.json configs[1].  Per-GPU work is fixed -> "weak".
                 Steps rotate over NBUF distinct global tensors (>= 1.25 GiB per rank > 256 MiB Infinity Cache).
  llama-shard    one "step" = one pass over the 7 weight matrices of one Llama-3-8B decoder layer (BASELINE.json
                 configs[3]), each ROW-SHARDED over the N ranks; total work is fixed -> "strong".  --op hypernet
                 (default): 2:4 N:M mask -> BFP16_64 in one launch per weight (`dmxq_weight_hypernet`, the weight path of
                 modeling/nn/core.py:178-198); --op bfp: plain BFP16_16 (`dmxq_bfp_qdq`).
  Outside the timed region the output shards are RCCL-all_gathered once and rank 0 asserts bit-equality with its own
  whole-tensor result (shard -> op -> concat == op on the whole).

Timing: W eager warm-up steps; the K steps are enqueued either as K eager C-ABI calls (K <= 64) or as one replay of a
hipGraph captured from them (larger K; one untimed replay first: graph upload).  Then R wall-clock regions of exactly K
steps, each bracketed by barrier + synchronize on both sides, per region the MAX over ranks; `ms_per_step` / `value` come
from the MEDIAN region (`config.replays`, `config.replay_ms` list them all).  Then R HIP-event regions of the same K steps
for the roofline (events on the launch stream; a queued device-side delay in front of the first event keeps host enqueue
latency out of the events).

The JSON line carries
  roofline     : algorithmic bytes per launch / average launch duration, measured with HIP events on the launch stream
                 over the same timed replays, vs 8.0 TB/s peak HBM.  `kernel` is what the library's dispatcher reports
                 for this call (dmxq_bfp_qdq_describe), `traffic` is labelled with its source.
  cpu_baseline : three labelled legs on this box's host cores (rank 0, N = 1 only, bounded samples): the reference's
                 own compiled CPU kernel (oracle/_ref/quant_cpu.so) inside a restatement of the reference's
                 Format.cast loop ("reference"), and the C oracle port on all cores and on one thread ("port").
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ROWS = COLS = 4096
BLOCK, PRECISION = 16, 8
PEAK_HBM = 8.0e12  # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_ELEM = 4  # bf16 in + bf16 out (CastTo contract, numerical/cast.py:262,306)
# Llama-3-8B decoder layer (hidden 4096, 32 heads / 8 KV heads, intermediate 14336): [out_features, in_features]
LLAMA_LAYER = [("q_proj", 4096, 4096), ("k_proj", 1024, 4096), ("v_proj", 1024, 4096), ("o_proj", 4096, 4096),
               ("gate_proj", 14336, 4096), ("up_proj", 14336, 4096), ("down_proj", 4096, 14336)]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", choices=["c2", "replica", "llama-shard", "layer"], default="c2")
    ap.add_argument("--model", choices=["opt125m", "llama", "whisper"], default="llama",
                    help="--workload layer: which configured layer (BASELINE.json configs 3 / 4 / 5), see tools/bench_layer.py")
    ap.add_argument("--layer-modes", default="live,folded,unfused", help="--workload layer: which variants to time")
    ap.add_argument("--op", choices=["hypernet", "hypernet-each", "bfp"], default="hypernet",
                    help="llama-shard only.  hypernet: the rank's seven weight shards in ONE launch (dmxq_weight_hypernet_multi); "
                         "hypernet-each: one dmxq_weight_hypernet launch per weight (rounds 2-3); bfp: plain BFP16_16 per weight")
    ap.add_argument("--replays", type=int, default=15, help="timed replays of the K-step graph (median reported)")
    ap.add_argument("--nbuf", type=int, default=20, help="c2: distinct in/out buffer pairs (20 x 64 MiB = 1.25 GiB)")
    ap.add_argument("--layers", type=int, default=2, help="llama-shard: distinct layer copies rotated over")
    ap.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                    help="how the K steps are enqueued: one hipGraph replay, or K eager C-ABI calls.  auto = eager up to 64 "
                         "steps (a graph launch costs ~15 us on the GPU timeline before its first kernel: 0.8 us per step at "
                         "K = 20, measured), graph beyond (0.3 us per launch less than eager enqueueing)")
    ap.add_argument("--no-graph", action="store_true", help="same as --launch eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the gather + whole-tensor equality check")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--preroll", type=int, default=0, help="untimed launches queued right in front of each HIP-event region")
    ap.add_argument("--no-resident", action="store_true",
                    help="skip the cache-resident loop (one 64 MiB buffer pair, served by the Infinity Cache): a rocprofv3 "
                         "kernel-trace of the run then holds rotating-buffer launches only")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="harness transport for barrier / max-over-ranks / gather.  nccl (= RCCL, the default and what the driver "
                         "runs).  gloo: HARNESS-ONLY alternative that lets N ranks SHARE the box's GPUs (rank r -> cuda:(r mod "
                         "device_count)), so the whole N > 1 path -- shard, K launches per rank, max over ranks, gather, whole-tensor "
                         "bit compare -- runs on real HIP kernels on a 1-GPU lease (RCCL refuses two ranks per device); collectives "
                         "then carry host copies.  Throughput of such a run is NOT a scaling figure: the ranks share one GPU")
    ap.add_argument("--sync", choices=["block", "poll"], default="block",
                    help="how the end of a wall-clock region is observed: block = torch.cuda.synchronize() alone (the contract's form); "
                         "poll = spin on hipStreamQuery until the launch stream is drained, then torch.cuda.synchronize().  Measured "
                         "(tools/region_probe.py, profiles/r04_region_probe.txt): polling is 2-7 us per region SLOWER than the blocking "
                         "call, which already spins; hipStreamSynchronize, event waits and hipDeviceScheduleSpin change nothing.  Both "
                         "figures are in the line (`config.sync_alt`)")
    ap.add_argument("--preheat", type=int, default=2000,
                    help="untimed launches right before the timed regions (c2 only; ~22 ms of continuous load): the host-side input "
                         "generation leaves the GPU idle for seconds, and the first regions after that measured ~10 us slower")
    ap.add_argument("--input", choices=["portable", "device"], default="portable",
                    help="c2 inputs: portable = tests/_data.py's counter-based generator on the host (identical bits on every "
                         "machine: outputs are compared with the reference's committed digests); device = torch.randn on the GPU")
    ap.add_argument("--no-tier2", action="store_true",
                    help="c2 at N = 1 only: skip the second tier (tools/bench_tier2.py: the config 3 / 4 / 5 kernels and the three configured "
                         "layers, timed and oracle-checked after the headline measurement; `ops` / `layers` in the JSON line)")
    ap.add_argument("--tier2-only", default=None, help="comma-separated substrings of second-tier op names / config tags (c3,c4,c5)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (exercises the self-launch path)")
    return ap.parse_args()


def self_launch(args):
    """`bench.py --gpus N` without a launcher: run the N ranks as a CHILD `torch.distributed.run` and relay its output.
    Called before anything in this process touches HIP (importing torch does not), and as a child process, never an exec:
    a process that has initialised the GPU must not be replaced on this pool, and this way the rule cannot be broken by a
    later edit either."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    for ln in p.stdout:   # rank 0's JSON line goes to stdout, anything else the ranks or RCCL print to stderr
        out = sys.stdout if ln.lstrip().startswith("{") else sys.stderr
        out.write(ln)
        out.flush()
    sys.exit(p.wait())


def synth(seed, rows, cols, device):
    """N(0,1) * exp(2 * N(0,1)) in bf16: per-block exponent spread, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randn(rows, cols, generator=g, device=device)
    b = torch.randn(rows, cols, generator=g, device=device)
    return (a * torch.exp(2.0 * b)).to(torch.bfloat16)


def llama_seeds(t):
    """(weight seed, score seed) of the t-th weight of bench.py's llama-shard layer copy 0: what oracle/gen_golden_r6.py committed the
    reference's digests for (tests/golden/llama_shard_digests.json)"""
    return 7700 + t, 5000 + t


def slot_seed(rank, slot):
    """seed of rank `rank`'s rotation slot `slot` (c2): what oracle/gen_golden_r4.py committed the reference's digests for"""
    return 1000 * rank + slot


def portable_inputs(seeds, rows, cols):
    """c2 inputs from tests/_data.py's counter-based generator (splitmix64 on the linear index -> Box-Muller in fp64 -> bf16),
    kind "heavy" (N(0,1) * exp(4 N(0,1)): per-block exponent spread), computed on the host cores in index chunks: the same bits
    on every machine, so a run's outputs can be compared with the reference's committed digests (tests/golden/c2_digests.json)."""
    from concurrent.futures import ThreadPoolExecutor

    if os.path.join(ROOT, "tests") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _data import make_chunked

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    inner = max(1, min(16, cores))
    with ThreadPoolExecutor(max(1, min(8, cores // inner))) as ex:
        return list(ex.map(lambda sd: make_chunked("heavy", (rows, cols), sd, torch.bfloat16, workers=inner), seeds))


def _time_cpu(fn, seconds, max_n):
    fn()  # cold call (page-in, allocator)
    best, n, t_end = float("inf"), 0, time.perf_counter() + seconds
    while n < 2 or (time.perf_counter() < t_end and n < max_n):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
        n += 1
    return best, n


def _port_leg_main(argv):
    """`python bench.py --cpu-port-leg THREADS SECONDS TENSOR.pt` (internal): times the oracle's whole-tensor C port in THIS process,
    started with OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES in its environment (libgomp reads them when it is loaded: inside the
    bench process torch has loaded it long before)."""
    threads, seconds, path = int(argv[0]), float(argv[1]), argv[2]
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O

    x = torch.load(path)
    best, n = _time_cpu(lambda: O.bfp_cast(x, PRECISION, BLOCK).to(torch.bfloat16), seconds, 200)
    print(json.dumps({"best": best, "n": n, "threads": threads}))


def cpu_baseline(seconds):
    """Times the CPU paths on the host cores for the same 4096x4096 bf16 workload (three labelled legs).  The timed
    input is the bench's own rotation slot 0 of rank 0 (the counter-generated tensor, tests/_data.py: the same bits on every box);
    the OpenMP leg runs in a fresh process with its threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores), one thread per
    physical core -- in-process, with torch's libgomp already initialised and 2 SMT threads per core, the all-cores figure moved
    between 0.17 and 0.58 Gelements/s from box to box."""
    import subprocess
    import tempfile

    x = portable_inputs([slot_seed(0, 0)], ROWS, COLS)[0]
    legs = []
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    try:
        sys.path.insert(0, ref_dir)
        import quant_cpu  # the reference's C++ extension, compiled from its own sources by oracle/Makefile

        def ref_fn():
            # restatement of numerical/format.py:322-341 + cast.py:306 around the reference's native call
            xf = x.float()
            chunks = torch.split(xf.reshape(-1, COLS), BLOCK, dim=-1)
            out = torch.cat([quant_cpu.block_quantize_nearest(c.contiguous(), PRECISION, 0, True) for c in chunks], dim=-1)
            return out.reshape(ROWS, COLS).to(torch.bfloat16)

        best, n = _time_cpu(ref_fn, seconds, 50)
        legs.append({"kind": "reference", "value": round(ROWS * COLS / best / 1e9, 5), "unit": "Gelements/s",
                     "cores": torch.get_num_threads(),
                     "sample": f"{n} full passes over the bench's slot-0 tensor, 4096x4096 bf16 (min {best * 1e3:.1f} ms): reference "
                               "quant_cpu.block_quantize_nearest per [4096,16] chunk inside the reference's split/cat loop"})
    except Exception as e:  # _ref not built (never on the GPU box: the prebuilt .so travels with the snapshot)
        legs.append({"kind": "reference", "value": None, "error": repr(e)[:200]})
    try:
        logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:   # physical cores among the allowed CPUs (one OpenMP thread each)
            sib = set()
            for c in (os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else range(logical)):
                with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                    sib.add(f.read().strip())
            physical = max(1, len(sib))
        except OSError:
            physical = logical
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "x.pt")
            torch.save(x, path)
            for threads, budget in ((physical, min(4.0, seconds)), (1, min(6.0, seconds))):
                env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_DYNAMIC="false")
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-port-leg", str(threads), str(budget), path],
                                   env=env, capture_output=True, text=True, timeout=600)
                out = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not out:
                    raise RuntimeError(f"port leg failed: {r.stderr[-300:]}")
                d = json.loads(out[-1])
                legs.append({"kind": "port", "value": round(ROWS * COLS / d["best"] / 1e9, 5), "unit": "Gelements/s", "cores": threads,
                             "sample": f"{d['n']} full passes over the bench's slot-0 tensor, 4096x4096 bf16 (min {d['best'] * 1e3:.1f} ms): "
                                       f"oracle/oracle.c whole-tensor port in a fresh process, OpenMP threads = {threads} pinned "
                                       f"(OMP_PROC_BIND=close, OMP_PLACES=cores; {physical} physical / {logical} logical CPUs allowed)"})
    except Exception as e:
        legs.append({"kind": "port", "value": None, "error": repr(e)[:200]})
    head = next((l for l in legs if l.get("value")), legs[0])
    return {"value": head["value"], "unit": "Gelements/s", "cores": head.get("cores"), "kind": head["kind"],
            "sample": head.get("sample"), "legs": legs}


class Workload:
    """What one step launches on this rank, and how to check it.  All launches go through the C ABI (include/dmxq.h)."""

    def __init__(self, args, L, lib, rank, world, dev):
        from dmx_compressor_amd import parallel

        self.args, self.L, self.lib, self.rank, self.world, self.dev = args, L, lib, rank, world, dev
        self.P = parallel
        bf16 = lib.BF16
        self.calls = []   # per rotation slot: list of (fn_name, tuple_of_args_without_stream)
        self.bytes_per_step = 0
        self.elems_per_step_rank = 0
        if args.workload in ("c2", "replica"):
            self.scaling = "weak"
            self.global_rows = ROWS * world
            self.shard = parallel.row_shards(self.global_rows, world)[rank]
            rows = self.shard[1] - self.shard[0]
            self.seeds = [slot_seed(rank, i) for i in range(args.nbuf)]
            if args.input == "portable":
                self.host_ins = portable_inputs(self.seeds, rows, COLS)
                self.ins = [t.to(dev) for t in self.host_ins]
            else:
                self.host_ins = None
                self.ins = [synth(sd, rows, COLS, dev) for sd in self.seeds]
            self.outs = [torch.empty_like(t) for t in self.ins]
            for i in range(args.nbuf):
                self.calls.append([("dmxq_bfp_qdq", (ctypes.c_void_p(self.ins[i].data_ptr()), ctypes.c_void_p(self.outs[i].data_ptr()),
                                                     bf16, bf16, rows, COLS, 1, BLOCK, PRECISION, lib.ROUND_NEAREST, 1, 0))])
            self.elems_per_step_rank = rows * COLS
            self.bytes_per_step = BYTES_PER_ELEM * rows * COLS
            self.launches_per_step = 1
            self.describe_args = (bf16, bf16, rows, COLS, 1, BLOCK, PRECISION, lib.ROUND_NEAREST, 1)
            self.name = ("BFP[8|8]{16}(SN) fused Q->DQ, 4096x4096 bf16 -> bf16 per GPU, block_dim=-1, RNE (BASELINE.json "
                         "configs[1]); rank r owns rows [4096 r, 4096 (r+1)) of a global [N*4096, 4096] tensor")
        else:
            self.scaling = "strong"
            hyper = args.op in ("hypernet", "hypernet-each")
            B = 64 if hyper else 16
            self.layers = []
            self.keep = []   # host descriptor arrays of the multi-tensor calls (must outlive the launches that read them)
            if os.path.join(ROOT, "tests") not in sys.path:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
            for c in range(args.layers):
                layer = []
                for t, (nm, rows, cols) in enumerate(LLAMA_LAYER):
                    s, e = parallel.row_shards(rows, world)[rank]
                    if c == 0 and args.input == "portable":
                        # layer copy 0 (the one that is checked): every rank generates ITS rows of the global tensor with the counter-based
                        # generator of tests/_data.py (element = f(seed, linear index)): the same bits on every machine, so each rank's
                        # shards can be compared with the SHA-256 of what the REFERENCE's Sparsify -> CastTo makes of those rows
                        # (tests/golden/llama_shard_digests.json, oracle/gen_golden_r6.py) -- round 6
                        from _data import make_chunked
                        ws, ss = llama_seeds(t)
                        w = make_chunked("heavy", (e - s, cols), ws, torch.bfloat16, start=s * cols).to(dev) if e > s else torch.empty(0, cols, dtype=torch.bfloat16, device=dev)
                        sc = None
                        if hyper:
                            sc = (make_chunked("uniform", (e - s, cols), ss, torch.bfloat16, start=s * cols).to(dev) if e > s
                                  else torch.empty(0, cols, dtype=torch.bfloat16, device=dev))
                        layer.append((nm, rows, cols, (s, e), w, sc, torch.empty_like(w)))
                        continue
                    # (other copies, --input device: every rank synthesises the SAME global tensor on the device and keeps its rows)
                    w_full = synth(77 + 10 * c + t, rows, cols, dev)
                    w = w_full[s:e].clone()
                    sc = None
                    if hyper:
                        g = torch.Generator(device=dev).manual_seed(5000 + 10 * c + t)
                        sc = torch.rand(rows, cols, generator=g, device=dev).to(torch.bfloat16)[s:e].clone()
                    del w_full
                    layer.append((nm, rows, cols, (s, e), w, sc, torch.empty_like(w)))
                self.layers.append(layer)
            torch.cuda.empty_cache()
            for layer in self.layers:
                cl = []
                if args.op == "hypernet":   # the whole layer's shards in one launch
                    live = [t for t in layer if t[3][1] > t[3][0]]
                    descs = (lib.HypernetDesc * len(live))()
                    for d, (nm, rows, cols, (s, e), w, sc, out) in zip(descs, live):
                        d.w, d.score, d.sq_scale, d.out, d.rows, d.L = w.data_ptr(), sc.data_ptr(), None, out.data_ptr(), e - s, cols
                    self.keep.append(descs)
                    if live:
                        cl.append(("dmxq_weight_hypernet_multi", (descs, len(live), bf16, bf16, 2, 4, bf16, B, PRECISION, 1)))
                    self.calls.append(cl)
                    continue
                for nm, rows, cols, (s, e), w, sc, out in layer:
                    n = (e - s)
                    if n == 0:
                        continue
                    if hyper:
                        cl.append(("dmxq_weight_hypernet", (ctypes.c_void_p(w.data_ptr()), bf16, ctypes.c_void_p(sc.data_ptr()), bf16, 2, 4,
                                                            None, ctypes.c_void_p(out.data_ptr()), bf16, n, cols, B, PRECISION, 1)))
                    else:
                        cl.append(("dmxq_bfp_qdq", (ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()), bf16, bf16, n, cols, 1, B,
                                                    PRECISION, lib.ROUND_NEAREST, 1, 0)))
                self.calls.append(cl)
            per_elem = 6 if hyper else 4  # w + score + out | in + out, all bf16
            mine = sum((e - s) * cols for _, _, cols, (s, e), *_ in self.layers[0])
            self.elems_per_step_rank = mine
            self.bytes_per_step = per_elem * mine
            self.launches_per_step = max(1, len(self.calls[0]))
            self.describe_args = None
            self.name = ("Llama-3-8B decoder-layer weights (q,k,v,o,gate,up,down = 218.1 M elements, bf16), each row-sharded over "
                         "the N ranks; " + ("BTOPK{2:4,-1} mask -> BFP[8|8]{64}(SN) in one launch per weight (dmxq_weight_hypernet), "
                                            "score bf16" if args.op == "hypernet-each" else
                                            "BTOPK{2:4,-1} mask -> BFP[8|8]{64}(SN), the rank's seven shards in ONE launch (dmxq_weight_hypernet_multi), "
                                            "score bf16" if hyper else "BFP[8|8]{16}(SN) Q->DQ (dmxq_bfp_qdq)")
                         + " (BASELINE.json configs[3])")
        self.total_elems_per_step = None  # filled by main (sum over ranks)

    def launch(self, slot, stream_ptr):
        for name, a in self.calls[slot % len(self.calls)]:
            rc = getattr(self.L, name)(*a, stream_ptr)
            if rc != 0:
                raise RuntimeError(f"{name} failed: {rc}")

    def check(self, dist):
        """Outside the timed region.  (1) EVERY rotation slot is launched once more and its output compared, bit for bit, with
        the CPU oracle (oracle/oracle.c: the checker, never the thing measured) and -- for the portable inputs -- with the SHA-256
        the REFERENCE's own CastTo produced for that slot in the build container (tests/golden/c2_digests.json, written by
        oracle/gen_golden_r4.py).  (2) The output shards of slot 0 are gathered (harness-only all_gather) and rank 0 compares them
        with its own whole-tensor launch: shard -> op -> concat must equal op on the whole tensor."""
        import dmx_compressor_amd as d

        for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle as O
        from _data import bits_equal, sha256_bits

        P, rank, world = self.P, self.rank, self.world
        stream = torch.cuda.current_stream(self.dev)
        sp = ctypes.c_void_p(stream.cuda_stream)
        for i in range(len(self.calls)):
            self.launch(i, sp)
        torch.cuda.synchronize(self.dev)
        if self.args.workload in ("c2", "replica"):
            gold = {}
            gp = os.path.join(ROOT, "tests", "golden", "c2_digests.json")
            if self.host_ins is not None and os.path.exists(gp):
                gold = json.load(open(gp))["slots"]
            n, n_ref, n_foreign = len(self.ins), 0, 0
            for i in range(n):
                x = self.host_ins[i] if self.host_ins is not None else self.ins[i].cpu()
                got = self.outs[i].cpu()
                bad = bits_equal(got, O.bfp_cast(x, PRECISION, BLOCK).to(torch.bfloat16))
                assert bad == 0, f"rank {rank} slot {i}: {bad} elements differ from oracle.bfp_cast"
                g = gold.get(str(self.seeds[i]))
                if g is not None:
                    if sha256_bits(x) != g["input_sha256"]:
                        n_foreign += 1      # this host's libm rounded some fp64 log / cos differently: the oracle check stands
                    else:
                        assert sha256_bits(got) == g["output_sha256"], f"rank {rank} slot {i}: output differs from the reference's digest"
                        n_ref += 1
            msg = f"{n}/{n} slots == oracle.bfp_cast (bit-exact)"
            if gold:
                # every rank asserted its own slots above; what rank 0 reports is the MINIMUM over ranks of the slots that matched the
                # reference's digests (round 6: digests for the 20 slots of all eight ranks)
                lo = torch.tensor([float(n_ref)], dtype=torch.float64, device="cpu" if self.args.dist_backend == "gloo" else self.dev)
                if dist is not None and world > 1:
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                msg += f"; {int(lo[0])}/{n} slots == SHA-256 of the reference's CastTo output (tests/golden/c2_digests.json)"
                if n_foreign:
                    msg += f" ({n_foreign} inputs differ from the committed input digests on this host)"
            if world > 1:
                full_in = P.gather_rows(self.ins[0], self.global_rows, world)
                full_out = P.gather_rows(self.outs[0], self.global_rows, world)
                if rank == 0:
                    whole = d.CastTo(format="BFP[8|8]{16}(SN)")(full_in)
                    assert torch.equal(whole.view(torch.int16), full_out.view(torch.int16)), \
                        "gathered shard outputs differ from the whole-tensor CastTo result"
                msg = f"all_gather of {world} row shards == whole-tensor CastTo on rank 0 (bit-exact); every rank: " + msg
            return msg
        hyper = self.args.op in ("hypernet", "hypernet-each")
        B = 64 if hyper else 16
        # the reference's own output for these rows (tests/golden/llama_shard_digests.json: SHA-256 per eighth of every weight's rows, so
        # that the shards of N = 1, 2, 4, 8 ranks are unions of digested pieces)
        gold, n_ref, n_pieces = None, 0, 0
        gp = os.path.join(ROOT, "tests", "golden", "llama_shard_digests.json")
        if self.args.input == "portable" and os.path.exists(gp) and world in (1, 2, 4, 8):
            gold = json.load(open(gp))["tensors"]
        if gold is not None:
            key = "hypernet_sha256" if hyper else "bfp_sha256"
            for nm, rows, cols, (s, e), w, sc, out in self.layers[0]:
                piece = rows // 8
                for k in range(s // piece, e // piece):
                    g = gold[nm]["pieces"][k]
                    a, b = k * piece - s, (k + 1) * piece - s
                    n_pieces += 1
                    if sha256_bits(w[a:b]) != g["w_sha256"] or (hyper and sha256_bits(sc[a:b]) != g["score_sha256"]):
                        continue    # this host's libm rounded some fp64 log / cos differently: the oracle check below stands
                    assert sha256_bits(out[a:b]) == g[key], f"rank {rank} {nm} rows [{k * piece}, {(k + 1) * piece}): output differs from the reference's digest"
                    n_ref += 1
        for nm, rows, cols, (s, e), w, sc, out in self.layers[0]:
            if e > s:   # this rank's shard against the oracle composed like the reference (sparse.py:287-301 -> format.py:304-343)
                wc = w.cpu()
                want = O.bfp_cast(O.sparsify(wc, sc.cpu(), 2, 4) if hyper else wc, PRECISION, B).to(torch.bfloat16)
                bad = bits_equal(out.cpu(), want)
                assert bad == 0, f"rank {rank} {nm}: {bad} elements differ from the oracle chain"
            full_w = P.gather_rows(w, rows, world)
            full_o = P.gather_rows(out, rows, world)
            full_s = P.gather_rows(sc, rows, world) if hyper else None
            if rank == 0:
                if hyper:
                    whole = d.ops.weight_hypernet(full_w, PRECISION, 64, True, full_s, 2, 4)
                    chain = d.ops.bfp_qdq(d.ops.nm_sparsify(full_w, full_s, 2, 4), PRECISION, 64)
                    assert torch.equal(whole.view(torch.int16), chain.view(torch.int16)), f"{nm}: fused != mask -> BFP chain"
                else:
                    whole = d.ops.bfp_qdq(full_w, PRECISION, 16)
                assert torch.equal(whole.view(torch.int16), full_o.view(torch.int16)), f"{nm}: gathered shards differ from whole tensor"
            del full_w, full_o, full_s
        msg = (f"7 weights: all_gather of {world} row shards == whole-tensor result on rank 0 (bit-exact); every rank: its shards == "
               "oracle " + ("sparsify -> bfp_cast" if hyper else "bfp_cast") + " (bit-exact)")
        if gold is not None:
            t = torch.tensor([float(n_ref), float(n_pieces)], dtype=torch.float64, device="cpu" if self.args.dist_backend == "gloo" else self.dev)
            if dist is not None and world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            msg += (f"; {int(t[0])}/{int(t[1])} row pieces (eighths of the 7 weights, all ranks) == SHA-256 of the reference's "
                    + ("Sparsify -> CastTo" if hyper else "CastTo") + " output (tests/golden/llama_shard_digests.json)")
        return msg


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        have = torch.cuda.device_count()   # does not initialise HIP on this image
        if args.gpus > have and not (args.dist_backend == "gloo" and have >= 1):
            print(f"bench.py --gpus {args.gpus}: this box has {have} GPU(s)", file=sys.stderr)
            sys.exit(2)
        self_launch(args)
    if args.workload == "layer":   # one configured model layer end to end (tools/bench_layer.py); single GPU, its own JSON line
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_layer", os.path.join(ROOT, "tools", "bench_layer.py"))
        bl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bl)
        if not torch.cuda.is_available():
            print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
            sys.exit(2)
        steps = args.steps if args.steps != 2000 else 20
        print(json.dumps(bl.run(args.model, steps, min(args.warmup, 10), modes=tuple(args.layer_modes.split(",")))), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world   # under a launcher the environment is authoritative
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    if args.dist_backend == "gloo":   # harness-only: ranks may share a GPU (see --dist-backend)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "WORLD_SIZE" in os.environ:   # also for a 1-rank launch: the same RCCL code path as N > 1
        import torch.distributed as dist

        # RCCL prints a version banner on STDOUT when its communicator comes up; rank 0's JSON line must stay alone there
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dist_backend == "gloo":
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
            dist.barrier()                                  # (communicator creation is lazy without this)
            torch.cuda.synchronize(dev)
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    from dmx_compressor_amd import _lib

    L = _lib.lib()  # raises if libdmxq.so is missing: no fallback
    wl = Workload(args, L, _lib, rank, world, dev)
    K, R = args.steps, max(1, args.replays)
    if args.no_graph:
        args.launch = "eager"
    if args.launch == "auto":
        args.launch = "eager" if K * wl.launches_per_step <= 64 else "graph"
    args.no_graph = args.launch == "eager"

    stream = torch.cuda.Stream(device=dev)
    sp = ctypes.c_void_p(stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    walls, evs = [], []
    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            wl.launch(i, sp)
        graph = None
        if not args.no_graph:
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            # thread_local: with RCCL initialised (N > 1) its watchdog thread polls events while we capture; only calls
            # made by THIS thread belong to the capture
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                for i in range(K):
                    wl.launch(i, sp)
            graph.replay()  # untimed: the first replay uploads the graph
            torch.cuda.synchronize(dev)
        def run_k():
            if graph is not None:
                graph.replay()
            else:
                for i in range(K):
                    wl.launch(i, sp)

        # (a) whole-job wall clock: R timed regions of exactly K steps, barrier + synchronize on both sides.  A region carries
        # ~12-16 us that are not kernel time (tools/region_probe.py: K = 1 takes 24.4 us for one 10.9 us kernel; K = 20 ->
        # 231 us, K = 40 -> 450 us: 10.93 us per step + 13 us): the first launch's way to an idle queue and the completion's way
        # back to the host.  `--sync poll` observes the end by spinning on hipStreamQuery instead; it measured slower than the
        # blocking call (which spins itself); the line carries both (`config.sync_alt`).
        def region(poll):
            barrier()
            t0 = time.perf_counter()
            run_k()
            if poll:
                while not stream.query():
                    pass
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            barrier()
            return t1 - t0

        poll = args.sync == "poll"
        if args.workload in ("c2", "replica"):
            for i in range(args.preheat):   # untimed: the GPU sat idle while the host generated and checked inputs
                wl.launch(i, sp)
        for _ in range(10):          # untimed regions
            region(poll)
        for _ in range(R):
            walls.append(region(poll))
        walls_alt = [region(not poll) for _ in range(R)]
        # (a') the same regions with 10 K steps (eager launches only): a region carries ~13 us that are not kernel time, 0.65 us per
        # step at the driver's K = 20 and 0.07 at 200 -- reported side by side, never as `value`
        walls_10x = []
        if graph is None and K * 10 * wl.launches_per_step <= 4096:
            def region_10x():
                barrier()
                t0 = time.perf_counter()
                for i in range(10 * K):
                    wl.launch(i, sp)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                barrier()
                return t1 - t0
            region_10x()
            walls_10x = [region_10x() for _ in range(max(3, R // 3))]
        # (b) the kernels' own time with HIP events on the launch stream, over R more regions of the same K steps.  A
        # short device-side delay is queued in front of the first event so that the host has finished enqueueing the
        # region before the GPU reaches it: the events then bracket back-to-back kernel execution and not the
        # host's graph-launch latency (~15 us per replay, i.e. ~0.8 us per step at K = 20, which (a) rightly includes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        torch.cuda._sleep(2_000_000)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        cycles_per_us = 2_000_000 / max(e0.elapsed_time(e1) * 1e3, 1e-3)
        delay = int(cycles_per_us * (60.0 if graph is not None else 15.0 * min(K, 200)))
        for _ in range(R):
            barrier()
            torch.cuda._sleep(delay)
            for i in range(args.preroll):
                wl.launch(K + i, sp)
            e0.record(stream)
            run_k()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            evs.append(e0.elapsed_time(e1) / 1e3)
        barrier()

    cdev = "cpu" if args.dist_backend == "gloo" else dev
    if walls_10x:
        t10 = torch.tensor(walls_10x, device=cdev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t10, op=dist.ReduceOp.MAX)
        walls_10x = t10.tolist()
    t = torch.tensor([walls, evs, walls_alt], device=cdev, dtype=torch.float64)
    elems = torch.tensor([float(wl.elems_per_step_rank)], device=cdev, dtype=torch.float64)
    # every rank's OWN median region and element count, side by side with the max-over-ranks figure (round 6): a straggler shows as a
    # low per-rank minimum instead of only lowering the total
    mine = torch.tensor([statistics.median(walls), statistics.median(evs), float(wl.elems_per_step_rank)], device=cdev, dtype=torch.float64)
    per_rank = [mine]
    if dist is not None:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # per replay: the slowest rank
        dist.all_reduce(elems, op=dist.ReduceOp.SUM)
    walls, evs, walls_alt = t[0].tolist(), t[1].tolist(), t[2].tolist()
    wall_med, ev_med = statistics.median(walls), statistics.median(evs)
    total_elems = float(elems[0])

    # cache-resident rate (one buffer pair, 64 MiB < 256 MiB Infinity Cache): always on record, never `value`
    resident = None
    if args.workload in ("c2", "replica") and not args.no_resident:
        with torch.cuda.stream(stream):
            for _ in range(50):
                wl.launch(0, sp)
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record(stream)
            for _ in range(300):
                wl.launch(0, sp)
            r1.record(stream)
            torch.cuda.synchronize(dev)
        resident = wl.elems_per_step_rank * world / (r0.elapsed_time(r1) / 300 * 1e-3) / 1e9

    checked = None if args.no_check else wl.check(dist)

    if rank == 0:
        ms_per_step = wall_med * 1e3 / K
        value = total_elems / (wall_med / K) / 1e9
        step_s = ev_med / K
        launch_s = step_s / wl.launches_per_step
        achieved = wl.bytes_per_step / step_s
        traffic, traffic_src = None, None
        for fn in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tp = os.path.join(ROOT, "profiles", fn)
            if os.path.exists(tp) and args.workload != "llama-shard":
                try:
                    traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
                    traffic_src = f"profiles/{fn} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, committed; NOT measured in this run)"
                    break
                except Exception:
                    pass
        kernel = None
        if wl.describe_args is not None and hasattr(L, "dmxq_bfp_qdq_describe"):
            buf = ctypes.create_string_buffer(256)
            if L.dmxq_bfp_qdq_describe(*wl.describe_args, 1, buf, 256) == 0:
                kernel = buf.value.decode()
        line = {
            "metric": "Gelements/s fused Q->DQ (BFP16, group=16) on 4096x4096 bf16" if args.workload != "llama-shard"
                      else "Gelements/s over row-sharded Llama-3-8B layer weights",
            "value": round(value, 2), "unit": "Gelements/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 6), "higher_is_better": True,
            "scaling": wl.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name,
                       "buffers": (f"{args.nbuf} rotating in/out pairs = {args.nbuf * 2 * ROWS * COLS * 2 / 2**30:.2f} GiB per GPU "
                                   "(> 256 MiB Infinity Cache)") if args.workload != "llama-shard"
                                  else f"{args.layers} rotating layer copies ({wl.bytes_per_step / 2**20:.0f} MiB touched per step on this rank)",
                       "launch": "K eager C-ABI calls per region" if args.no_graph else "one hipGraph replay of the K steps' C-ABI launches per region",
                       "launches_per_step": wl.launches_per_step,
                       "replays": R,
                       "replay_ms": [round(w * 1e3, 4) for w in walls],
                       "sync": ("region end = hipStreamQuery spin until the stream is drained, then torch.cuda.synchronize()"
                                if args.sync == "poll" else "region end = blocking torch.cuda.synchronize()"),
                       "sync_alt": {"sync": "block" if args.sync == "poll" else "poll",
                                    "ms_per_step": round(statistics.median(walls_alt) * 1e3 / K, 6),
                                    "value": round(total_elems / (statistics.median(walls_alt) / K) / 1e9, 2)},
                       "at_10x_steps": ({"steps": 10 * K, "ms_per_step": round(statistics.median(walls_10x) * 1e3 / (10 * K), 6),
                                         "value": round(total_elems / (statistics.median(walls_10x) / (10 * K)) / 1e9, 2),
                                         "note": "the same eager launches in regions of 10 K steps: the ~13 us a region carries besides "
                                                 "kernel time weigh a tenth as much; informative, `value` is the K-step figure"}
                                        if walls_10x else None),
                       "dist_backend": (args.dist_backend + (" (harness-only transport; ranks share GPUs: not a scaling figure)"
                                                             if args.dist_backend == "gloo" else " (RCCL)")) if dist is not None else None,
                       "input": ("tests/_data.py make('heavy', (4096, 4096), seed = 1000 rank + slot, bf16): counter-based, host-generated"
                                 if args.input == "portable" else "torch.randn on the device") if args.workload != "llama-shard"
                                else ("layer copy 0: tests/_data.py make('heavy' | 'uniform', rows of the shard, seed = 7700 + t | 5000 + t, bf16), counter-based, "
                                      "host-generated; other copies: torch.randn on the device" if args.input == "portable" else "torch.randn on the device"),
                       "timing": "one untimed replay, then R wall-clock regions of exactly K steps (barrier+sync, K steps, sync; "
                                 "max over ranks; median region -> ms_per_step, value) and R HIP-event regions of the same K steps "
                                 "(events on the launch stream, a queued device-side delay in front so that host launch latency is "
                                 "not inside the events; median region -> roofline)",
                       "per_gpu_elements_per_step": wl.elems_per_step_rank,
                       "per_rank": {"value": [round(float(p[2]) / (float(p[0]) / K) / 1e9, 2) if float(p[2]) > 0 else 0.0 for p in per_rank],
                                    "value_min": round(min((float(p[2]) / (float(p[0]) / K) / 1e9) for p in per_rank), 2),
                                    "value_max": round(max((float(p[2]) / (float(p[0]) / K) / 1e9) for p in per_rank), 2),
                                    "event_us_per_step": [round(float(p[1]) * 1e6 / K, 3) for p in per_rank],
                                    "note": "each rank's own elements per step / its own median wall-clock region (Gelements/s); `value` is "
                                            "all ranks' elements / the per-region maximum over ranks"},
                       "check": checked},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                         "frac": round(achieved / PEAK_HBM, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "kernel_source": "dmxq_bfp_qdq_describe (the dispatcher's own choice for this call)" if kernel else None,
                         "algorithmic_bytes_per_launch": wl.bytes_per_step // wl.launches_per_step if wl.launches_per_step == 1 else None,
                         "algorithmic_bytes_per_step": wl.bytes_per_step,
                         "avg_launch_us": round(launch_s * 1e6, 3),
                         "event_us_per_step": [round(e * 1e6 / K, 3) for e in evs]},
        }
        if resident is not None:
            line["cache_resident_value"] = round(resident, 2)
        try:   # which library this line was measured on (tools/stamp.py: commit recorded in the build container, SHA-256 recomputed here)
            import importlib.util
            spec = importlib.util.spec_from_file_location("dmxq_stamp", os.path.join(ROOT, "tools", "stamp.py"))
            st = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(st)
            line["build"] = st.current()
        except Exception as e:   # noqa: BLE001
            line["build"] = {"error": repr(e)[:120]}
        if world == 1 and args.workload == "c2" and not args.no_tier2:
            # the second tier: configs 3 / 4 / 5's kernels and layers, AFTER the headline measurement above, in the
            # same process and the same driver-observed run; never raises (a failure is recorded in the entry it belongs to)
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("bench_tier2", os.path.join(ROOT, "tools", "bench_tier2.py"))
                t2 = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(t2)
                del wl.ins, wl.outs
                torch.cuda.empty_cache()
                line.update(t2.run(dev, args.tier2_only.split(",") if args.tier2_only else None,
                                   log=lambda m: print("[tier2] " + m, file=sys.stderr, flush=True)))
            except Exception as e:   # noqa: BLE001
                line["ops"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-port-leg":
        _port_leg_main(sys.argv[2:])
    else:
        main()
