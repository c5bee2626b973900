#!/bin/bash
# tools/collect_profiles.sh <round tag, e.g. r06> — everything under profiles/ that is measured on the GPU box, in one go, from ONE library
# build.  In the build container first: `python tools/stamp.py --write` (records the commit and the SHA-256 of the built libraries; the
# GPU box has no .git).  Then through gpurun from the repo root; results land in gpurun_out/<tag>/ and are copied to profiles/ by hand.
# Every text file starts with the stamp line of tools/stamp.py --header (commit, library SHA-256, "the stamped build" or not); every
# bench.py JSON line carries the same as its `build` field; CSV files get the line as a leading '#' comment.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
STAMP="$(python3 tools/stamp.py --header)"
echo "$STAMP" > $OUT/STAMP.txt
txt() { local f=$1; shift; { echo "$STAMP"; echo "# command: $*"; "$@" 2>&1 | grep -v amdgpu.ids; } > $OUT/$f; }
# 1. bench lines: the driver's configuration (three runs), the default one, the sharded Llama workloads, the N > 1 path on real kernels with
#    the box's single GPU (ranks share cuda:0 over gloo), a 1-rank torchrun (RCCL init path), the self-launch path
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20.json 2> $OUT/bench.err
for i in 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tier2 > $OUT/bench_line_steps20_run$i.json 2>> $OUT/bench.err; done
python3 bench.py --no-cpu-baseline --no-tier2 > $OUT/bench_line_default.json 2>> $OUT/bench.err
python3 bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_world2_gloo.json 2>> $OUT/bench.err
python3 bench.py --gpus 8 --dist-backend gloo --nbuf 4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_world8_gloo.json 2>> $OUT/bench.err
python3 bench.py --gpus 8 --dist-backend gloo --workload llama-shard --op hypernet --layers 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_world8_gloo_llama_hypernet.json 2>> $OUT/bench.err
python3 bench.py --workload llama-shard --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_llama_hypernet.json 2>> $OUT/bench.err
python3 bench.py --workload llama-shard --op bfp --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_llama_bfp.json 2>> $OUT/bench.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-tier2 > $OUT/bench_line_torchrun1.json 2>> $OUT/bench.err
python3 bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline --no-tier2 > $OUT/bench_line_spawn1.json 2>> $OUT/bench.err
for mdl in opt125m llama whisper; do python3 bench.py --workload layer --model $mdl > $OUT/layer_$mdl.json 2>> $OUT/bench.err; done
# 2. rocprofv3 kernel statistics of the driver's command (second tier included: its kernels are in the same table), and the two HBM
#    traffic passes of the headline kernel (separate, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-resident --no-check > $OUT/prof_bench.log 2>&1
for mdl in opt125m llama whisper; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_layer_$mdl -o run --output-format csv -- python3 $R/bench.py --workload layer --model $mdl --layer-modes live > $OUT/prof_layer_$mdl.log 2>&1
  { echo "$STAMP"; python3 $R/tools/bench_layer.py --summarise $OUT/prof_layer_$mdl; } > $OUT/layer_shares_$mdl.txt 2>&1
  rm -rf $OUT/prof_layer_$mdl
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-resident --no-check --no-tier2 --replays 3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-resident --no-check --no-tier2 --replays 3 > $OUT/pmc_write.log 2>&1
cd $R
python3 - "$OUT" "$STAMP" <<'PY'
import csv, glob, json, os, sys
out, stamp = sys.argv[1], sys.argv[2]
def counter(d, name):
    v = []
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bfp_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return v
fe, wr = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
if fe and wr:
    rd, ww = 2 * sum(fe) / len(fe) * 1024, sum(wr) / len(wr) * 1024
    json.dump({"build": stamp,
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-resident --no-check --no-tier2 --replays 3`, "
                         f"mean over {len(fe)} / {len(wr)} dispatches of dmxq::bfp_rows_kernel (4096x4096 bf16)",
               "FETCH_SIZE_KB_raw": round(sum(fe) / len(fe), 2), "WRITE_SIZE_KB_raw": round(sum(wr) / len(wr), 2),
               "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane streaming reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16 B/lane stores",
               "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(ww), "hbm_bytes_per_launch": int(rd + ww),
               "algorithmic_bytes_per_launch": 67108864, "traffic_over_algorithmic": round((rd + ww) / 67108864, 4)},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
# kernel stats: keep the library's kernels only
for f in glob.glob(os.path.join(out, "prof_bench", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "dmxq" in r[0]]
    with open(os.path.join(out, "bench_kernel_stats.csv"), "w") as g:
        g.write(stamp + "\n# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-resident --no-check (second tier included)\n")
        csv.writer(g).writerows(keep)
PY
rm -rf $OUT/prof_bench $OUT/pmc_fetch $OUT/pmc_write
# 3. shape / op tables, counters of the second-tier kernels, host overhead
txt tier2.txt python3 tools/bench_tier2.py --no-layers --json $OUT/tier2.json
txt ops_roofline_table.txt python3 tools/bench_ops.py
txt secondary_shapes.txt python3 tools/bench_shapes.py
txt mid_shapes.txt python3 tools/bench_shapes.py --mid
txt row_ops.txt python3 tools/bench_rows.py
txt conv_shapes.txt python3 tools/bench_conv_shapes.py
txt small_tensor_ops.txt python3 tools/bench_small.py
txt shard_sets.txt python3 tools/bench_shard_sets.py
txt host_overhead.txt python3 tools/host_overhead.py
txt accuracy_table.txt python3 tools/accuracy_table.py
txt eager_profile_opt125m.txt python3 tools/profile_eager_layer.py --model opt125m
bash tools/collect_pmc.sh gpurun_out/$TAG/pmc "per-channel along last,group_size=128,group_minmax,channel_maxabs,bf16 score,SBFP12,rnd=3,histc,bfloat16->bfloat16 B=16 wl=8 sym rnd=2,scale_channels,layernorm,rmsnorm,softmax,unary,_cast,lut16,E4M3,block_dim=-2,feature map,hypernet,bfp_pack" > /dev/null 2>&1
ls -la $OUT
