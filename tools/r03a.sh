#!/bin/bash
# round 3, first GPU pass: tests, the driver's bench line, a CLEAN kernel-stats profile (rotating launches only), tile sweep
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20.json 2> $OUT/bench.err
python3 bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_spawn1.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-resident --no-check > $OUT/prof_bench.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "prof_bench", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "dmxq" in r[0]]
    csv.writer(open(os.path.join(out, "bench_kernel_stats.csv"), "w")).writerows(keep)
PY
rm -rf $OUT/prof_bench
cat $OUT/bench_kernel_stats.csv
for rows in 4096 4100 4352 4608 5000 5120 6144 8192; do
  echo "== rows $rows" >> $OUT/tune_sweep.txt
  TUNE_SET=sweep timeout 120 tools/tune_bfp 5 $rows 4096 >> $OUT/tune_sweep.txt 2>&1
done
timeout 600 python3 tools/bench_shapes.py > $OUT/secondary_shapes.txt 2>&1
head -30 $OUT/secondary_shapes.txt
