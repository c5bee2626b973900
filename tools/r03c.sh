#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_modules.py -m gpu -q -x > $OUT/pytest_r3.log 2>&1; echo "rc=$?" >> $OUT/pytest_r3.log
tail -40 $OUT/pytest_r3.log | cut -c1-300
for mdl in opt125m llama whisper; do
  timeout 600 python3 bench.py --workload layer --model $mdl > $OUT/layer_$mdl.json 2> $OUT/layer_$mdl.err
  tail -3 $OUT/layer_$mdl.err | cut -c1-300
  cat $OUT/layer_$mdl.json | cut -c1-1500
done
