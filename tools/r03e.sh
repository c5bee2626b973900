#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_modules.py tests/test_gpu_elementwise.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-300
timeout 900 python3 tools/bench_ops.py --only "fixed_qdq,scale_channels" > $OUT/ops.txt 2>&1
cat $OUT/ops.txt
for mdl in opt125m llama whisper; do
  timeout 600 python3 bench.py --workload layer --model $mdl > $OUT/layer_$mdl.json 2> $OUT/layer_$mdl.err
  python3 -c "import json,sys; d=json.load(open('$OUT/layer_$mdl.json')); print('$mdl', d['layer_us'])"
done
