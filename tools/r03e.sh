#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_sparse_calib_approx.py tests/test_golden.py tests/test_gpu_model_shapes.py tests/test_gpu_round2.py tests/test_gpu_modules.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log | cut -c1-250
timeout 600 python3 tools/bench_ops.py --only "group_minmax,channel_maxabs" > $OUT/ops.txt 2>&1
cat $OUT/ops.txt
