#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_elementwise.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_gpu_seams.py tests/test_gpu_model_shapes.py tests/test_gpu_modules.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-250
timeout 600 python3 tools/bench_ops.py --only "float_qdq" > $OUT/ops.txt 2>&1
cat $OUT/ops.txt
