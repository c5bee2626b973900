#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_modules.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log | cut -c1-250
