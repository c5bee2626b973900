#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_act_cast.py -m gpu -q > $OUT/pytest_act.log 2>&1; echo "rc=$?" >> $OUT/pytest_act.log
tail -60 $OUT/pytest_act.log | cut -c1-300
timeout 900 python3 tools/bench_ops.py > $OUT/ops_roofline_table.txt 2>&1
cat $OUT/ops_roofline_table.txt
