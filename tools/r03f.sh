#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_modules.py tests/test_gpu_model_shapes.py tests/test_gpu_multi_and_shapes.py tests/test_gpu_elementwise.py tests/test_gpu_plan_branches.py -m gpu -q -x > $O/pytest3.log 2>&1; echo "rc=$?" >> $O/pytest3.log
tail -4 $O/pytest3.log | cut -c1-250
timeout 900 python3 tools/bench_ops.py --only "hypernet" > $O/ops_hn.txt 2>&1
cat $O/ops_hn.txt
