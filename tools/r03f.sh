#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/pytest2.log 2>&1; echo "rc=$?" >> $O/pytest2.log
tail -4 $O/pytest2.log | cut -c1-250
timeout 900 python3 tools/bench_ops.py > $O/ops2.txt 2>&1
timeout 600 python3 tools/bench_shapes.py > $O/shapes2.txt 2>&1
echo done
