#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 900 python3 tools/bench_ops.py --only "gelu,silu" > $OUT/ops.txt 2>&1
grep -v replaces $OUT/ops.txt
