#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
TUNE_SET=blocks tools/tune_bfp 7 4096 4096 > $OUT/tune_blocks.txt 2>&1
cat $OUT/tune_blocks.txt
for rows in 1024 2048 3072; do echo "== rows $rows"; TUNE_SET=small tools/tune_bfp 5 $rows 4096; done > $OUT/tune_small.txt 2>&1
cat $OUT/tune_small.txt
