#!/bin/bash
# tools/collect_pmc.sh <out_dir> <bench_ops --only filter> — three rocprofv3 counter passes (SQ shares, FETCH_SIZE, WRITE_SIZE;
# separate passes with --kernel-trace only, as the pool requires) around tools/bench_ops.py, then the per-kernel summary.
set -u
OUT=$1; ONLY=$2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY -d $R/$OUT/sq -o run --output-format csv -- python3 $R/tools/bench_ops.py --iters 20 --only "$ONLY" > $R/$OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/fetch -o run --output-format csv -- python3 $R/tools/bench_ops.py --iters 20 --only "$ONLY" > $R/$OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/write -o run --output-format csv -- python3 $R/tools/bench_ops.py --iters 20 --only "$ONLY" > $R/$OUT/write.log 2>&1
{ python3 $R/tools/stamp.py --header; echo "# rocprofv3 --kernel-trace --pmc <SQ set | FETCH_SIZE | WRITE_SIZE> (three separate passes) -- python3 tools/bench_ops.py --iters 20 --only '$ONLY'"; python3 $R/tools/pmc_summary.py $R/$OUT/sq $R/$OUT/fetch $R/$OUT/write; } > $R/$OUT/summary.txt 2>&1
# keep only the small summaries (the raw CSVs of a torch process are tens of MB)
rm -rf $R/$OUT/sq $R/$OUT/fetch $R/$OUT/write
cat $R/$OUT/summary.txt
