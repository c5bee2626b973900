#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo "rc=$?" >> $OUT/pytest_all.log
tail -15 $OUT/pytest_all.log | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
