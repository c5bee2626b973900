#!/bin/bash
# layer-level evidence: rocprofv3 kernel shares per model (eager, live weights) + a cProfile of the host side of one opt-125m layer
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mdl in opt125m llama whisper; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$mdl -o run --output-format csv -- python3 $R/bench.py --workload layer --model $mdl --layer-modes live > $OUT/prof_$mdl.log 2>&1
  python3 $R/tools/bench_layer.py --summarise $OUT/prof_$mdl > $OUT/layer_shares_$mdl.txt 2>&1
  rm -rf $OUT/prof_$mdl
  cat $OUT/layer_shares_$mdl.txt | cut -c1-200
done
cd $R
python3 - > $OUT/host_profile_opt125m.txt 2>&1 <<'PY'
import cProfile, pstats, io, sys, os, importlib.util, torch
sys.path.insert(0, os.getcwd())
spec = importlib.util.spec_from_file_location("bench_layer", "tools/bench_layer.py"); bl = importlib.util.module_from_spec(spec); spec.loader.exec_module(bl)
import dmx_compressor_amd as d
dev = torch.device("cuda", 0)
m, x, extra = bl.build_layer("opt125m", dev)
with torch.no_grad():
    d.nn.fold_weights_and_biases(m)
    for _ in range(20): m(x, *extra)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): m(x, *extra)
    torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(45); print(s.getvalue())
PY
head -75 $OUT/host_profile_opt125m.txt | cut -c1-180
