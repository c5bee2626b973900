#!/bin/bash
# traffic of the exact-depth kernels (4300 x 4096 bf16: 512 x 17, partial last tile; 3072 x 4096: 512 x 12): FETCH_SIZE / WRITE_SIZE in separate passes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04/pmc_deep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run --output-format csv -- python3 $R/tools/probe_deep.py 4300 3072 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run --output-format csv -- python3 $R/tools/probe_deep.py 4300 3072 > $OUT/write.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
def counter(d, name):
    v = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bfp_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                v[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return v
fe, wr = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
with open(os.path.join(out, "summary.txt"), "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python3 tools/probe_deep.py 4300 3072`\n")
    fh.write("# FETCH_SIZE doubled (gfx950 counts 128-B requests of 16 B/lane streaming reads at 64 B: MI355X_MICROARCH.md), WRITE_SIZE as is; KB -> bytes\n")
    for k in sorted(fe):
        rd = 2 * sum(fe[k]) / len(fe[k]) * 1024
        ww = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))) * 1024
        fh.write(f"{k}\n    dispatches {len(fe[k])}  read {rd/1e6:.2f} MB  write {ww/1e6:.2f} MB per launch\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
rm -rf $OUT/fetch $OUT/write
