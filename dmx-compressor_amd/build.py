"""Builds libdmxq.so (the C-ABI HIP library, include/dmxq.h) for gfx950 with hipcc, in-tree.

`python dmx-compressor_amd/build.py` or `dmx_compressor_amd.build.build()`.  hipcc cross-compiles without a
GPU.  The .so is git-ignored (history stays source-only) but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libdmxq.so")
SOURCES = ["bfp.hip", "bfp_cols.hip", "bfp_urows.hip", "blockfmt.hip", "bfp_pack.hip", "hypernet.hip", "elementwise.hip", "nm_mask.hip", "topk.hip", "reduce.hip", "approx.hip"]
# bit-exact fp32: no fast-math, no fma contraction; fp32 denormals stay on (gfx950 default)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
         "-fgpu-flush-denormals-to-zero" if False else "-fno-gpu-flush-denormals-to-zero"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "bfp_math.hpp"), os.path.join(CSRC, "bfp_rows.hpp"), os.path.join(CSRC, "stream.hpp"), os.path.join(HERE, "..", "include", "dmxq.h"), os.path.abspath(__file__)]
    hipcc = _hipcc()

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
