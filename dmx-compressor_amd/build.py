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
TORCH_LIB = os.path.join(HERE, "lib", "dmxq_torch.so")
# "file.hip" or "file.hip#N": the file compiled with -DDMXQ_EW_PART=N into file_pN.o (elementwise.hip: three objects, approx.hip: two, in parallel)
SOURCES = ["bfp.hip#1", "approx.hip#3", "approx.hip#5", "approx.hip#1", "approx.hip#2", "approx.hip#4", "elementwise.hip#2", "elementwise.hip#1", "elementwise.hip#3", "bfp_cols.hip#1", "bfp_cols.hip#2", "bfp_cols.hip#3", "bfp.hip#2", "bfp.hip#3", "bfp_urows.hip", "bfp_smallinner.hip", "bfp_slab.hip", "blockfmt.hip", "bfp_pack.hip", "hypernet.hip", "hypernet_multi.hip", "nm_mask.hip", "topk.hip", "reduce.hip", "unary.hip", "act_cast.hip", "lut16.hip", "fixed_multi.hip", "rope.hip"]
# bit-exact fp32: no fast-math, no fma contraction; fp32 denormals stay on (gfx950 default).
# --offload-compress (round 5): the gfx950 code objects are stored zstd-compressed inside the fat binary and unpacked by the HIP runtime at
# load: libdmxq.so 88 MB -> 17 MB (what a gpurun snapshot pushes, what a wheel would ship), load time and kernels unchanged
# (profiles/r05_lib_size.txt).
# -amdgpu-kernarg-preload-count=16 (round 5): the first 16 dwords of a kernel's arguments arrive in SGPRs with the wave (gfx940+ kernarg
# preloading) instead of through s_load + s_waitcnt ahead of the workgroup's first data load: +0.9 % on average over the op table, up
# to +5 % (same-lease A/B of two builds, profiles/r05_kernarg_preload.txt) -- "nothing may precede a workgroup's first load" applied
# to the one thing every kernel did first.
FLAGS = ["--offload-arch=gfx950", "--offload-compress", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-O3", "-fPIC", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
         "-fgpu-flush-denormals-to-zero" if False else "-fno-gpu-flush-denormals-to-zero"]


# Kernels whose hand-issued loads and hand-counted s_waitcnt (asm volatile: invisible to the compiler's waitcnt pass) are only correct while
# the compiler adds NO vector memory operation of its own, i.e. no scratch (a spill store can retire out of order against the loads on
# gfx9): checked at build time from the compiler's own resource remarks -- a compiler upgrade or an edit that makes one of them spill
# fails the build instead of silently reading registers before their data has landed.  source -> substrings of kernel names
NO_SCRATCH = {"lut16.hip": ["lut16_apply_kernel"], "fixed_multi.hip": ["stream_multi2_kernel", "stream_multi_kernel"]}


def _check_no_scratch(src, remarks_file):
    import re

    wanted, cur, seen = NO_SCRATCH[src], None, set()
    for line in open(remarks_file, errors="replace"):
        m = re.search(r"remark:\s+(Function Name|ScratchSize \[bytes/lane\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2)
        elif cur is not None and any(w in cur for w in wanted):
            seen.add(cur)
            if int(m.group(2)) != 0:
                raise RuntimeError(f"{src}: kernel {cur} uses {m.group(2)} bytes/lane of scratch (lut16: its hand-counted s_waitcnt would be "
                                   f"wrong; the multi-tensor kernels: every wave of a launch pays for a spilling body) -- build.py NO_SCRATCH")
    if not seen:
        raise RuntimeError(f"{src}: no kernel matching {wanted} in the compiler's resource remarks: the NO_SCRATCH check did not run")


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
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s.split("#")[0]))]
    hipcc = _hipcc()

    def deps_of(dfile):
        """prerequisites hipcc recorded for the object (-MD -MF): the headers a source REALLY includes, no hand-kept table"""
        try:
            txt = open(dfile).read().replace("\\\n", " ")
        except OSError:
            return None
        names = txt.split(":", 1)[1].split() if ":" in txt else []
        return [n for n in names if not n.startswith(("/opt/", "/usr/"))]

    def compile_one(spec):
        src, _, part = spec.partition("#")
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", f"_p{part}.o" if part else ".o"))
        deps = deps_of(o + ".d")
        if force or deps is None or any(not os.path.exists(d) for d in deps) or _stale(o, [s, os.path.abspath(__file__)] + deps):
            cmd = [hipcc] + FLAGS + ([f"-DDMXQ_EW_PART={part}"] if part else []) + ["-MD", "-MF", o + ".d", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            if src in NO_SCRATCH:
                with open(o + ".remarks", "w") as rf:
                    rc = subprocess.call(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=rf)
                # stderr went to the remarks file: show everything in it that is NOT a resource remark (errors, warnings) on the console
                diag = [l for l in open(o + ".remarks", errors="replace") if "remark:" not in l and l.strip()]
                if diag:
                    sys.stderr.write("".join(diag))
                if rc != 0:
                    os.remove(o + ".remarks")
                    raise subprocess.CalledProcessError(rc, cmd)
            else:
                subprocess.check_call(cmd)
        if src in NO_SCRATCH:   # (every build, also when the object was up to date: the remarks of ITS compilation are kept beside it)
            if not os.path.exists(o + ".remarks"):
                os.remove(o)
                return compile_one(spec)
            _check_no_scratch(src, o + ".remarks")
        return o

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(srcs) + 1)) as ex:
        binding = ex.submit(build_torch_binding, force, verbose, False)  # g++ only: runs beside the hipcc jobs
        objs = list(ex.map(compile_one, srcs))
        binding_obj = binding.result()
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    link_torch_binding(binding_obj, force, verbose)
    return LIB


FENCES_LIB = os.path.join(HERE, "lib", "libdmxq_gate_fences.so")


def build_gate_fences_variant(verbose: bool = False) -> str:
    """lib/libdmxq_gate_fences.so: the product's objects with csrc/reduce.hip rebuilt -DDMXQ_GATE_FENCES=1 -- the FORMAL release / acquire
    form of the reductions' init gate (the default build orders with s_waitcnt + sc1 stores: a hardware argument, include/dmxq.h).
    Not used by the product; tests/test_gpu_round6.py runs the gate's stress tests against BOTH builds, loading this one
    through DMXQ_BINDING=ctypes DMXQ_LIB_PATH.  Built by __graft_entry__.build() so that it travels to the GPU box."""
    src = os.path.join(CSRC, "reduce.hip")
    base = os.path.join(OBJ, "reduce.o")
    obj = os.path.join(OBJ, "reduce_gate_fences.obj")   # (not *.o: build() links every *.o of this directory's product list only, but keep it apart anyway)
    deps = [src, os.path.join(CSRC, "gate_registry.hpp"), os.path.join(CSRC, "common.hpp"), os.path.abspath(__file__)]
    if _stale(obj, deps):
        cmd = [_hipcc()] + FLAGS + ["-DDMXQ_GATE_FENCES=1", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    objs = []
    for spec in SOURCES:
        s_, _, part = spec.partition("#")
        o = os.path.join(OBJ, s_.replace(".hip", f"_p{part}.o" if part else ".o"))
        objs.append(obj if o == base else o)
    if _stale(FENCES_LIB, objs):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-o", FENCES_LIB] + objs)
    return FENCES_LIB


def _torch_flags():
    import sysconfig

    import torch.utils.cpp_extension as cpp

    inc = []
    for p in cpp.include_paths():
        inc += ["-isystem", p]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc += ["-isystem", os.path.join(rocm, "include"), "-isystem", sysconfig.get_paths()["include"]]
    return inc, cpp.library_paths()[0]


def build_torch_binding(force: bool = False, verbose: bool = False, link: bool = True) -> str:
    """csrc/torch_binding.cpp -> build/torch_binding.o (host-only C++: g++ against the torch headers of this image;
    the TORCH_LIBRARY(dmxq) registration over the C ABI)."""
    src = os.path.join(CSRC, "torch_binding.cpp")
    obj = os.path.join(OBJ, "torch_binding.o")
    os.makedirs(OBJ, exist_ok=True)
    if force or _stale(obj, [src, os.path.join(HERE, "..", "include", "dmxq.h"), os.path.abspath(__file__)]):
        inc, _ = _torch_flags()
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               "-D_GLIBCXX_USE_CXX11_ABI=1", "-Wno-deprecated-declarations"] + inc + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if link:
        link_torch_binding(obj, force, verbose)
    return obj


def link_torch_binding(obj: str, force: bool = False, verbose: bool = False) -> str:
    if force or _stale(TORCH_LIB, [obj, LIB]):
        _, tlib = _torch_flags()
        cmd = [os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", TORCH_LIB, obj, "-L" + os.path.dirname(LIB), "-ldmxq",
               "-Wl,-rpath,$ORIGIN", "-L" + tlib, "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
               "-ltorch_hip", "-ltorch_python"]   # (torch_python: the pybind casters of the direct entry points, PyInit_dmxq_fast)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return TORCH_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
