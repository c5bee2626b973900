#!/usr/bin/env python3
"""tools/sanitize/build_host_asan.py — libdmxq's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU ASan and
XNACK are not available on this pool and are not attempted).  Every source of dmx-compressor_amd/build.py is compiled with
    hipcc --offload-arch=gfx950 -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan
(the device code is built uninstrumented and never runs here) into tools/sanitize/_out/libdmxq_asan.so.  What it is for: the argument
validation, the launch planning (rows_plan, lastdim_plan, slab / column tiling), the multi-tensor descriptor packers (fixed_multi.hip,
hypernet_multi.hip, bfp.hip's multi launcher) and the gate registry -- all host code that runs BEFORE a launch and that a GPU-less
process exercises completely: tools/sanitize/host_driver.py and tests/test_abi_and_host.py call it through ctypes with the ASan runtime
preloaded; a launch then fails with "no device", which the library reports as a status.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "dmx-compressor_amd"))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("_dmxq_build", os.path.join(ROOT, "dmx-compressor_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)

OUT = os.path.join(HERE, "_out")
FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-fPIC", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero",
         "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-shared-libsan", "-fno-omit-frame-pointer"]


def main():
    os.makedirs(OUT, exist_ok=True)
    hipcc = B._hipcc()

    def one(spec_):
        src, _, part = spec_.partition("#")
        s = os.path.join(B.CSRC, src)
        o = os.path.join(OUT, src.replace(".hip", f"_p{part}.o" if part else ".o"))
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(os.path.join(B.CSRC, f)) for f in os.listdir(B.CSRC)):
            subprocess.check_call([hipcc] + FLAGS + ([f"-DDMXQ_EW_PART={part}"] if part else []) + ["-c", s, "-o", o])
        return o

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8)) as ex:
        objs = list(ex.map(one, B.SOURCES))
    lib = os.path.join(OUT, "libdmxq_asan.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
