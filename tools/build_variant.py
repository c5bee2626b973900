#!/usr/bin/env python3
"""tools/build_variant.py <out.so> <source.hip[#part]> -DNAME=VALUE [...] — a variant of libdmxq.so for same-lease A/B runs: the product's
objects with ONE source recompiled under extra defines (e.g. -DDMXQ_EXP_PACK_PACE=2), linked into <out.so>.  Loaded through
DMXQ_BINDING=ctypes DMXQ_LIB_PATH=<out.so> (tools/ab_libs.sh)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dmx-compressor_amd"))
import build  # noqa: E402

out, spec, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
src, _, part = spec.partition("#")
base = src.replace(".hip", f"_p{part}.o" if part else ".o")
o = os.path.join(build.OBJ, "variant_" + str(os.getpid()) + "_" + base)
subprocess.check_call([build._hipcc()] + build.FLAGS + defs + ([f"-DDMXQ_EW_PART={part}"] if part else []) + ["-c", os.path.join(build.CSRC, src), "-o", o])
objs = [os.path.join(build.OBJ, f) for f in sorted(os.listdir(build.OBJ)) if f.endswith(".o") and f != base and not f.startswith("variant_") and f != "torch_binding.o"] + [o]
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-o", out] + objs)
os.remove(o)
print(out)
