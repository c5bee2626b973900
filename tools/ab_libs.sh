#!/bin/bash
# tools/ab_libs.sh '<bench_ops --only filter>' product <variant.so> ... | python tools/ab_table.py
# A/B of library builds in ONE lease (round 5: VALU-heavy kernels swing 20-30 % between leases): the op table per library, interleaved
# twice.  A variant = the product's objects relinked with ONE recompiled object (e.g. -DDMXQ_EXP_STREAM_PACE=4), loaded through DMXQ_LIB_PATH
# (ctypes binding); `product` = dmx-compressor_amd/lib/libdmxq.so.
ONLY="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (rep $rep)"
    if [ "$lib" = "product" ]; then DMXQ_BINDING=ctypes python tools/bench_ops.py --only "$ONLY" 2>&1 | grep -v amdgpu.ids
    else DMXQ_BINDING=ctypes DMXQ_LIB_PATH=$lib python tools/bench_ops.py --only "$ONLY" 2>&1 | grep -v amdgpu.ids; fi
  done
done
