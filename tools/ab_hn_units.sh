#!/bin/bash
# tools/ab_hn_units.sh -- A/B of the fused weight chain's units per lane (csrc/hypernet_rows.hpp kHnUnits: 4 in the product) on the
# per-rank shard sets of a Llama-3-8B layer (tools/bench_shard_sets.py): a second libdmxq.so with DMXQ_HN_UNITS=${1:-8} is built from the
# product's objects plus the two re-compiled sources and selected with DMXQ_LIB_PATH (ctypes binding).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
B=dmx-compressor_amd/build
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero"
mkdir -p /tmp/hn8
hipcc $FLAGS -DDMXQ_HN_UNITS=${1:-8} -DDMXQ_HN_UNITS_SMALL=${2:-2} -c dmx-compressor_amd/csrc/hypernet_multi.hip -o /tmp/hn8/hypernet_multi.o &
hipcc $FLAGS -DDMXQ_HN_UNITS=${1:-8} -DDMXQ_HN_UNITS_SMALL=${2:-2} -c dmx-compressor_amd/csrc/hypernet.hip -o /tmp/hn8/hypernet.o &
wait
OBJS=$(ls $B/*.o | grep -v -e hypernet.o -e hypernet_multi.o -e torch_binding.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/hn8/libdmxq.so $OBJS /tmp/hn8/hypernet.o /tmp/hn8/hypernet_multi.o
echo "== kHnUnits = 4 above 160 M elements, 2 below (product)"
DMXQ_BINDING=ctypes python tools/bench_shard_sets.py
echo "== kHnUnits = ${1:-8}, kHnUnitsSmall = ${2:-2}"
DMXQ_BINDING=ctypes DMXQ_LIB_PATH=/tmp/hn8/libdmxq.so python tools/bench_shard_sets.py
