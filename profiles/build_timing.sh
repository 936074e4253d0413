#!/bin/bash
# Build container: variants_tmp/lib_timing.so - the library with -DDIRAL_TIMING (per-phase shader clocks: phase_timing.py,
# launch_timeline.py, kslots_timing.py), every instantiation, next to the shipped one.   bash profiles/build_timing.sh
set -e
cd "$(dirname "$0")/.."
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -DDIRAL_TIMING"
mkdir -p variants_tmp/obj_timing
for tu in diral_env k_fast64 k_wide2 k_wide4; do $CC -c diral_amd/csrc/$tu.hip -o variants_tmp/obj_timing/$tu.o & done
wait
$CC -shared variants_tmp/obj_timing/diral_env.o variants_tmp/obj_timing/k_fast64.o variants_tmp/obj_timing/k_wide2.o variants_tmp/obj_timing/k_wide4.o \
  diral_amd/build/k_general.o diral_amd/build/k_observe.o diral_amd/build/k_large.o -o variants_tmp/lib_timing.so
ls -la variants_tmp/lib_timing.so
