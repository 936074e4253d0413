#!/bin/bash
# Build container: variant libraries lib_<name>.so / lib_<name>t.so (-DDIRAL_TIMING) with diral_env, k_wide2 and k_wide4 recompiled
# (bench-only instantiations of the wide kernels); the other objects from diral_amd/build.   bash profiles/ab_p2.sh <name> ["-D..."]
set -e
cd "$(dirname "$0")/.."
NAME=${1:-p2}; FLAGS="$2"
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC"
mkdir -p variants_tmp/obj_$NAME variants_tmp/obj_${NAME}t
for tu in k_wide2 k_wide4; do
  $CC $FLAGS -DDIRAL_WIDE_BENCH_ONLY -c diral_amd/csrc/$tu.hip -o variants_tmp/obj_$NAME/$tu.o &
  $CC $FLAGS -DDIRAL_WIDE_BENCH_ONLY -DDIRAL_TIMING -c diral_amd/csrc/$tu.hip -o variants_tmp/obj_${NAME}t/$tu.o &
done
$CC $FLAGS -c diral_amd/csrc/diral_env.hip -o variants_tmp/obj_$NAME/diral_env.o &
$CC $FLAGS -DDIRAL_TIMING -c diral_amd/csrc/diral_env.hip -o variants_tmp/obj_${NAME}t/diral_env.o &
[ -f variants_tmp/obj_timing/k_fast64.o ] || $CC -DDIRAL_TIMING -DDIRAL_FAST_BENCH_ONLY -c diral_amd/csrc/k_fast64.hip -o variants_tmp/obj_timing/k_fast64.o &
wait
O="diral_amd/build/k_general.o diral_amd/build/k_observe.o"
$CC -shared variants_tmp/obj_$NAME/diral_env.o variants_tmp/obj_$NAME/k_wide2.o variants_tmp/obj_$NAME/k_wide4.o diral_amd/build/k_fast64.o $O -o variants_tmp/lib_$NAME.so
$CC -shared variants_tmp/obj_${NAME}t/diral_env.o variants_tmp/obj_${NAME}t/k_wide2.o variants_tmp/obj_${NAME}t/k_wide4.o variants_tmp/obj_timing/k_fast64.o $O -o variants_tmp/lib_${NAME}t.so
ls -la variants_tmp/lib_$NAME.so variants_tmp/lib_${NAME}t.so
