#!/bin/bash
# Build container: variants of the library with only k_wide4 (+ diral_env for -DDIRAL_TIMING) recompiled, bench-only
# instantiations (seconds): variants_tmp/lib_<name>.so and lib_<name>t.so (timing).   bash profiles/ab_closure.sh <name> ["-D..."]
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS="-DDIRAL_WIDE_BENCH_ONLY $2"
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC"
mkdir -p variants_tmp/obj_$NAME variants_tmp/obj_${NAME}t
$CC $FLAGS -c diral_amd/csrc/k_wide4.hip -o variants_tmp/obj_$NAME/k_wide4.o &
$CC $FLAGS -DDIRAL_TIMING -c diral_amd/csrc/k_wide4.hip -o variants_tmp/obj_${NAME}t/k_wide4.o &
$CC $FLAGS -DDIRAL_TIMING -c diral_amd/csrc/diral_env.hip -o variants_tmp/obj_${NAME}t/diral_env.o &
wait
O="diral_amd/build/k_fast64.o diral_amd/build/k_wide2.o diral_amd/build/k_general.o diral_amd/build/k_observe.o"
$CC -shared diral_amd/build/diral_env.o variants_tmp/obj_$NAME/k_wide4.o $O -o variants_tmp/lib_$NAME.so
$CC -shared variants_tmp/obj_${NAME}t/diral_env.o variants_tmp/obj_${NAME}t/k_wide4.o $O -o variants_tmp/lib_${NAME}t.so
ls -la variants_tmp/lib_$NAME.so variants_tmp/lib_${NAME}t.so
