#!/bin/bash
# Build container: variants of the library with only k_wide2 recompiled (bench-only instantiations): lib_<name>.so, lib_<name>t.so (timing;
# needs variants_tmp/obj_smint/ of `profiles/ab_closure.sh smin` for the timing diral_env / k_wide4).   bash profiles/ab_wide2.sh <name> ["-D..."]
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS="-DDIRAL_WIDE_BENCH_ONLY $2"
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC"
mkdir -p variants_tmp/obj_$NAME variants_tmp/obj_${NAME}t
$CC $FLAGS -c diral_amd/csrc/k_wide2.hip -o variants_tmp/obj_$NAME/k_wide2.o &
$CC $FLAGS -DDIRAL_TIMING -c diral_amd/csrc/k_wide2.hip -o variants_tmp/obj_${NAME}t/k_wide2.o &
wait
O="diral_amd/build/k_fast64.o diral_amd/build/k_general.o diral_amd/build/k_observe.o"
$CC -shared diral_amd/build/diral_env.o diral_amd/build/k_wide4.o variants_tmp/obj_$NAME/k_wide2.o $O -o variants_tmp/lib_$NAME.so
$CC -shared variants_tmp/obj_smint/diral_env.o variants_tmp/obj_smint/k_wide4.o variants_tmp/obj_${NAME}t/k_wide2.o $O -o variants_tmp/lib_${NAME}t.so
ls -la variants_tmp/lib_$NAME.so
