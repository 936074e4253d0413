#!/bin/bash
# A tuning variant of the library with only SOME translation units recompiled (the others come from the product's
# objects in diral_amd/build/): seconds instead of minutes.  Build container.
#   bash profiles/build_variant.sh <name> "<-D flags>" <tu> [tu ...]     ->  variants_tmp/lib_<name>.so
# e.g. bash profiles/build_variant.sh x1 "-DDIRAL_WIDE_BENCH_ONLY -DFOO=1" k_wide4
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; shift 2
mkdir -p variants_tmp/obj_$NAME
OBJS=""
for tu in diral_env k_fast64 k_wide2 k_wide4 k_general k_observe k_large; do
  if [[ " $* " == *" $tu "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC $FLAGS \
      -c diral_amd/csrc/$tu.hip -o variants_tmp/obj_$NAME/$tu.o &
    OBJS="$OBJS variants_tmp/obj_$NAME/$tu.o"
  else
    OBJS="$OBJS diral_amd/build/$tu.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o variants_tmp/lib_$NAME.so
ls -la variants_tmp/lib_$NAME.so
