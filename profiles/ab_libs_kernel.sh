#!/bin/bash
# GPU box: kernel time of one workload under several variant libraries, INTERLEAVED (boxes drift by +- 2 % within minutes:
# a variant is only better if it is better in every round).   bash profiles/ab_libs_kernel.sh <workload> <rounds> <lib> [lib ...]
# <lib>: a name under variants_tmp/ (lib_<name>.so) or "product" (diral_amd/libdiral_env.so)
W=$1; R=$2; shift 2
ST=100; [ $W = c2 ] && ST=600; [ $W = c4shard ] && ST=200; [ $W = c5 ] && ST=150
for r in $(seq 1 $R); do
  for L in "$@"; do
    LIB=$PWD/variants_tmp/lib_$L.so; [ $L = product ] && LIB=$PWD/diral_amd/libdiral_env.so
    DIRAL_LIB=$LIB timeout 300 python bench.py --workload $W --lean --steps $ST --warmup 20 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W round $r %-10s kernel %.4f ms  step %.4f ms' % ('$L', d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
