#!/bin/bash
# WORKLOAD=c2 bash profiles/bench_variants.sh [variant.so names...]   (GPU box; prints value / kernel_ms / frac per variant)
R=$PWD
WL=${WORKLOAD:-c2}
for v in default "$@"; do
  L=$R/variants_tmp/lib_$v.so; [ "$v" = default ] && L=$R/diral_amd/libdiral_env.so
  printf "== %-14s " $v
  DIRAL_LIB=$L python $R/bench.py --workload $WL --steps ${STEPS:-300} --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g agent-steps/s  %.2f us  frac %.3f' % (d['value'], d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))"
done
