#!/bin/bash
W=$1; E=$2; shift 2
for l in "$@"; do
  if [ "$l" = base ]; then unset DIRAL_LIB; else export DIRAL_LIB=$PWD/variants_tmp/lib_$l.so; fi
  python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "wide or 256" 2>&1 | tail -1
  for r in 1 2; do
  python bench.py --workload $W --emit-chobs $E --lean --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', d['config']['workload'][:14], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done
