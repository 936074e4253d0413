#!/bin/bash
# profiles/ab_driverlike.sh <rounds> libs... : the driver's command (--steps 20 --warmup 5, lean), wall-clock ms_per_step and the event time per step
R=$1; shift
for r in $(seq $R); do
for l in "$@"; do
  if [ "$l" = base ]; then unset DIRAL_LIB; else export DIRAL_LIB=$PWD/variants_tmp/lib_$l.so; fi
  python bench.py --lean --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l  step %.2f us  events %.2f us  %.3f G' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value']/1e9))"
done; done
