#!/bin/bash
# kernel time of the c2 workload against the batch size (is the cost per launch a + b B ?): bash profiles/batch_sweep.sh [sizes...]
S="$@"; [ -z "$S" ] && S="1024 2048 4096 8192 16384 32768"
for B in $S; do
  python bench.py --workload c2 --batch $B --lean --steps 300 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B $B kernel %.2f us  step %.2f us  per env %.2f ns' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e6/$B))"
done
