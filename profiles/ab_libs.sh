#!/bin/bash
# profiles/ab_libs.sh <workload> <emit> <rounds> libs... : interleaved rounds, 400 timed steps each; prints every kernel time and the minimum per lib
W=$1; E=$2; R=$3; shift 3
for r in $(seq $R); do
for l in "$@"; do
  if [ "$l" = base ]; then unset DIRAL_LIB; else export DIRAL_LIB=$PWD/variants_tmp/lib_$l.so; fi
  python bench.py --workload $W --emit-chobs $E --lean --steps 400 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', '$W', round(d['roofline']['kernel_ms'],4))"
done; done | tee /tmp/ab_$W.txt | python -c "
import sys, collections
d = collections.defaultdict(list)
for ln in sys.stdin:
    l, w, t = ln.split(); d[l].append(float(t))
for l, v in d.items(): print(l, sys.argv[1] if len(sys.argv) > 1 else '', 'min %.4f med %.4f  all %s' % (min(v), sorted(v)[len(v)//2], ' '.join('%.3f' % x for x in v)))
" $W
