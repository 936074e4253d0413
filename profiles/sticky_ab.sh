#!/bin/bash
# kernel time at sticky action distributions (SURVEY 8d "converged" policy), this build vs the round-2 tree in variants_tmp/r02
for w in c2 c3 c5; do for s in 0.9 0.97; do
  st=300; [ $w = c3 ] && st=60; [ $w = c5 ] && st=100
  a=$(python bench.py --workload $w --sticky $s --lean --steps $st --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  b=$(cd variants_tmp/r02 && python bench.py --workload $w --sticky $s --lean --steps $st --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "$w sticky $s: now $a ms   round-2 $b ms"
done; done
