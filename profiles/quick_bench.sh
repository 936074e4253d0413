#!/bin/bash
# kernel time (HIP events inside bench.py) of the four bench workloads, lean runs; GPU box.
#   bash profiles/quick_bench.sh [tag]
T=${1:-q}
for w in c2 c4shard c3 c5; do
  st=400; [ $w = c3 ] && st=100; [ $w = c5 ] && st=150; [ $w = c4shard ] && st=200
  python bench.py --workload $w --lean --steps $st --warmup 20 2>/dev/null | tail -1 > gpurun_out/${T}_$w.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_$w.json')); r=d['roofline']
print('%-8s kernel %.4f ms  step %.4f ms  %.4g agent-steps/s  frac %.3f' % ('$w', r['kernel_ms'], d['ms_per_step'], d['value'], r['frac']))"
done
