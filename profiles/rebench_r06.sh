#!/bin/bash
# GPU box: the bench lines of profiles/r06 again, after profiles/collect_r06.sh has written profiles/pmc_counters.json for the
# kernel sources of the session (the lines taken inside the session carry the counters of the previous one)
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r06_full.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_r06_driverlike.json
for w in c3 c5; do
  python bench.py --workload $w --lean --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/bench_r06_$w.json
  python bench.py --workload $w --lean --steps 100 --warmup 10 --emit-chobs 0 2>/dev/null | tail -1 > gpurun_out/bench_r06_${w}_nochobs.json
done
for i in 1 2 3; do for H in "" 0; do
  if [ -z "$H" ]; then unset HSA_ENABLE_INTERRUPT; else export HSA_ENABLE_INTERRUPT=$H; fi
  python bench.py --steps 20 --warmup 5 --lean 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HSA_ENABLE_INTERRUPT=$H: %.3f us/step, kernel %.3f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done; done
