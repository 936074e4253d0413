#!/bin/bash
# The round-2 profiling session (GPU box, through gpurun, from the repo root):
#   bash profiles/session_r02.sh
# kernel-trace + PMC passes for the five bench workloads / channel-observation settings, then the
# bench lines of the same box.  profiles/collect_r02.sh copies the results into profiles/r02/.
set -u
FULL_PMC=1 bash profiles/run_profile.sh c2_chobs1 > /dev/null 2>&1
FULL_PMC=1 bash profiles/run_profile.sh c2_chobs0 --emit-chobs 0 > /dev/null 2>&1
for w in c3 c5; do
  bash profiles/run_profile.sh ${w}_chobs1 --workload $w --steps 200 --warmup 20 > /dev/null 2>&1
  bash profiles/run_profile.sh ${w}_chobs0 --workload $w --emit-chobs 0 --steps 200 --warmup 20 > /dev/null 2>&1
done
bash profiles/run_profile.sh c4shard --workload c4shard --steps 200 --warmup 20 > /dev/null 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r02_full.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_r02_driverlike.json
for w in c3 c5; do
  python bench.py --workload $w --lean --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/bench_r02_$w.json
  python bench.py --workload $w --lean --steps 100 --warmup 10 --emit-chobs 0 2>/dev/null | tail -1 > gpurun_out/bench_r02_${w}_nochobs.json
done
python examples/rollout_sps.py --envs 4096 --slots 1000 2>&1 | grep -v amdgpu | tail -2 > gpurun_out/rollout_r02.txt
python examples/rollout_sps.py --envs 4096 --slots 1000 --policy random 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/rollout_r02.txt
for t in c2_chobs1 c2_chobs0 c3_chobs1 c3_chobs0 c5_chobs1 c5_chobs0 c4shard; do echo "== $t"; grep "steady state" gpurun_out/prof_$t/summary.txt; done
cat gpurun_out/rollout_r02.txt
