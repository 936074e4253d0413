#!/bin/bash
# The round-3 profiling session (GPU box, through gpurun, from the repo root):
#   bash profiles/session_r03.sh
# kernel-trace + PMC passes (every counter group in its own run) for the bench workloads, then the bench lines of
# the same box, the bench under torch.distributed.run with one rank (RCCL group creation), the side paths.
# profiles/collect_r03.sh copies the results into profiles/r03/ and builds profiles/pmc_counters.json.
set -u
mkdir -p gpurun_out
FULL_PMC=1 bash profiles/run_profile.sh c2_chobs1 > /dev/null 2>&1
FULL_PMC=1 bash profiles/run_profile.sh c2_chobs0 --emit-chobs 0 > /dev/null 2>&1
for w in c3 c5; do
  FULL_PMC=1 bash profiles/run_profile.sh ${w}_chobs1 --workload $w --steps 200 --warmup 20 > /dev/null 2>&1
  bash profiles/run_profile.sh ${w}_chobs0 --workload $w --emit-chobs 0 --steps 200 --warmup 20 > /dev/null 2>&1
done
FULL_PMC=1 bash profiles/run_profile.sh c4shard --workload c4shard --steps 200 --warmup 20 > /dev/null 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r03_full.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_r03_driverlike.json
for w in c3 c5; do
  python bench.py --workload $w --lean --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/bench_r03_$w.json
  python bench.py --workload $w --lean --steps 100 --warmup 10 --emit-chobs 0 2>/dev/null | tail -1 > gpurun_out/bench_r03_${w}_nochobs.json
done
NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
  bench.py --gpus 1 --steps 200 --warmup 20 --lean > gpurun_out/bench_r03_torchrun1.stdout 2> gpurun_out/bench_r03_torchrun1.log
tail -1 gpurun_out/bench_r03_torchrun1.stdout > gpurun_out/bench_r03_torchrun1.json        # (NCCL_DEBUG=INFO prints to stdout)
grep -v '^{' gpurun_out/bench_r03_torchrun1.stdout >> gpurun_out/bench_r03_torchrun1.log
python bench.py --gpus 2 > gpurun_out/bench_r03_gpus2.txt 2>&1; echo "rc=$?" >> gpurun_out/bench_r03_gpus2.txt
python examples/rollout_sps.py --envs 4096 --slots 1000 2>&1 | grep -v amdgpu | tail -2 > gpurun_out/rollout_r03.txt
python examples/rollout_sps.py --envs 4096 --slots 1000 --policy random 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/rollout_r03.txt
python profiles/side_paths.py 2>&1 | grep -v amdgpu > gpurun_out/side_paths_r03.txt
WORKLOADS=c2,c5,c3 python profiles/secondary_modes.py 2>&1 | grep -v amdgpu > gpurun_out/secondary_modes_r03.txt
python profiles/two_streams.py 2>&1 | grep -v amdgpu > gpurun_out/two_streams_r03.txt
bash profiles/batch_sweep.sh 64 256 1024 1792 2048 3584 4096 8192 32768 2>&1 | grep -v amdgpu > gpurun_out/batch_sweep_r03.txt
python profiles/end_effects.py 2>&1 | grep -v amdgpu > gpurun_out/end_effects_r03.txt
python profiles/lag_distribution.py 2>&1 | grep -v amdgpu > gpurun_out/lag_distribution_r03.txt
(hipcc --offload-arch=gfx950 -O3 profiles/micro/any_order.hip -o /tmp/any_order 2>/dev/null && timeout 60 /tmp/any_order) > gpurun_out/any_order_r03.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_secondary -o t -- env WORKLOADS=c2,c5,c3 python $GRAFT_REPO_ROOT/profiles/secondary_modes.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_side -o t -- python $GRAFT_REPO_ROOT/profiles/side_paths.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_secondary gpurun_out/prof_side -name "*_kernel_trace.csv" | xargs rm -f
for t in c2_chobs1 c2_chobs0 c3_chobs1 c3_chobs0 c5_chobs1 c5_chobs0 c4shard; do echo "== $t"; grep "steady state" gpurun_out/prof_$t/summary.txt; done
cat gpurun_out/rollout_r03.txt gpurun_out/two_streams_r03.txt; tail -3 gpurun_out/bench_r03_torchrun1.log | cut -c1-200
