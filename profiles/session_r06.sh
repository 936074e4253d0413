#!/bin/bash
# The round-6 profiling session (GPU box, through gpurun, from the repo root):
#   PMC_COMMIT=<git short hash> bash profiles/session_r06.sh
# kernel-trace + PMC passes (every counter group in its own run) for the bench workloads, then the bench lines of the same
# box, the launch timeline of the headline kernel with and without slow-first dispatch, the rollouts, the side paths.
# profiles/collect_r06.sh copies the results into profiles/r06/ and builds profiles/pmc_counters.json.
set -u
mkdir -p gpurun_out
export PMC_COMMIT=${PMC_COMMIT:-unknown}
FULL_PMC=1 bash profiles/run_profile.sh c2_chobs1 > /dev/null 2>&1
FULL_PMC=1 bash profiles/run_profile.sh c2_chobs0 --emit-chobs 0 > /dev/null 2>&1
for w in c3 c5; do
  FULL_PMC=1 bash profiles/run_profile.sh ${w}_chobs1 --workload $w --steps 200 --warmup 20 > /dev/null 2>&1
  bash profiles/run_profile.sh ${w}_chobs0 --workload $w --emit-chobs 0 --steps 200 --warmup 20 > /dev/null 2>&1
done
FULL_PMC=1 bash profiles/run_profile.sh c4shard --workload c4shard --steps 200 --warmup 20 > /dev/null 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r06_full.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_r06_driverlike.json
for w in c3 c5; do
  python bench.py --workload $w --lean --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/bench_r06_$w.json
  python bench.py --workload $w --lean --steps 100 --warmup 10 --emit-chobs 0 2>/dev/null | tail -1 > gpurun_out/bench_r06_${w}_nochobs.json
done
NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
  bench.py --gpus 1 --steps 200 --warmup 20 --lean > gpurun_out/bench_r06_torchrun1.stdout 2> gpurun_out/bench_r06_torchrun1.log
tail -1 gpurun_out/bench_r06_torchrun1.stdout > gpurun_out/bench_r06_torchrun1.json
grep -v '^{' gpurun_out/bench_r06_torchrun1.stdout >> gpurun_out/bench_r06_torchrun1.log
(bash profiles/scale.sh 1; bash profiles/scale.sh 1 2; echo "rc=$?") > gpurun_out/scale_r06.txt 2>&1
python profiles/rollout_lines.py 2>&1 | grep -v amdgpu > gpurun_out/rollout_r06.txt
python profiles/side_paths.py 2>&1 | grep -v amdgpu > gpurun_out/side_paths_r06.txt
WORKLOADS=c2,c5,c3 python profiles/secondary_modes.py 2>&1 | grep -v amdgpu > gpurun_out/secondary_modes_r06.txt
for v in 0 1; do echo "== DIRAL_NO_SLOW_FIRST=$v"; DIRAL_NO_SLOW_FIRST=$v bash profiles/batch_sweep.sh 64 256 1024 1792 2048 3584 4096 8192 32768 2>&1 | grep -v amdgpu; done > gpurun_out/batch_sweep_r06.txt
if [ -f variants_tmp/lib_timing.so ]; then
  for v in 0 1; do echo "== DIRAL_NO_SLOW_FIRST=$v"; DIRAL_NO_SLOW_FIRST=$v B=4096 DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/launch_timeline.py 2>&1 | grep -v amdgpu; done > gpurun_out/launch_timeline_r06.txt
  for B in 64 4096; do echo "=== B=$B"; B=$B DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/phase_timing.py 2>&1 | grep -v amdgpu | head -10; done > gpurun_out/phase_timing_r06.txt
fi
python profiles/kslots_bench.py 2>&1 | grep -v amdgpu > gpurun_out/kslots_r06.txt
python profiles/prefill_bench.py 2>&1 | grep -v amdgpu > gpurun_out/prefill_r06.txt
python profiles/large_path_bench.py 2>&1 | grep -v amdgpu > gpurun_out/large_path_r06.txt
# C5: the two table forms of 64 < N <= 128, the packed one with and without its slow envs dispatched first; how many passes leave the codes
(for F in plane packed; do for S in 0 1; do [ $F = plane ] && [ $S = 0 ] && continue; for i in 1 2; do
  DIRAL_NO_SLOW_FIRST=$S DIRAL_TABLE_FORM=$F python bench.py --workload c5 --lean --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 form=$F NO_SLOW_FIRST=$S: %.4f ms/step' % d['ms_per_step'], d['roofline'].get('kernel'))"
done; done; done; python profiles/flag_fraction.py 2>&1 | grep -v amdgpu) > gpurun_out/c5_forms_r06.txt
if [ -f variants_tmp/lib_timing.so ]; then
  for W in c3:8192 c5:16384; do w=${W%%:*}; B=${W##*:}; for b in 64 $B; do echo "=== $w B=$b"; SLOW_SPLIT=1 DIRAL_NO_SLOW_FIRST=1 WORKLOAD=$w B=$b DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/phase_timing.py 2>&1 | grep -v amdgpu | head -12; done; done > gpurun_out/phase_timing_wide_r06.txt
  DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/kslots_timing.py 2>&1 | grep -v amdgpu > gpurun_out/kslots_timing_r06.txt
fi
# the round's switches, interleaved on this box (variants_tmp/lib_*.so: profiles/ab_r06_build.sh in the build container)
if [ -f variants_tmp/lib_w4old.so ]; then
  (bash profiles/ab_libs_kernel.sh c3 3 w4old w4fma product; bash profiles/ab_libs_kernel.sh c5 3 w2old w2noguard product; bash profiles/ab_libs_kernel.sh c2 3 f64old product) > gpurun_out/ab_switches_r06.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_secondary -o t -- env WORKLOADS=c2,c5,c3 python $GRAFT_REPO_ROOT/profiles/secondary_modes.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rollout -o t -- python $GRAFT_REPO_ROOT/profiles/rollout_lines.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_large -o t -- python $GRAFT_REPO_ROOT/profiles/large_path_prof.py 1024 64 16000 256 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_large -name "*_kernel_trace.csv" | xargs rm -f
find gpurun_out/prof_secondary gpurun_out/prof_rollout -name "*_kernel_trace.csv" | xargs rm -f
for t in c2_chobs1 c2_chobs0 c3_chobs1 c3_chobs0 c5_chobs1 c5_chobs0 c4shard; do echo "== $t"; grep "steady state" gpurun_out/prof_$t/summary.txt; done
cat gpurun_out/rollout_r06.txt gpurun_out/scale_r06.txt
