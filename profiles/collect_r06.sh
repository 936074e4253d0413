#!/bin/bash
# Copy the summaries of the round-6 profiling session (profiles/session_r06.sh) from gpurun_out/ into profiles/r06/,
# write the resource-usage table of the build and profiles/pmc_counters.json.  Run in the build container.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r06
for t in c2_chobs1 c2_chobs0 c3_chobs1 c3_chobs0 c5_chobs1 c5_chobs0 c4shard; do
  cp gpurun_out/prof_$t/summary.txt profiles/r06/${t}_summary.txt
  find gpurun_out/prof_$t/trace -name "*kernel_stats.csv" -exec cp {} profiles/r06/${t}_kernel_stats.csv \;
done
for f in full driverlike c3 c3_nochobs c5 c5_nochobs torchrun1; do cp gpurun_out/bench_r06_$f.json profiles/r06/bench_$f.json; done
grep -v "amdgpu.ids" gpurun_out/bench_r06_torchrun1.log | cut -c1-400 > profiles/r06/bench_torchrun1_rccl.log
for f in scale rollout side_paths secondary_modes batch_sweep launch_timeline phase_timing phase_timing_wide kslots kslots_timing c5_forms ab_switches prefill large_path; do [ -f gpurun_out/${f}_r06.txt ] && cp gpurun_out/${f}_r06.txt profiles/r06/$f.txt; done
find gpurun_out/prof_secondary -name "*kernel_stats.csv" -exec cp {} profiles/r06/secondary_modes_kernel_stats.csv \;
find gpurun_out/prof_rollout -name "*kernel_stats.csv" -exec cp {} profiles/r06/rollout_kernel_stats.csv \;
find gpurun_out/prof_large -name "*kernel_stats.csv" -exec cp {} profiles/r06/large_path_kernel_stats.csv \;
bash profiles/resource_usage.sh profiles/r06/resource_usage.txt
python3 profiles/make_pmc_json.py profiles/r06 c2=c2_chobs1 c2_nochobs=c2_chobs0 c3=c3_chobs1 c3_nochobs=c3_chobs0 c5=c5_chobs1 c5_nochobs=c5_chobs0 c4shard=c4shard
