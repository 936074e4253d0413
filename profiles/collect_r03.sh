#!/bin/bash
# Copy the summaries of the round-3 profiling session (profiles/session_r03.sh) from gpurun_out/ into profiles/r03/,
# write the resource-usage table of the build and profiles/pmc_counters.json.  Run in the build container.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r03
for t in c2_chobs1 c2_chobs0 c3_chobs1 c3_chobs0 c5_chobs1 c5_chobs0 c4shard; do
  cp gpurun_out/prof_$t/summary.txt profiles/r03/${t}_summary.txt
  find gpurun_out/prof_$t/trace -name "*kernel_stats.csv" -exec cp {} profiles/r03/${t}_kernel_stats.csv \;
done
for f in full driverlike c3 c3_nochobs c5 c5_nochobs torchrun1; do cp gpurun_out/bench_r03_$f.json profiles/r03/bench_$f.json; done
grep -v "amdgpu.ids" gpurun_out/bench_r03_torchrun1.log | cut -c1-400 > profiles/r03/bench_torchrun1_rccl.log
cp gpurun_out/bench_r03_gpus2.txt profiles/r03/bench_gpus2_on_one_gpu.txt
cp gpurun_out/rollout_r03.txt profiles/r03/rollout_example.txt
cp gpurun_out/side_paths_r03.txt profiles/r03/side_paths.txt
cp gpurun_out/secondary_modes_r03.txt profiles/r03/secondary_modes.txt
cp gpurun_out/two_streams_r03.txt profiles/r03/two_streams.txt
for f in batch_sweep end_effects lag_distribution any_order; do cp gpurun_out/${f}_r03.txt profiles/r03/$f.txt; done
find gpurun_out/prof_secondary -name "*kernel_stats.csv" -exec cp {} profiles/r03/secondary_modes_kernel_stats.csv \;
find gpurun_out/prof_side -name "*kernel_stats.csv" -exec cp {} profiles/r03/side_paths_kernel_stats.csv \;
bash profiles/resource_usage.sh profiles/r03/resource_usage.txt
python3 profiles/make_pmc_json.py profiles/r03 c2=c2_chobs1 c2_nochobs=c2_chobs0 c3=c3_chobs1 c3_nochobs=c3_chobs0 c5=c5_chobs1 c5_nochobs=c5_chobs0 c4shard=c4shard
