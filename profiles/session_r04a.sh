#!/bin/bash
# round 4, first look: baseline of HEAD on this box + per-phase clocks of the C2 kernel at several batch sizes
set -u
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --lean 2>/dev/null | tail -1 > gpurun_out/r04a_bench_lean.json
for B in 64 256 1792 4096; do
  echo "=== B=$B"; B=$B DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/phase_timing.py 2>&1 | grep -v amdgpu
done > gpurun_out/r04a_phase_timing.txt
bash profiles/batch_sweep.sh 64 128 256 512 1024 1792 4096 2>&1 | grep -v amdgpu > gpurun_out/r04a_batch_sweep.txt
cat gpurun_out/r04a_phase_timing.txt gpurun_out/r04a_batch_sweep.txt
python -c "
import json; d=json.load(open('gpurun_out/r04a_bench_lean.json')); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))"
