#!/bin/bash
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04d_pytest.txt
cat gpurun_out/r04d_pytest.txt
for v in 0 1; do
  echo "== DIRAL_NO_SLOW_FIRST=$v"
  DIRAL_NO_SLOW_FIRST=$v bash profiles/batch_sweep.sh 256 1792 4096 8192 32768 2>&1 | grep -v amdgpu
done | tee gpurun_out/r04d_sweep.txt
B=4096 DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/launch_timeline.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04d_timeline.txt
python bench.py --steps 20 --warmup 5 --lean 2>/dev/null | tail -1 > gpurun_out/r04d_bench_lean.json
python -c "
import json; d=json.load(open('gpurun_out/r04d_bench_lean.json')); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))"
