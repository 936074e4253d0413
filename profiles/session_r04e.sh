#!/bin/bash
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04e_pytest.txt
cat gpurun_out/r04e_pytest.txt
bash profiles/batch_sweep.sh 64 256 1792 4096 8192 32768 2>&1 | grep -v amdgpu | tee gpurun_out/r04e_sweep.txt
for B in 64 4096; do echo "=== B=$B"; B=$B DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/phase_timing.py 2>&1 | grep -v amdgpu | head -10; done | tee gpurun_out/r04e_phase.txt
B=4096 DIRAL_LIB=$PWD/variants_tmp/lib_timing.so python profiles/launch_timeline.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04e_timeline.txt
bash profiles/quick_bench.sh r04e 2>&1 | grep -v amdgpu
