#!/bin/bash
# round 4: parity + quick numbers of the working tree
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04b_pytest.txt
cat gpurun_out/r04b_pytest.txt
bash profiles/quick_bench.sh r04b 2>&1 | grep -v amdgpu
bash profiles/batch_sweep.sh 64 256 1792 4096 32768 2>&1 | grep -v amdgpu | tee gpurun_out/r04b_batch_sweep.txt
