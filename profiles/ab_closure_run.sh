#!/bin/bash
# GPU box: parity of the C3 benchmarked instantiation, phase timing and the bench line of variant libraries.
#   bash profiles/ab_closure_run.sh <name> [name ...]
for L in "$@"; do
  echo "=== $L"
  DIRAL_LIB=$PWD/variants_tmp/lib_$L.so timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c3_bench" 2>&1 | tail -2
  for B in 64 8192; do DIRAL_LIB=$PWD/variants_tmp/lib_${L}t.so WORKLOAD=c3 B=$B timeout 200 python profiles/phase_timing.py 2>&1 | grep -v amdgpu.ids | head -10; done
  DIRAL_LIB=$PWD/variants_tmp/lib_$L.so timeout 200 python bench.py --workload c3 --lean --steps 100 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench c3: %.4f ms/step, kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
