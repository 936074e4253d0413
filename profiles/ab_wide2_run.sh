#!/bin/bash
# GPU box: C5 parity of the benchmarked instantiation, phase timing, bench line of variant libraries ("product" = the shipped library)
for L in "$@"; do
  echo "=== $L"
  if [ "$L" = product ]; then unset DIRAL_LIB; else export DIRAL_LIB=$PWD/variants_tmp/lib_$L.so; fi
  timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c5_bench" 2>&1 | tail -1
  if [ "$L" != product ]; then for B in 64 16384; do DIRAL_LIB=$PWD/variants_tmp/lib_${L}t.so WORKLOAD=c5 B=$B timeout 200 python profiles/phase_timing.py 2>&1 | grep -v amdgpu.ids | head -10; done; fi
  timeout 200 python bench.py --workload c5 --lean --steps 100 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench c5: %.4f ms/step' % d['ms_per_step'])"
done
