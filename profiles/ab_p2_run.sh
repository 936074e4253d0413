#!/bin/bash
# GPU box: the packed table form at 64 < N <= 128 (DIRAL_TABLE_FORM=packed) against the plane form: parity, then the C5 bench line of both
export DIRAL_LIB=$PWD/variants_tmp/lib_${1:-p2}.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "packed_tables_and_xpos" 2>&1 | tail -3
for F in plane packed; do
  echo "=== form $F"
  DIRAL_TABLE_FORM=$F timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c5" 2>&1 | tail -1
  for i in 1 2; do DIRAL_TABLE_FORM=$F timeout 200 python bench.py --workload c5 --lean --steps 100 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench c5: %.4f ms/step' % d['ms_per_step'], d['roofline'].get('kernel'))"; done
done
