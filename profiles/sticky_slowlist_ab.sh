for r in 1 2; do for l in base slow4 slow2; do
  if [ "$l" = base ]; then unset DIRAL_LIB; else export DIRAL_LIB=$PWD/variants_tmp/lib_$l.so; fi
  for st in 0 0.9; do python bench.py --workload c2 --lean --steps 300 --warmup 20 --sticky $st 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l sticky $st', round(d['roofline']['kernel_ms']*1e3,2))"; done
done; done
