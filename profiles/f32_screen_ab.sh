# float32 screening of the histogram bin off (DIRAL_F32_MARGIN=0) and on, kernel time of C2 and of the C4 shard
for r in 1 2 3; do for v in 1 0; do
  if [ $v = 1 ]; then export DIRAL_F32_MARGIN=0; else unset DIRAL_F32_MARGIN; fi
  for w in c2 c4shard; do st=400; [ $w = c4shard ] && st=150
  python bench.py --workload $w --lean --steps $st --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('noscreen=$v', '$w', round(d['roofline']['kernel_ms']*1e3,2))"; done
done; done
