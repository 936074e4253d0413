#!/bin/bash
# A/B helper (GPU box): wide-kernel parity subset, then C3 / C5 kernel times with and without the channel observation
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -x -q -p no:cacheprovider 2>&1 | tail -2
for w in c3 c5; do for e in 0 1; do
python bench.py --workload $w --emit-chobs $e --lean --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w chobs=$e', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
done; done
