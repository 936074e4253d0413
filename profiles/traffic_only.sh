#!/bin/bash
# bash profiles/traffic_only.sh <tag> [bench args...]  - FETCH_SIZE / WRITE_SIZE passes only (GPU box)
set -u
TAG=${1:-t}; shift || true
R=$PWD; OUT=$R/gpurun_out/traffic_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- \
    python $R/bench.py --steps 40 --warmup 0 --lean "$@" > $OUT/$c.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c and "step_" in r.get("Kernel_Name", ""):
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        m = sum(v) / len(v)
        print("%-12s %-60s n=%d mean=%.4g KB -> %.3f GB%s" % (c, k, len(v), m, m * (2 if c == "FETCH_SIZE" else 1) / 1e6,
                                                           " (x2 gfx950 correction)" if c == "FETCH_SIZE" else ""))
PY
