#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag>/ directory written by run_profile.sh:
per-kernel time stats and per-launch PMC averages for the step kernel.

PMC means are taken over the LAST `LAST` dispatches of each step kernel (env LAST,
default 40): the timed launches of the pass, after bench.py's untimed pre-roll."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
LAST = int(os.environ.get("LAST", "40"))
KF = os.environ.get("KERNEL_FILTER", "step_")        # substring of the kernel names summarised (default: the step kernels)


def find(pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")[:78]
            print("%-78s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
                name, row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))

# steady-state average of the step kernel from the raw trace: the last 1000 dispatches (the timed
# launches of the default command), which is what bench.py's HIP events bracket
for f in find("*kernel_trace.csv"):
    if os.sep + "trace" + os.sep not in f:
        continue
    durs = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if KF in row.get("Kernel_Name", ""):
                durs[row["Kernel_Name"]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    for k, v in durs.items():
        v.sort()
        tail = [x[1] for x in v[-1000:]]
        print("steady state: %-60s last %d dispatches avg_ns=%.0f (all %d: %.0f)" % (
            k[:60], len(tail), sum(tail) / len(tail), len(v), sum(x[1] for x in v) / len(v)))

print("\n== PMC per launch of the step kernel (mean over its last %d dispatches) ==" % LAST)
for f in find("*counter_collection.csv"):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if KF not in row.get("Kernel_Name", ""):
                continue
            acc[(row["Kernel_Name"][:48], row["Counter_Name"])].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
    for (kn, cn), v in sorted(acc.items()):
        v.sort()
        tail = [x[1] for x in v[-LAST:]]
        print("%-24s mean=%.6g  n=%d of %d  (%s; %s)" % (cn, sum(tail) / len(tail), len(tail), len(v),
                                                        os.path.basename(os.path.dirname(f)), kn))

# kernel duration INSIDE each PMC pass (every pass also carries --kernel-trace): the clock and busy
# fractions derived from a counter use the duration of the launches that counter was read on
print("\n== step-kernel duration inside each PMC pass (mean over the same last %d dispatches) ==" % LAST)
for f in find("*kernel_trace.csv"):
    if os.sep + "trace" + os.sep in f:
        continue
    durs = []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if KF in row.get("Kernel_Name", ""):
                durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    if durs:
        durs.sort()
        tail = [x[1] for x in durs[-LAST:]]
        print("PASS_NS %-12s mean=%.6g  n=%d of %d" % (os.path.basename(os.path.dirname(f)), sum(tail) / len(tail), len(tail), len(durs)))
