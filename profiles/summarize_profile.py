#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag>/ directory written by run_profile.sh:
per-kernel time stats and per-launch PMC averages for the step kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")[:70]
            print("%-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
                name, row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))

print("\n== PMC per launch of step_kernel (mean over dispatches) ==")
for f in find("*counter_collection.csv"):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "step_" not in row.get("Kernel_Name", ""):
                continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-24s mean=%.6g  n=%d  (%s)" % (k, sum(v) / len(v), len(v), os.path.basename(os.path.dirname(f))))
