#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a summarize_profile.py summary.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 this rocprofv3 reports exactly half
of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section), so
the read side is doubled.  Calibration on a known byte count in THIS access
pattern: the kernel reads every table word once = B*N*64*(4+8) bytes
(201.3 MB at B=4096, N=64) and 2*FETCH_SIZE*1024 lands within 3 % of that plus
the per-vehicle arrays.
"""
import json
import re
import sys

summary, workload, out = sys.argv[1], sys.argv[2], sys.argv[3]
txt = open(summary).read()
f = float(re.search(r"FETCH_SIZE\s+mean=([0-9.e+]+)", txt).group(1))
w = float(re.search(r"WRITE_SIZE\s+mean=([0-9.e+]+)", txt).group(1))
try:
    d = json.load(open(out))
except Exception:
    d = {}
d[workload] = {
    "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
    "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
    "rule": "(2*FETCH_SIZE + WRITE_SIZE)*1024; gfx950 FETCH_SIZE reports half of a coalesced read",
    "source": summary,
}
json.dump(d, open(out, "w"), indent=1)
print(d[workload])
