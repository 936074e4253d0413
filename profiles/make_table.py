#!/usr/bin/env python3
"""Markdown table of profiles/pmc_counters.json (the per-launch table of profiles/README.md): python3 profiles/make_table.py"""
import json
import os

d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_counters.json")))
cols = [("c2", "C2, channel obs"), ("c2_nochobs", "C2 without"), ("c4shard", "C4 shard (B = 32 768)"), ("c3", "C3, channel obs"),
        ("c3_nochobs", "C3 without"), ("c5", "C5, channel obs"), ("c5_nochobs", "C5 without")]


def t(ms):
    return "%.1f µs" % (ms * 1e3) if ms < 0.2 else "%.3f ms" % ms


def gb(b):
    return "%.0f MB" % (b / 1e6) if b < 1e9 else "%.2f GB" % (b / 1e9)


def traffic(v):
    f, w = 2 * v["FETCH_SIZE_KiB"] * 1024, v["WRITE_SIZE_KiB"] * 1024
    if f + w < 1e9:
        return "%.0f + %.0f = %.0f MB" % (f / 1e6, w / 1e6, (f + w) / 1e6)
    return "%.2f + %.2f = %.2f GB" % (f / 1e9, w / 1e9, (f + w) / 1e9)


def opt(x, fmt="%.2f"):
    return fmt % x if x is not None else "–"


rows = [
    ("kernel (rocprofv3 steady-state avg)", lambda v: t(v["kernel_ms_profiled"])),
    ("HBM traffic: 2·FETCH + WRITE", traffic),
    ("HBM rate in the traffic passes (÷ 8 TB/s)", lambda v: "%.2f TB/s (%.2f)" % (v["hbm_bytes_per_launch"] / (v["kernel_ms_traffic_passes"] * 1e-3) / 1e12,
                                                                                 v["hbm_bytes_per_launch"] / (v["kernel_ms_traffic_passes"] * 1e-3) / 8e12)),
    ("VALU / SALU / LDS instructions per wave", lambda v: "%.0f / %.0f / %.0f" % (v["per_wave"]["valu"], v["per_wave"]["salu"], v["per_wave"]["lds"])),
    ("clock: GRBM_GUI_ACTIVE ÷ 8 ÷ duration (GHz)", lambda v: opt(v.get("grbm_clock_GHz"))),
    ("VALU busy: 4·SQ_ACTIVE_INST_VALU ÷ (1024 · clock · duration)", lambda v: opt(v.get("valu_busy"))),
    ("LDS busy: SQ_LDS_IDX_ACTIVE ÷ (256 · clock · duration)", lambda v: opt(v.get("lds_busy"))),
    ("CUs busy: SQ_BUSY_CU_CYCLES ÷ (256 · clock · duration)", lambda v: opt(v.get("cu_busy"))),
    ("VALU busy while the CU is busy", lambda v: opt(v.get("valu_busy_while_cu_busy"))),
]
print("| | " + " | ".join(n for _, n in cols) + " |")
print("|---|" + "---|" * len(cols))
for name, fn in rows:
    print("| " + name + " | " + " | ".join(fn(d[k]) for k, _ in cols) + " |")
