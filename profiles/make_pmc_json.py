#!/usr/bin/env python3
"""profiles/pmc_counters.json from the rocprofv3 summaries of a profiling session (profiles/run_profile.sh):
what bench.py quotes as `roofline.traffic`, `counter_frac`, `valu_busy`, `lds_busy`, `cu_busy`, `clock_GHz`.

  python profiles/make_pmc_json.py profiles/r03 c2=c2_chobs1 c2_nochobs=c2_chobs0 c3=c3_chobs1 ...

Every counter is the mean over the last 40 launches of its own --pmc pass (separate runs, never combined with
sys / hip traces); every rate uses the kernel duration measured INSIDE that pass (PASS_NS lines of the summary).
  hbm bytes   = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        gfx950: FETCH_SIZE reports half of a coalesced read
  clock       = GRBM_GUI_ACTIVE / 8 XCDs / duration           (the guide's effective clock; on a kernel of < 100 us it
                                                               includes the dispatch gaps: capped at the 2.4 GHz maximum)
  valu_busy   = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * clock * duration)     (SQ_ACTIVE_* count quad-cycles)
  lds_busy    = SQ_LDS_IDX_ACTIVE / (256 CUs * clock * duration)
  cu_busy     = SQ_BUSY_CU_CYCLES / (256 CUs * clock * duration)              (FULL_PMC sessions only)
"""
import json
import re
import subprocess
import sys

MAX_CLOCK_GHZ = 2.4


def traffic_factors():
    """bytes really moved / (counter x 1024) for the access forms of the step kernels, measured on known byte counts by
    profiles/micro/traffic_calib.hip (profiles/calibrate_traffic.sh -> profiles/traffic_calibration.json): FETCH_SIZE
    x 2.000 for one dword, 8 B and 16 B per lane alike and for the table's read-modify-write row pattern; WRITE_SIZE
    x 1.000 (non-temporal 16 B stores: 0.994).  Falls back to the guide's (2, 1) when the file is missing."""
    import os
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic_calibration.json")) as fh:
            forms = json.load(fh)["forms"]
        f = forms["rw_dword_rows"]["FETCH_SIZE_factor"]
        w = forms["rw_dword_rows"]["WRITE_SIZE_factor"]
        return f, w, "profiles/traffic_calibration.json (rw_dword_rows: the table's access form; dword / 8 B / 16 B reads all x %.3f)" % f
    except Exception:
        return 2.0, 1.0, "MI355X_MICROARCH.md (FETCH_SIZE reports half of a coalesced read); no calibration file"


def parse(path):
    txt = open(path).read()
    g = {}
    for m in re.finditer(r"^(\w+)\s+mean=([0-9.e+-]+)\s+n=\d+ of \d+\s+\((\w+);", txt, re.M):
        g[m.group(1)] = (float(m.group(2)), m.group(3))
    dur = {m.group(1): float(m.group(2)) for m in re.finditer(r"^PASS_NS (\w+)\s+mean=([0-9.e+-]+)", txt, re.M)}
    m = re.search(r"steady state:.*avg_ns=([0-9.]+) ", txt)
    return g, dur, (float(m.group(1)) if m else None)


def record(summary, head):
    g, dur, steady = parse(summary)
    def val(k):
        return g[k][0] if k in g else None
    def ns(k):
        return dur.get(g[k][1]) if k in g else None
    stamp = re.search(r"^CSRC_SHA (\w+)\s+COMMIT (\S+)", open(summary).read(), re.M)
    csrc_sha, commit = (stamp.group(1), stamp.group(2)) if stamp else (None, head)
    out = {"source": "%s (rocprofv3 --pmc passes of `python bench.py --lean ...`, separate runs, mean over the last 40 launches "
                     "after bench.py's pre-roll; kernel sources %s, commit %s)" % (summary, csrc_sha, commit),
           "csrc_sha": csrc_sha, "commit": commit,
           "kernel_ms_profiled": steady / 1e6 if steady else None}
    f, w = val("FETCH_SIZE"), val("WRITE_SIZE")
    if f is not None and w is not None:
        ff, wf, src = traffic_factors()
        out.update(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, hbm_bytes_per_launch=(ff * f + wf * w) * 1024.0,
                   hbm_read_bytes_per_launch=ff * f * 1024.0, hbm_write_bytes_per_launch=wf * w * 1024.0,
                   hbm_rule="(%.4f*FETCH_SIZE + %.4f*WRITE_SIZE)*1024; factors from %s" % (ff, wf, src))
        tns = [x for x in (ns("FETCH_SIZE"), ns("WRITE_SIZE")) if x]
        if tns:
            out["kernel_ms_traffic_passes"] = sum(tns) / len(tns) / 1e6
    clock = None
    if val("GRBM_GUI_ACTIVE") and ns("GRBM_GUI_ACTIVE"):
        raw = val("GRBM_GUI_ACTIVE") / 8.0 / ns("GRBM_GUI_ACTIVE")
        clock = min(raw, MAX_CLOCK_GHZ)
        out.update(grbm_clock_GHz=raw, clock_GHz=clock,
                   clock_note="GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of that pass" + (
                       "; above the 2.4 GHz maximum because GUI_ACTIVE spans the dispatch gaps of a short kernel: capped" if raw > MAX_CLOCK_GHZ else ""))
    if clock:
        if val("SQ_ACTIVE_INST_VALU") and ns("SQ_ACTIVE_INST_VALU"):
            out["valu_busy"] = 4.0 * val("SQ_ACTIVE_INST_VALU") / (1024 * clock * ns("SQ_ACTIVE_INST_VALU"))
        if val("SQ_LDS_IDX_ACTIVE") and ns("SQ_LDS_IDX_ACTIVE"):
            out["lds_busy"] = val("SQ_LDS_IDX_ACTIVE") / (256 * clock * ns("SQ_LDS_IDX_ACTIVE"))
        if val("SQ_BUSY_CU_CYCLES") and ns("SQ_BUSY_CU_CYCLES"):
            out["cu_busy"] = val("SQ_BUSY_CU_CYCLES") / (256 * clock * ns("SQ_BUSY_CU_CYCLES"))
    if val("SQ_BUSY_CU_CYCLES") and val("SQ_ACTIVE_INST_VALU"):
        out["valu_busy_while_cu_busy"] = 4.0 * val("SQ_ACTIVE_INST_VALU") / (4.0 * val("SQ_BUSY_CU_CYCLES"))
    if val("SQ_WAVES"):
        wv = val("SQ_WAVES")
        out["per_wave"] = {k.replace("SQ_INSTS_", "").lower(): val(k) / wv for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if val(k)}
        if val("SQ_WAVE_CYCLES"):
            out["per_wave"]["lifetime_cycles"] = 4.0 * val("SQ_WAVE_CYCLES") / wv
        out["per_wave"]["waves"] = wv
        out["per_wave"]["note"] = ("divided by SQ_WAVES; step_fast64 launches B + B/4 workgroups (slow-first dispatch, DESIGN.md 3.2 item 5), "
                                   "the B/4 first ones mostly exit at once: per wave that runs an env multiply by 1.25")
    return out


def main():
    d = sys.argv[1]
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {}
    for spec in sys.argv[2:]:
        key, tag = spec.split("=")
        out[key] = record("%s/%s_summary.txt" % (d, tag), head)
    json.dump(out, open("profiles/pmc_counters.json", "w"), indent=1)
    for k, v in out.items():
        print("%-12s %.4g ms  hbm %.4g GB  clock %s  valu %s  lds %s  cu %s" % (
            k, v.get("kernel_ms_profiled") or 0, (v.get("hbm_bytes_per_launch") or 0) / 1e9, v.get("clock_GHz"), v.get("valu_busy"),
            v.get("lds_busy"), v.get("cu_busy")))


if __name__ == "__main__":
    main()
