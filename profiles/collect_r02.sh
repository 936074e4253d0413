#!/bin/bash
# Copy the summaries of a gpurun profiling session (profiles/session_r02.sh: run_profile.sh tags c2_chobs1,
# c2_chobs0, c3_chobs1, c3_chobs0, c5_chobs1, c5_chobs0, c4shard; bench lines; the rollout example) from
# gpurun_out/ into profiles/r02/ and rebuild profiles/pmc_traffic.json.  Run in the build container.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r02
rm -f profiles/r02/c3_chobs1_traffic.txt profiles/r02/c5_chobs1_traffic.txt
for t in c2_chobs1 c2_chobs0 c3_chobs1 c3_chobs0 c5_chobs1 c5_chobs0 c4shard; do
  cp gpurun_out/prof_$t/summary.txt profiles/r02/${t}_summary.txt
  find gpurun_out/prof_$t/trace -name "*kernel_stats.csv" -exec cp {} profiles/r02/${t}_kernel_stats.csv \;
done
cp gpurun_out/rollout_r02.txt profiles/r02/rollout_example.txt
for f in full driverlike c3 c3_nochobs c5 c5_nochobs; do cp gpurun_out/bench_r02_$f.json profiles/r02/bench_$f.json; done
bash profiles/resource_usage.sh profiles/r02/resource_usage.txt
python3 - <<'PY'
import json, re, subprocess
out = {}
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
rule = "(2*FETCH_SIZE + WRITE_SIZE)*1024; gfx950 FETCH_SIZE reports half of a coalesced read (MI355X_MICROARCH.md, HBM)"
def add(key, summary):
    txt = open(summary).read()
    f = float(re.search(r"FETCH_SIZE\s+mean=([0-9.e+]+)", txt).group(1)); w = float(re.search(r"WRITE_SIZE\s+mean=([0-9.e+]+)", txt).group(1))
    out[key] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024.0, "rule": rule,
                "source": summary + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean over the last 40 launches after "
                          "bench.py's pre-roll; kernels of commit " + head + ")"}
add("c2", "profiles/r02/c2_chobs1_summary.txt"); add("c2_nochobs", "profiles/r02/c2_chobs0_summary.txt")
add("c3_nochobs", "profiles/r02/c3_chobs0_summary.txt"); add("c5_nochobs", "profiles/r02/c5_chobs0_summary.txt")
add("c3", "profiles/r02/c3_chobs1_summary.txt"); add("c5", "profiles/r02/c5_chobs1_summary.txt")
add("c4shard", "profiles/r02/c4shard_summary.txt")
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, "%.4g GB" % (v["hbm_bytes_per_launch"] / 1e9))
PY
python3 - <<'PY'
import re
for t, in (("c2_chobs1",), ("c2_chobs0",), ("c3_chobs1",), ("c3_chobs0",), ("c5_chobs1",), ("c5_chobs0",), ("c4shard",)):
    txt = open("profiles/r02/%s_summary.txt" % t).read()
    ns = float(re.search(r"steady state:.*avg_ns=([0-9.]+) ", txt).group(1))
    g = lambda k: float(re.search(k + r"\s+mean=([0-9.e+]+)", txt).group(1))
    w = g("SQ_WAVES"); cyc = ns * 1e-9 * 2.1e9
    print("%-10s %.1f us | VALU/SALU/LDS per wave %.0f / %.0f / %.0f | VALU busy %.2f LDS busy %.2f (conflicts %.0f%%) | 2*fetch %.1f MB write %.1f MB -> %.2f TB/s" % (
        t, ns / 1e3, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w,
        4 * g("SQ_ACTIVE_INST_VALU") / (1024 * cyc), g("SQ_LDS_IDX_ACTIVE") / (256 * cyc), 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
        2 * g("FETCH_SIZE") * 1024 / 1e6, g("WRITE_SIZE") * 1024 / 1e6, (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / ns / 1e3))
PY
