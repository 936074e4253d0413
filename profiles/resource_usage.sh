#!/bin/bash
# Register / spill / LDS / occupancy report of every step-kernel instantiation, from the
# compiler's own remarks (hipcc cross-compiles without a GPU).  Usage:
#   bash profiles/resource_usage.sh [out.txt] [extra -D flags...]
OUT=${1:-/dev/stdout}; shift
cd "$(dirname "$0")/.."
for tu in k_fast64 k_wide2 k_wide4 k_general k_observe k_large diral_env; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC "$@" \
    -Rpass-analysis=kernel-resource-usage -c diral_amd/csrc/$tu.hip -o /dev/null 2>&1 |
  python3 -c '
import re, sys, subprocess
txt = sys.stdin.read()
for blk in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = blk.split("\n")[0].split()[0].strip()
    def g(k):
        m = re.search(k + r": (\d+)", blk); return int(m.group(1)) if m else -1
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void diral::", "")
    print("%-52s vgpr %3d spill %2d  sgpr %3d spill %2d  scratch %3d B/lane  occupancy %d  lds %d" % (
        dn, g("    VGPRs"), g("VGPRs Spill"), g("TotalSGPRs"), g("SGPRs Spill"), g("ScratchSize \[bytes/lane\]"), g("Occupancy \[waves/SIMD\]"), g("LDS Size \[bytes/block\]")))
'
done > "$OUT"
