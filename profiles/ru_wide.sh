#!/bin/bash
# resource usage of the benchmarked wide instantiations (quick compile check): bash profiles/ru_wide.sh [-D flags]
cd "$(dirname "$0")/../diral_amd/csrc"
for t in k_wide2 k_wide4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC "$@" -c --cuda-device-only $t.hip -o /tmp/$t.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|step_wide_kernelILi[24]ELb0ELb1ELb0ELb0ELb[01]ELb[01]EE" -A9 | grep -E "error|Name|VGPRs:|Spill|Scratch" | sed 's/.*remark: *//' | head -24 &
done; wait
