#!/bin/bash
# rocprofv3 passes for the kernels BESIDE the step kernel (GPU box, from the repo root):
#   bash profiles/secondary_prof.sh <tag> <kernel-name substring> -- <command...>
# e.g.  bash profiles/secondary_prof.sh type1_c3 posdist_type1 -- env WORKLOADS=c3 MODE_FILTER=type-1 SLOTS=60 python profiles/secondary_modes.py
# Writes gpurun_out/prof_<tag>/summary.txt: kernel stats, then per launch of the named kernels the PMC means (each
# counter group in its OWN run, never combined with sys / hip traces) and the kernel duration inside each pass.
set -u
TAG=$1; KF=$2; shift 3
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD=("$@")
for i in "${!CMD[@]}"; do [ -f "$R/${CMD[$i]}" ] && CMD[$i]="$R/${CMD[$i]}"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "${CMD[@]}" > $OUT/trace.log 2>&1
pmc() { local name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- "${CMD[@]}" > $OUT/$name.log 2>&1; }
pmc pmc_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
pmc pmc_sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_grbm GRBM_GUI_ACTIVE
cd $R
KERNEL_FILTER=$KF LAST=40 python profiles/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv" | xargs rm -f
grep -E "posdist|sps_|driver_shape|observe|steady|mean=|PASS_NS" $OUT/summary.txt | head -60
