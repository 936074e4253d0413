#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun, from the repo root):
#   bash profiles/run_profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/ : kernel-trace stats + separate PMC passes.
# Counters are collected in their OWN runs (never with sys/hip traces).
# Every pass runs `bench.py --lean`: its untimed pre-roll (>= 60 slots and >= 0.3 s) precedes
# the timed launches, and summarize_profile.py averages the LAST $LAST dispatches of the step
# kernel only (steady state: the timed launches of the pass).
set -u
TAG=${1:-r02}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# the kernel-trace pass profiles the SAME command the driver runs (default steps/warmup), minus the host-side extras
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --lean "$@" > $OUT/trace.log 2>&1
pmc() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- \
    python $R/bench.py --steps 40 --warmup 0 --lean ${EXTRA:-} > $OUT/$name.log 2>&1
}
EXTRA="$*"
pmc pmc_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pmc pmc_sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
if [ "${FULL_PMC:-0}" = 1 ]; then
pmc pmc_sq3 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_MISC
pmc pmc_sq4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT
fi
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_grbm GRBM_GUI_ACTIVE
cd $R
echo "CSRC_SHA $(python -c 'from diral_amd.build import source_digest; print(source_digest())')  COMMIT ${PMC_COMMIT:-unknown}" > $OUT/summary.txt
LAST=40 python profiles/summarize_profile.py $OUT >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the raw per-dispatch tables have done their job (summary.txt; trace/*kernel_stats.csv is kept): gpurun merges at most 64 MiB back
[ "${KEEP_RAW:-0}" = 1 ] || find $OUT -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv" | xargs rm -f
