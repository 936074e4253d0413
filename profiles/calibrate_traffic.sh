#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts, per access width (GPU box, from the repo root):
#   bash profiles/calibrate_traffic.sh      -> gpurun_out/traffic_calib/, profiles/traffic_calibration.json
set -u
R=$PWD
OUT=$R/gpurun_out/traffic_calib
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 profiles/micro/traffic_calib.hip -o /tmp/traffic_calib 2>/dev/null || exit 1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- /tmp/traffic_calib > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- /tmp/traffic_calib > $OUT/write.log 2>&1
cd $R
python profiles/traffic_calib.py $OUT $OUT/traffic_calibration.json
find $OUT -name "*_kernel_trace.csv" | xargs rm -f
