#!/bin/bash
# kernel-trace of the FaugPapa path (which kernel takes the time)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_fp; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/time_methods.py ${1:-200} FaugPapaTFTPoseEstimation > $OUT/trace.log 2>&1
cd $R
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cut -c1-220 $f | head -12
tail -2 $OUT/trace.log
