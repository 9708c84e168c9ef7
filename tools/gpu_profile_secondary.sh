#!/bin/bash
# rocprofv3 kernel-trace stats of the secondary methods (10 k x 200 batch): one CSV per method under gpurun_out/prof_secondary/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_secondary; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in LinearFPoseEstimation ResslTFTPoseEstimation NordbergTFTPoseEstimation FaugPapaTFTPoseEstimation PiPoseEstimation PiColPoseEstimation OptimFPoseEstimation; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -- python3 $R/tools/bench_one.py $m 10 > $OUT/$m.log 2>&1
  f=$(find $OUT/$m -name "*kernel_stats.csv" | head -1)
  grep -E "^\"Name\"|tff::" $f > $OUT/${m}_kernel_stats.csv
done
cd $R; head -4 $OUT/*_kernel_stats.csv | cut -c1-160
