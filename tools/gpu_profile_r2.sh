#!/bin/bash
# Round-2 rocprofv3 evidence (program directly after `--`; --pmc passes separate from each other and without trace domains
# other than --kernel-trace).  $1 = headline | ressl | faugpapa | nordberg
R=$GRAFT_REPO_ROOT
WHAT=${1:-headline}
OUT=$R/gpurun_out/prof_r2_$WHAT; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
case $WHAT in
  headline) CMD="python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3"; export PROFILE_KERNEL=k_linear_tft_pose;;
  ressl)    CMD="python3 $R/tools/bench_one.py ResslTFTPoseEstimation 10"; export PROFILE_KERNEL="k_gh_block";;
  nordberg) CMD="python3 $R/tools/bench_one.py NordbergTFTPoseEstimation 10"; export PROFILE_KERNEL="k_gh_block";;
  faugpapa) CMD="python3 $R/tools/bench_one.py FaugPapaTFTPoseEstimation 6"; export PROFILE_KERNEL="k_gh_block";;
esac
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
cd $R
python3 tools/summarize_profile.py gpurun_out/prof_r2_$WHAT > gpurun_out/prof_r2_$WHAT/summary.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); grep -E "^\"Name\"|tff::" $f > $OUT/kernel_stats.csv
head -6 $OUT/kernel_stats.csv | cut -c1-200
python3 - <<PY
import json
s = json.load(open("gpurun_out/prof_r2_$WHAT/summary.txt"))
for grp in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for k, v in s.get(grp, {}).items():
        print(grp, k[-70:], "%.4g" % v["mean_per_dispatch"], v["dispatches"])
PY
