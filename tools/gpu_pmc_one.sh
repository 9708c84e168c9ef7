#!/bin/bash
# rocprofv3 evidence for one method (kernel trace + PMC passes; program directly after `--`).  $1 = method name, $2 = tag, [$3 = N]
R=$GRAFT_REPO_ROOT
M=${1:-FaugPapaTFTPoseEstimation}; TAG=${2:-fp}; N=${3:-200}
OUT=$R/gpurun_out/prof_r3_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/bench_one.py $M 6 10000 $N"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
cd $R
PROFILE_KERNEL=${PROFILE_KERNEL:-k_} python3 tools/summarize_profile.py gpurun_out/prof_r3_$TAG > gpurun_out/prof_r3_$TAG/summary.json
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); grep -E "^\"Name\"|tff::" $f > $OUT/kernel_stats.csv
cut -c1-160 $OUT/kernel_stats.csv | head -8
python3 - <<PY
import json
s = json.load(open("gpurun_out/prof_r3_$TAG/summary.json"))
for grp in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for k, v in s.get(grp, {}).items():
        print(grp, k[-60:], "%.4g" % v["mean_per_dispatch"], v["dispatches"])
PY
