#!/bin/bash
# Round-5 rocprofv3 evidence for every method bench.py reports (kernel trace + four separate PMC passes each; program directly after `--`).
# Usage: bash tools/gpu_profile_r5.sh [tags...]   tags: headline headline1 ressl nordberg faugpapa pi picol linearf optimf config4tft config4f config4count
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
run_one() {   # tag, command, kernel substring, units per launch, algorithmic bytes per launch, waves per SIMD
    local TAG=$1 CMD=$2 KERN=$3 UNITS=$4 ALG=$5 OCC=$6
    local OUT=$R/gpurun_out/prof_r5_$TAG; rm -rf $OUT; mkdir -p $OUT
    ( cd /tmp && export TMPDIR=/tmp
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
      timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
      timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
      timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
      timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1 )
    python3 $R/tools/summarize_r3.py $OUT "$KERN" $UNITS $ALG $OCC > $R/gpurun_out/r5/r5_${TAG}_summary.json
    f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); grep -E "^\"Name\"|tff::" $f > $R/gpurun_out/r5/r5_${TAG}_kernel_stats.csv
    python3 - <<PY
import json
s = json.load(open("$R/gpurun_out/r5/r5_${TAG}_summary.json"))
print("$TAG: %.1f us, VALU/unit %.0f, busy %.2f, waiting %.2f, lanes %.2f, HBM x%.2f, issue fraction %.2f" % (s.get("average_ns", 0) / 1e3, s.get("valu_instructions_per_unit", 0),
      s.get("valu_busy_fraction", 0), s.get("waiting_fraction", 0), s.get("lane_utilisation", 0), s.get("hbm_bytes_over_algorithmic", 0), s.get("fraction_of_issue_limit", 0)))
PY
}
ALG=$((10000 * (48 * 200 + 216 + 216 + 192)))
for T in ${@:-headline headline1 ressl nordberg faugpapa pi picol linearf optimf config4tft config4f}; do
  case $T in
    headline)   run_one headline "python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3" "k_linear_tft_pose_rows" 10000 $ALG 2;;          # bench.py as the driver runs it: two streams, consecutive batches overlap
    headline1)  run_one headline1 "python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --streams 1" "k_linear_tft_pose_rows" 10000 $ALG 2;;   # one stream: a launch has the GPU to itself
    ressl)      run_one ressl "python3 $R/tools/bench_one.py ResslTFTPoseEstimation 8" "k_gh_block" 10000 $ALG 2;;
    nordberg)   run_one nordberg "python3 $R/tools/bench_one.py NordbergTFTPoseEstimation 8" "k_gh_block" 10000 $ALG 2;;
    faugpapa)   run_one faugpapa "python3 $R/tools/bench_one.py FaugPapaTFTPoseEstimation 6" "k_fp_block" 10000 $ALG 2;;
    pi)         run_one pi "python3 $R/tools/bench_one.py PiPoseEstimation 8" "k_pi_block" 10000 $ALG 2;;
    picol)      run_one picol "python3 $R/tools/bench_one.py PiColPoseEstimation 6" "k_pi_block" 10000 $ALG 2;;
    linearf)    run_one linearf "python3 $R/tools/bench_one.py LinearFPoseEstimation 20" "k_linear_f_pose_rows" 10000 $ALG 2;;
    optimf)     run_one optimf "python3 $R/tools/bench_one.py OptimFPoseEstimation 10" "k_optimf_refine" 10000 $ALG 2;;
    config4tft) run_one config4tft "python3 $R/tools/config4_split.py 1000000" "k_linear_tft_pose_rows_exact" 1000000 $((1000000 * 440)) 2;;
    config4count) run_one config4count "python3 $R/tools/config4_split.py 1000000" "k_inlier_count_rows" 1000000 $((1000000 * 196)) 2;;   # per hypothesis: two poses in (192 B), one int32 out; the 19 KB scene is staged once per workgroup
    config4f)   run_one config4f "python3 $R/tools/config4_split.py 1000000" "k_linear_f_pose_rows_exact" 1000000 $((1000000 * 444)) 2;;
  esac
done
