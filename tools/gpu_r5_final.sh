#!/bin/bash
# Round 5, end of round: parity suite, bench line, refreshed rocprofv3 evidence for what changed after gpu_r5_full.sh (headline, config 4), overlap trace.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_final.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_final.log
grep -E "passed|failed" $O/pytest_gpu_final.log | tail -2
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"
bash tools/gpu_profile_r5.sh headline headline1 config4tft config4f linearf pi picol faugpapa 2>&1 | grep -v "^$" | tail -10
( cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_r5_overlap
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5_overlap -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 > $O/overlap_trace.log 2>&1 )
python tools/overlap_trace.py $R/gpurun_out/prof_r5_overlap k_linear_tft_pose_rows $O/r5_headline | tail -3
python tools/config4_split.py 1000000 2>&1 | tail -4
