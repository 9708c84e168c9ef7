#!/bin/bash
# Round 5: the whole GPU evidence in one call -- parity suite, bench line, rocprofv3 evidence for every method, overlap trace, soak.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_full.log
tail -6 $O/pytest_gpu_full.log
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"
bash tools/gpu_profile_r5.sh headline headline1 ressl nordberg faugpapa pi picol linearf optimf config4tft config4f 2>&1 | grep -v "^$" | tail -15
( cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_r5_overlap
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5_overlap -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 > $O/overlap_trace.log 2>&1 )
python tools/overlap_trace.py $R/gpurun_out/prof_r5_overlap k_linear_tft_pose_rows $O/r5_headline | tail -4
timeout 900 python tools/soak_linear_parity.py > $O/r5_soak_linear_parity.txt 2>&1; tail -5 $O/r5_soak_linear_parity.txt
timeout 600 python tools/bench_n_sweep.py > $O/r5_n_sweep.txt 2>&1; tail -8 $O/r5_n_sweep.txt
