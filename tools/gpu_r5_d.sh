#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5; mkdir -p $O
cd $R
for m in ResslTFTPoseEstimation NordbergTFTPoseEstimation; do timeout 300 python tools/gh_phase_profile.py 10000 200 0 $m; done > $O/r5_gh_phases.txt 2>&1
timeout 300 python tools/fp_phase_profile.py >> $O/r5_gh_phases.txt 2>&1
cat $O/r5_gh_phases.txt
timeout 900 python -m pytest tests/test_gpu_gh_noise.py tests/test_gpu_dropin_single.py -m gpu -q -x --timeout 900 -k "Faug or faug" > $O/pytest_gpu_d.log 2>&1; tail -3 $O/pytest_gpu_d.log
timeout 600 python tools/bench_methods.py > $O/bench_methods_d.txt 2>&1; grep -E "reconst=0|Bundle" $O/bench_methods_d.txt
