#!/bin/bash
# Round 5, second end-of-round pass (after the BundleAdjustment views kernel and the minimal-sample changes of the exact row kernels): parity suite, bench line,
# refreshed rocprofv3 evidence for the config-4 kernels and the headline, config-4 phase shares and split.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_final3.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_final3.log
grep -E "passed|failed" $O/pytest_gpu_final3.log | tail -2
timeout 600 python bench.py > $O/bench_final3.json 2> $O/bench_final3.err; echo "bench rc $?"
bash tools/gpu_profile_r5.sh headline headline1 config4tft config4f 2>&1 | grep -v "^$" | tail -6
python tools/config4_phase_profile.py 200000 > $O/r5_config4_phases.txt 2>&1; tail -30 $O/r5_config4_phases.txt
python tools/config4_split.py 1000000 2>&1 | tail -4
