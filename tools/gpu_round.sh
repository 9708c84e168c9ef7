#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, phase profile, rocprof kernel trace.  Outputs under gpurun_out/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
timeout 300 python tools/phase_profile.py > gpurun_out/phase.log 2>&1; cat gpurun_out/phase.log
if [ "$1" == "prof" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 > $R/gpurun_out/prof.log 2>&1
cd $R
for f in $(find gpurun_out/prof -name '*kernel_stats.csv'); do head -3 $f | cut -c1-200; done
fi
