#!/bin/bash
# Round-2 GPU check: parity tests (all, no -x), smoke, bench, minimal-sample soak, config 4 timing.  Outputs under gpurun_out/.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
SOAK_N=${SOAK_N:-7,8,9} SOAK_B=${SOAK_B:-100} timeout 900 python tools/soak_linear_parity.py > gpurun_out/soak.log 2>&1; tail -30 gpurun_out/soak.log
timeout 300 python tools/config4_ransac.py > gpurun_out/config4.log 2>&1; tail -5 gpurun_out/config4.log
