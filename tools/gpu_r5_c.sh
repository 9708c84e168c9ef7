#!/bin/bash
# Round 5: Gauss-Helmert block kernels after a change -- 50-digit gates on both routes, route / drop-in tests, the method timings of bench.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5; mkdir -p $O
TAG=${1:-c}
cd $R
timeout 1500 python -m pytest tests/test_gpu_gh_noise.py tests/test_gpu_dropin_single.py tests/test_gpu_rows.py tests/test_nordberg_divergence.py -m gpu -q -x --timeout 900 > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_$TAG.log
tail -15 $O/pytest_gpu_$TAG.log
timeout 600 python tools/bench_methods.py > $O/bench_methods_$TAG.txt 2>&1; cat $O/bench_methods_$TAG.txt
