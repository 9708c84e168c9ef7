#!/bin/bash
# FaugPapa block kernel on the GPU: 50-digit deviation table, timing at several N
mkdir -p gpurun_out
timeout 600 python tools/diag_gh_noise_mp.py --faugpapa > gpurun_out/fp_noise.log 2>&1; tail -12 gpurun_out/fp_noise.log
for n in 200 100 12; do timeout 300 python tools/time_methods.py $n FaugPapaTFTPoseEstimation ResslTFTPoseEstimation >> gpurun_out/fp_time.log 2>&1; done; cat gpurun_out/fp_time.log
TFF_VARIANT=2 timeout 300 python tools/time_methods.py 200 FaugPapaTFTPoseEstimation >> gpurun_out/fp_time_old.log 2>&1; cat gpurun_out/fp_time_old.log
