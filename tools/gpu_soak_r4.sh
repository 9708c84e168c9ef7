#!/bin/bash
# Round-4 soak of the linear paths with the rows kernels as the default route: the two settings of profiles/r3_soak_linear_parity.txt and the
# fast-vs-exact soak of profiles/r3_soak_fast_vs_exact.txt.  Outputs under gpurun_out/.
R=$GRAFT_REPO_ROOT
{
echo "# tools/soak_linear_parity.py on MI355X (gpurun), round 4 (LinearTFT / LinearF: four triplets per wavefront for N >= 12; minimal samples: exact kernels): HIP (C ABI) vs numpy oracle; SOAK_N=7,8,9 SOAK_NOISE=0.5,1.0,2.0 SOAK_B=300"
SOAK_N=7,8,9 SOAK_NOISE=0.5,1.0,2.0 SOAK_B=300 timeout 1200 python3 $R/tools/soak_linear_parity.py 2>&1 | grep -v amdgpu.ids
echo "# SOAK_N=10,12,15,31,64,65,127,200,201,257,511 SOAK_NOISE=0.0,0.5,2.0 SOAK_B=60 (labels: N>=15 = every N > 9)"
SOAK_N=10,12,15,31,64,65,127,200,201,257,511 SOAK_NOISE=0.0,0.5,2.0 SOAK_B=60 timeout 1500 python3 $R/tools/soak_linear_parity.py 2>&1 | grep -v amdgpu.ids
} > $R/gpurun_out/r4_soak_linear_parity.txt
timeout 1200 python3 $R/tools/soak_fast_vs_exact.py 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/r4_soak_fast_vs_exact.txt
tail -5 $R/gpurun_out/r4_soak_linear_parity.txt; tail -8 $R/gpurun_out/r4_soak_fast_vs_exact.txt
