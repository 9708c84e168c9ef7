#!/bin/bash
# Round-5 soak of the linear paths with the rows kernels as the default route: the two settings of profiles/r3_soak_linear_parity.txt and the
# fast-vs-exact soak of profiles/r3_soak_fast_vs_exact.txt.  Outputs under gpurun_out/.
R=$GRAFT_REPO_ROOT
{
echo "# tools/soak_linear_parity.py on MI355X (gpurun), round 5 (LinearTFT / LinearF: four triplets per wavefront for N >= 12; minimal samples: exact kernels): HIP (C ABI) vs numpy oracle; SOAK_N=7,8,9 SOAK_NOISE=0.5,1.0,2.0 SOAK_B=300"
SOAK_N=7,8,9 SOAK_NOISE=0.5,1.0,2.0 SOAK_B=300 timeout 1200 python3 $R/tools/soak_linear_parity.py 2>&1 | grep -v amdgpu.ids
echo "# SOAK_N=10,12,15,31,64,65,127,200,201,257,511 SOAK_NOISE=0.0,0.5,2.0 SOAK_B=60 (labels: N>=15 = every N > 9)"
SOAK_N=10,12,15,31,64,65,127,200,201,257,511 SOAK_NOISE=0.0,0.5,2.0 SOAK_B=60 timeout 1500 python3 $R/tools/soak_linear_parity.py 2>&1 | grep -v amdgpu.ids
} > $R/gpurun_out/r5_soak_linear_parity.txt
timeout 1200 python3 $R/tools/soak_fast_vs_exact.py 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/r5_soak_fast_vs_exact.txt
tail -5 $R/gpurun_out/r5_soak_linear_parity.txt; tail -8 $R/gpurun_out/r5_soak_fast_vs_exact.txt
# the Gauss-Helmert kernels against the 50-digit fixtures after the round's changes (Pi / PiCol / FaugPapa: matrix-core sums; Ressl / Nordberg: fused check)
for f in "" "--nordberg" "--faugpapa" "--pi"; do
  timeout 900 python3 $R/tools/diag_gh_noise_mp.py $f 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/r5_gh_noise_mp${f/--/_}.txt
  tail -4 $R/gpurun_out/r5_gh_noise_mp${f/--/_}.txt
done
timeout 900 python3 $R/tools/diag_gh_noise_picol.py 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/r5_gh_noise_mp_picol.txt; tail -5 $R/gpurun_out/r5_gh_noise_mp_picol.txt
