#!/bin/bash
# The four synthetic sweeps of experiments.m (n_sim per interval value given as $1, default 100) and experiments_real.m on the EPFL
# fixture, all eight methods + BundleAdjustment, JSON under gpurun_out/results/.
R=${GRAFT_REPO_ROOT:-.}
NSIM=${1:-100}
mkdir -p $R/gpurun_out/results
cd $R
for opt in noise focal points angle; do
  python -m tft_vs_fund_amd.experiments --option $opt --n-sim $NSIM --out gpurun_out/results/synthetic_$opt.json
done
python -m tft_vs_fund_amd.experiments --real tests/golden/epfl.npz --out gpurun_out/results/real_epfl_fixture.json
ls -la gpurun_out/results
