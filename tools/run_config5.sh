#!/bin/bash
# configs[4]: EPFL triplet lists x noise trials x seven methods (results/real_<dataset>_trials.json); GPU box.
for d in fountain herzjesu; do
  python -m tft_vs_fund_amd.experiments --real tests/golden/epfl_all.npz --dataset $d --noise-trials ${1:-1000} --sigma ${2:-0.5} --out gpurun_out/real_${d}_trials.json
  python - <<PY
import json
r = json.load(open("gpurun_out/real_${d}_trials.json"))
print("$d", r["n_trials"], "trials, sigma", r["sigma"], "triplets", len(r["triplets"]))
for m, s in r["summary"].items():
    print("  %-28s problems %6d solved %6d  repr %.3f  rot %.3f deg (median %.3f)  t %.3f deg  iter %.2f  %.0f problems/s" % (m, s["problems"], s["solved"], s["mean_repr_err"], s["mean_rot_err_deg"], s["median_rot_err_deg"], s["mean_t_err_deg"], s["mean_iter"], s["problems_per_s"]))
PY
done
