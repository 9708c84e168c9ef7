#!/bin/bash
# Round 5, second GPU call: the moments pre-kernel A/B over N, route tests, refined issue-rate micro-benchmark, bench.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python tools/ab_pre.py 10000 30 LinearTFTPoseEstimation 12 32 48 64 100 200 300 500 1000 > $O/r5_ab_pre.txt 2>&1; cat $O/r5_ab_pre.txt
timeout 300 python tools/ab_pre.py 10000 8 ResslTFTPoseEstimation 60 200 > $O/r5_ab_pre_ressl.txt 2>&1; cat $O/r5_ab_pre_ressl.txt
./tools/micro/fp64_issue > $O/r5_fp64_issue.txt 2>&1; cat $O/r5_fp64_issue.txt
timeout 1500 python -m pytest tests/test_gpu_rows.py tests/test_gpu_dropin_single.py tests/test_gpu_parity.py tests/test_gpu_gh_noise.py -m gpu -q -x --timeout 900 > $O/pytest_gpu_b.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_b.log
tail -5 $O/pytest_gpu_b.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_b.json 2> $O/bench_b.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_b.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "in_flight", "overlap_factor")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["single_stream"], d["repetitions"]["ms_per_step_each"])
print({k: (round(v["ms_per_batch"], 4), round(v["value"] / 1e6, 3), v["mean_iterations"]) for k, v in d.get("secondary", {}).items()})
print({k: v["value"] for k, v in d.get("n_sweep", {}).items()})
PY
