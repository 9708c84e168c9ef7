#!/bin/bash
# Round 5, first GPU call: parity suite on both routes, the bench line, the issue-rate micro-benchmarks, the per-dispatch overlap trace.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "in_flight", "overlap_factor")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["single_stream"], d["repetitions"]["ms_per_step_each"])
print({k: (v["ms_per_batch"], v["value"]) for k, v in d.get("secondary", {}).items()})
print(d.get("config4"))
PY
./tools/micro/fp64_issue > $O/r5_fp64_issue.txt 2>&1; cat $O/r5_fp64_issue.txt
./tools/micro/dpp_fmac > $O/r5_dpp_fmac.txt 2>&1; tail -8 $O/r5_dpp_fmac.txt
( cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_r5_overlap
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5_overlap -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 > $O/overlap_trace.log 2>&1 )
python tools/overlap_trace.py $R/gpurun_out/prof_r5_overlap k_linear_tft_pose_rows $O/r5_headline
