"""bench.py's two-streams x overlapped-gather pipeline against RCCL itself, on the one GPU a test box has: `--force-process-group` creates the
nccl (= RCCL) process group as a clique of one and sends every step's records through `all_gather_into_tensor(async_op=True)` exactly as an
N-GPU run does (tft_vs_fund_amd/dist.py::OverlappedGather: Work.wait() before a buffer is reused, the gather ordered against the step's own
stream).  Every step zeroes its record buffer on its stream before the launch, so a gather that ran ahead of the kernels would deliver zeros;
the records the LAST step's gather delivered must equal the one-stream, no-collective result bit for bit.
This is the stream / process-group ordering an 8-GPU run depends on; it is NOT a multi-GPU measurement (README: UNMEASURED on more than one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("streams", [2, 1])
def test_bench_pipeline_through_a_clique_of_one_rccl_group(streams):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--streams", str(streams), "--steps", "8", "--warmup", "2", "--reps", "2",
                        "--batch", "4099", "--ncorr", "60", "--force-process-group", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=560, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["process_group_check"]
    assert c["forced"] is True and c["backend"] == "nccl" and c["world"] == 1 and c["streams"] == streams
    assert c["gathered_equals_single_stream"] is True and c["records"] == 51 * 4099
    assert d["n_gpus"] == 1 and d["config"]["failed_triplets"] == 0 and d["in_flight"] == streams
