"""bench.py's own main() and tools/config4_ransac.py at world size 2 on CPU (gloo), launched with the driver's command line, the HIP work
replaced by the scripts' test seams (--stub-compute / --stub): argument parsing, the RANK / WORLD_SIZE / MASTER_* rendezvous, rank-0 build +
barrier, the double-buffered step / all-gather pipeline, max-over-ranks timing, weak-scaling accounting and the single JSON line with
n_gpus = 2; for config 4 the contiguous sharding of an uneven hypothesis count and the order of the gathered counts.
The numbers these runs print are not measurements (the JSON says so); nothing here has run on two GPUs (README: UNMEASURED)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _torchrun(script, *args, nproc=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, script)] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


def test_bench_main_flow_at_world_size_two():
    B, K, W = 65, 4, 1
    d = _torchrun("bench.py", "--gpus", "2", "--steps", str(K), "--warmup", str(W), "--batch", str(B), "--ncorr", "20", "--stub-compute")
    assert d["n_gpus"] == 2 and d["steps"] == K and d["warmup"] == W and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["metric"].startswith("triplet-hypotheses/sec") and d["unit"] == "triplet-hypotheses/s" and d["dtype"] == "f64"
    assert d["stub"]["gather_check"] is True             # both ranks' records of the last step, complete and in rank order
    # whole-job aggregate: every rank owns a batch of its own (weak scaling), value = n_gpus * B * steps / max-over-ranks time
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * B) < 1e-6 * 2 * B
    assert d["config"]["batch_per_gpu"] == B and "all_gather" in d["config"]["gather"] and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["algorithmic_bytes_per_launch"] == B * (48 * 20 + 216 + 216 + 192)
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only


def test_bench_single_process_stub_has_no_collective():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-compute", "--steps", "3", "--warmup", "1", "--batch", "32", "--ncorr", "12"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["gather"] == "none (1 GPU)" and d["stub"]["gather_check"] is None


def test_config4_shard_and_count_gather_at_uneven_size():
    H = 1001                                             # 500 + 501 hypotheses
    d = _torchrun("tools/config4_ransac.py", "--hyp", str(H), "--stub")
    assert d["n_gpus"] == 2 and d["shard"] == [0, 500] and d["gathered"] == H and d["order_ok"] is True
    d3 = _torchrun("tools/config4_ransac.py", "--hyp", "7", "--stub", nproc=3)     # 2 + 2 + 3
    assert d3["n_gpus"] == 3 and d3["gathered"] == 7 and d3["order_ok"] is True
