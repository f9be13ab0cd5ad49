"""bench.py's multi-rank branch with REAL engines: two processes under `python -m torch.distributed.run`, as the driver
launches an N-GPU run, both on the one GPU of the test box (DFX_BENCH_SHARE_GPU=1 maps LOCAL_RANK onto the devices that
exist — a path test, not a scaling measurement).  Covers what the CPU test with the stub engine cannot: two HIP
processes with their own handles next to the gloo rendezvous, weak scaling (a clip per rank) and strong scaling (one clip
split by pair ranges), the aggregate line of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra):
    env = dict(os.environ, DFX_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--frames", "21", "--width", "320", "--height", "240", "--no-cpu-baseline", "--no-pcie"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    return json.loads(lines[0])


def test_two_ranks_weak_and_strong_lines():
    weak = _run([])
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["data"] == "synthetic"
    assert weak["config"]["pairs_per_step"] == 2 * 20 and weak["value"] > 0
    assert weak["metric"].startswith("NOT A SCALING MEASUREMENT")
    strong = _run(["--split", "clip", "--step", "2"])
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"
    assert strong["config"]["pairs_per_step"] == 21 - 2
    assert strong["roofline"]["achieved"] > 0


def test_self_launched_two_ranks():
    """`python3 bench.py --gpus 2` as the driver types it (no launcher, no WORLD_SIZE): bench.py starts its own ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DFX_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames",
           "21", "--width", "320", "--height", "240", "--no-cpu-baseline", "--no-pcie"]
    for extra, scaling, pairs in (([], "weak", 40), (["--split", "clip"], "strong", 20)):
        r = subprocess.run(cmd + extra, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["config"]["pairs_per_step"] == pairs
        assert out["value"] > 0 and out["data"] == "synthetic"


def test_self_launched_two_ranks_each_with_a_list_of_short_clips():
    """--clips: every rank joins its share of a list of short clips into one FlowBuffer (dfx_next_segments) — BASELINE
    configs[3] sharded over the GPUs of a node."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DFX_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "12",
           "--clips", "3", "--width", "224", "--height", "224", "--no-cpu-baseline", "--no-pcie", "--no-others"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["pairs_per_step"] == 2 * 3 * 11
    assert "3 clips per GPU" in out["config"]["workload"] and out["value"] > 0
