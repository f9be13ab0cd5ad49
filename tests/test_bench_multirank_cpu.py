"""bench.py's multi-rank branch at world size 2 on CPU (gloo): launched exactly as the driver launches it
(`python -m torch.distributed.run ... bench.py --gpus 2 ...`), with the stub engine (DFX_BENCH_STUB=1) standing
in for the GPU.  Checks the orchestration the driver's 8-GPU run depends on: rendezvous without RCCL, weak
scaling (one clip per rank) and strong scaling (--split clip: ONE clip split by denseflow_amd.shard.shard_pairs
into contiguous pair ranges that cover every flow exactly once), one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra):
    env = dict(os.environ, DFX_BENCH_STUB="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--frames", "41", "--width", "64", "--height", "48",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("step", [1, -2])
def test_weak_scaling_line_world2(step):
    out = _run(2, ["--step", str(step)])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["data"] == "stub"
    assert out["config"]["pairs_per_step"] == 2 * (41 - abs(step))  # every rank owns a whole clip
    assert out["value"] > 0 and out["steps"] == 2


@pytest.mark.parametrize("step", [1, 3])
def test_strong_scaling_line_world2_splits_one_clip(step):
    out = _run(2, ["--split", "clip", "--step", str(step)])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["pairs_per_step"] == 41 - abs(step)  # the ranks' ranges add up to the clip's flows
    assert "ONE clip split into 2" in out["config"]["workload"]


def test_single_process_stub_line():
    env = dict(os.environ, DFX_BENCH_STUB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--frames", "9",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["metric"].startswith("STUB") and "pcie_inclusive" not in out
