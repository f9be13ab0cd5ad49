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
    assert out["n_gpus"] == 1 and out["metric"].startswith("STUB") and "pcie_inclusive" not in out["config"]


def _self_launched(world, extra):
    """`python3 bench.py --gpus N ...` with NO launcher around it and no WORLD_SIZE in the environment — the way the
    driver invokes the bench (BENCH_r02.cmd).  bench.py must start its own N ranks (it used to run ONE rank
    and print "n_gpus": 1, which would have voided a driver SCALE run)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["DFX_BENCH_STUB"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--frames", "41", "--width", "64", "--height", "48", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])
def test_self_launch_weak(world):
    out = _self_launched(world, [])
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["data"] == "stub"
    assert out["config"]["pairs_per_step"] == world * 40


def test_self_launch_list_of_short_clips_per_rank():
    """--clips N: every rank owns N clips joined into one FlowBuffer (the videolist of BASELINE configs[3] sharded over the
    GPUs); pairs never cross a clip boundary, so a rank counts N * (frames - |step|) of them."""
    out = _self_launched(2, ["--clips", "3", "--step", "2"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["pairs_per_step"] == 2 * 3 * (41 - 2)
    assert "clips x 3 in one FlowBuffer" in out["config"]["workload"] and "3 clips per GPU" in out["config"]["workload"]


def test_self_launch_strong_splits_one_clip():
    out = _self_launched(2, ["--split", "clip", "--step", "-2"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["pairs_per_step"] == 41 - 2
    assert "ONE clip split into 2" in out["config"]["workload"]


def test_self_launch_propagates_a_failing_rank():
    """A rank that dies must end the whole command with a non-zero status (not hang the others in the barrier)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DFX_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--frames", "9", "--split", "clip", "--step", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=120, cwd=ROOT)
    assert r.returncode != 0  # shard_pairs rejects step 0 on every rank


def test_full_line_schema_and_size():
    """The N = 1 line as the driver records it (BENCH_r03.parsed.config kept scalars only and the 8 KB
    stdout tail cut the line).  DFX_BENCH_STUB=full walks the stub engine through EVERY leg — PCIe-inclusive, the other
    BASELINE configurations, the CPU comparator's real code on a tiny sample — so the schema is the measured line's:
    every FLAT_KEYS entry is a float directly under `config`, the joined 224x224 leg is 64 clips (BASELINE configs[3]'s
    per-GPU share), and the whole line is under 6 KB."""
    import math

    sys.path.insert(0, ROOT)
    import bench

    env = dict(os.environ, DFX_BENCH_STUB="full")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) < 6144, len(lines[0])
    out = json.loads(lines[0])
    cfg = out["config"]
    for k in bench.FLAT_KEYS:
        if k.endswith("traffic_frac") or k.endswith("useful_frac") or k == "parity_iters_equal":
            continue  # PMC- / dfx_stats-derived: only with a live pass on a GPU; iteration tables: GPU only
        assert isinstance(cfg.get(k), float) and math.isfinite(cfg[k]), (k, cfg.get(k))
    assert all(not isinstance(v, (dict, list)) or k in ("pcie_inclusive", "other_workloads") for k, v in cfg.items())
    # whatever the driver's record truncates must be the least important — config's keys are ordered:
    # workload, arithmetic, the parity proof, the other BASELINE configurations' rates and fractions, the non-converging
    # content, the other hypot readings; only then the PCIe-inclusive variants, the run's counts and the nested objects
    keys = list(cfg)
    assert keys[:2] == ["workload", "arithmetic"]
    flat_present = [k for k in bench.FLAT_KEYS if k in cfg]
    assert keys[2:2 + len(flat_present)] == flat_present
    # on a GPU parity_iters_equal and one *_traffic_frac per leg (the bytes that physically moved, beside every frac that
    # can exceed 1: VERDICT r5 #3) join these, 22 in all in front of the PCIe variants
    assert flat_present[:14] == ["parity_pairs", "parity_max_abs", "parity_legs_max_abs",
                                 "farn_1080p_pairs_per_s", "farn_1080p_frac", "tvl1_224x64_pairs_per_s", "tvl1_224x64_frac",
                                 "brox_4k_s2_pairs_per_s", "brox_4k_s2_frac",
                                 "tvl1_1080p_hard_pairs_per_s", "tvl1_1080p_hard_frac",
                                 "tvl1_1080p_noexit_pairs_per_s", "tvl1_1080p_noexit_frac", "tvl1_1080p_noexit_of_ceiling"]
    assert bench.FLAT_KEYS.index("pcie_jpeg_pairs_per_s") == 21  # the 22nd key: what the driver's record has kept so far
    assert keys.index("pcie_inclusive") > keys.index("pairs_per_step") > keys.index("tvl1_libm_pairs_per_s")
    assert isinstance(out["parity_check"], dict) and {"pairs", "max_abs", "iters_equal"} <= set(out["parity_check"])
    legs = {leg["key"]: leg for leg in cfg["other_workloads"]}
    assert set(legs) == {"farn_1080p", "tvl1_224", "tvl1_224x64", "brox_4k_s2", "tvl1_1080p_hard", "tvl1_1080p_noexit",
                         "tvl1_sqrt", "tvl1_libm"}
    for k in ("farn_1080p", "tvl1_224", "tvl1_224x64", "brox_4k_s2"):
        assert "max_abs" in legs[k]["parity_check"], k
    assert "x 64 in one FlowBuffer" in legs["tvl1_224x64"]["workload"] and "19136 pairs/step" in legs["tvl1_224x64"]["workload"]
    assert "130-frame clip" in legs["brox_4k_s2"]["workload"] and "128 pairs/step" in legs["brox_4k_s2"]["workload"]
    for top in ("roofline", "cpu_baseline"):
        assert isinstance(out[top], dict)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"])


def _world8(cmd_prefix, env_drop):
    """The driver's exact SCALE command at N = 8 — `bench.py --gpus 8 --steps 20 --warmup 5`, the headline workload's
    default arguments — with the stub engine, confined to 16 CPUs like a box of this pool (`taskset`, as many as this
    container has when it has fewer).  One JSON line, "n_gpus": 8, every rank's 299 pairs counted, well under a minute."""
    import shutil
    import time

    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env["DFX_BENCH_STUB"] = "1"
    cmd = cmd_prefix + [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    ncpu = len(os.sched_getaffinity(0))
    if shutil.which("taskset"):
        cmd = ["taskset", "-c", f"0-{min(16, ncpu) - 1}"] + cmd
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240, cwd=ROOT)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["config"]["pairs_per_step"] == 8 * 299 and "1920x1080" in out["config"]["workload"]
    assert abs(out["value"] - 20 * 8 * 299 / (out["ms_per_step"] * 20e-3)) < 1e-6 * out["value"]
    assert "pcie_inclusive" not in out["config"] and "cpu_baseline" not in out  # N = 1 legs only
    assert wall < 60.0, wall
    return out


def test_world8_self_launched_as_the_driver_runs_it():
    """No launcher, no WORLD_SIZE: bench.py starts its own eight ranks on a free port."""
    _world8([sys.executable], ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"))


def test_world8_under_torch_distributed_run():
    """The launcher form of the contract: python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py."""
    _world8([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
             "127.0.0.1", "--master-port", str(_free_port())], ())
