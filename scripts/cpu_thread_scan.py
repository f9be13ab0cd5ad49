#!/usr/bin/env python3
"""How many OpenMP threads should the CPU comparator use on this box?  Prints the cgroup CPU limits and times one
1080p pair of oracle/cpu_tvl1_baseline.c (through oracle/cpu_bench_child.py) for several thread counts."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from denseflow_amd.synth import SynthClip  # noqa: E402

for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.cpus"):
    try:
        print(p, "=", open(p).read().strip())
    except OSError:
        pass
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print(subprocess.run(["lscpu"], capture_output=True, text=True).stdout[:1500])
except OSError:
    pass
clip = SynthClip(1920, 1080, 2)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "f.npy")
    np.save(path, np.stack(clip.frames(3)))
    for n in [int(v) for v in (sys.argv[1:] or ["8", "16", "32", "64", "128"])]:
        for policy in ("passive", "active"):
            env = dict(os.environ, DFX_CPU_THREADS=str(n), OMP_WAIT_POLICY=policy)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench_child.py"), path, "cpu_tvl1", "6"],
                               capture_output=True, text=True, env=env)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"threads {n:4d} wait {policy:8s}: {d['value']:.3f} pairs/s  runs {d['runs']}", flush=True)
            except Exception:
                print("failed", n, r.stdout[-300:], r.stderr[-300:], flush=True)
