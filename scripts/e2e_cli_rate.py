#!/usr/bin/env python3
"""End-to-end rate of the host shell (build/denseflow): .y4m clip in, flow_x/flow_y JPEGs out.

Measures what SURVEY.md §8f-1 is about: the save stage.  Three configurations per algorithm:
  ref-like : DF_HOST_BOUND=1 DF_ENCODE_THREADS=1  (the reference's host-side bounding, one encoder thread)
  host-par : DF_HOST_BOUND=1, default encoder threads
  dev-bound: bounding (and, with EXTRA_ARGS="-nw=.. -nh=..", the resize) on the GPU, JPEG encoders on the host (DF_HOST_JPEG=1)
  device   : bounding AND JPEG coding on the GPU (the default): the save stage only writes files
Every line also reports the CPU seconds (user + system, all threads) the shell spent per pair.
Usage: python scripts/e2e_cli_rate.py [W H NF] ; needs a GPU; writes under $TMPDIR.
"""
import os
import re
import resource
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, NF = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1920, 1080, 129)))
NCLIPS = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # BASELINE config 4 shape: many short clips in one list
algos = os.environ.get("ALGOS", "farn,tvl1").split(",")
exe = os.path.join(ROOT, "build", "denseflow")
tmp = tempfile.mkdtemp(prefix="dfe2e_")
clips = []
for c in range(NCLIPS):
    clip = os.path.join(tmp, "clip.y4m" if NCLIPS == 1 else f"clip{c:04d}.y4m")
    frames = SynthClip(W, H, 2 if NCLIPS == 1 else 1000 + c).frames_torch(NF, torch.device("cuda", 0)).cpu().numpy()
    with open(clip, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F30:1 Ip A1:1 Cmono\n".encode())
        for fr in frames:
            f.write(b"FRAME\n")
            f.write(fr.tobytes())
    clips.append(clip)
lst = os.path.join(tmp, "list.txt")
open(lst, "w").write("".join(c + "\n" for c in clips))
extra = os.environ.get("EXTRA_ARGS", "").split()  # e.g. EXTRA_ARGS="-nw=224 -nh=224": adds the resize comparison
configs = [("ref-like", {"DF_HOST_BOUND": "1", "DF_ENCODE_THREADS": "1", "DF_HOST_RESIZE": "1"}),
           ("host-par", {"DF_HOST_BOUND": "1", "DF_HOST_RESIZE": "1"}), ("dev-bound/host-jpeg", {"DF_HOST_JPEG": "1"}),
           ("device", {})]
if os.environ.get("CONFIGS"):  # e.g. CONFIGS="dev-bound/host-jpeg,device": long clips without the slow reference-like runs
    configs = [c for c in configs if c[0] in os.environ["CONFIGS"].split(",")]
if extra:
    configs.insert(2, ("dev-bound/host-resize", {"DF_HOST_RESIZE": "1", "DF_HOST_JPEG": "1"}))
print(f"{NCLIPS} clip(s) {W}x{H} x {NF} frames, host cores {os.cpu_count()}", flush=True)
for algo in algos:
    for tag, env in configs:
        out = os.path.join(tmp, f"out_{algo}_{tag}")
        ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        t0 = time.perf_counter()
        r = subprocess.run([exe, lst, "-o=" + out, "-a=" + algo, "-s=1", "-b=20"] + extra, capture_output=True, text=True,
                           env={**os.environ, **env})
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        m = re.search(r"flow speed ([0-9.e+-]+)fps", r.stdout)
        n_files = sum(len(fs) for _, _, fs in os.walk(out)) if r.returncode == 0 else 0
        print(f"{algo:5s} {tag:21s}: rc={r.returncode} wall {dt:6.2f}s  files {n_files}  summary flow speed "
              f"{m.group(1) if m else '?'} fps  ({NCLIPS * (NF - 1) / dt:6.1f} pairs/s incl. process start-up; "
              f"{cpu_s / (NCLIPS * (NF - 1)) * 1e3:6.2f} CPU-ms per pair)", flush=True)
        shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(tmp, ignore_errors=True)
