#!/usr/bin/env python3
"""Per-stage wall time of one host-shell run (DF_STAGES=1) on a synthetic clip: which stage holds the pipeline?
Usage: python scripts/round5/e2e_stages.py W H NF ALGO [ST] [extra env K=V ...]"""
import os
import resource
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, NF = (int(v) for v in sys.argv[1:4])
algo = sys.argv[4]
st = sys.argv[5] if len(sys.argv) > 5 else "jpg"
env_extra = dict(kv.split("=", 1) for kv in sys.argv[6:])
tmp = tempfile.mkdtemp(prefix="dfstages_")
clip = os.path.join(tmp, "clip.y4m")
t0 = time.perf_counter()
frames = SynthClip(W, H, 2).frames_torch(NF, torch.device("cuda", 0)).cpu().numpy()
with open(clip, "wb") as f:
    f.write(f"YUV4MPEG2 W{W} H{H} F30:1 Ip A1:1 Cmono\n".encode())
    for fr in frames:
        f.write(b"FRAME\n")
        f.write(fr.tobytes())
print(f"clip written in {time.perf_counter() - t0:.1f} s", flush=True)
for rep in range(2):  # the second run reads the clip from the page cache for sure
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "build", "denseflow"), clip, "-o=" + os.path.join(tmp, f"out{rep}"), "-a=" + algo,
                        "-s=1", "-b=20", "-st=" + st], capture_output=True, text=True,
                       env={**os.environ, "DF_STAGES": "1", **env_extra})
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    print(f"run {rep}: rc {r.returncode}, wall {dt:.2f} s = {(NF - 1) / dt:.1f} flows/s incl. start-up, {cpu / (NF - 1) * 1e3:.2f} CPU-ms per pair "
          f"{env_extra}")
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "")
    print("\n".join(ln for ln in r.stderr.splitlines() if "stages" in ln))
    subprocess.run(["rm", "-rf", os.path.join(tmp, f"out{rep}")])
subprocess.run(["rm", "-rf", tmp])
