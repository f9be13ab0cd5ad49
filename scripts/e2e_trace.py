#!/usr/bin/env python3
"""Stage trace of one host-shell run (DF_TRACE=1) on a synthetic clip: where does the wall time go?
Usage: python scripts/e2e_trace.py [W H NF ALGO]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, NF = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1920, 1080, 257)))
algo = sys.argv[4] if len(sys.argv) > 4 else "farn"
NCLIPS = int(sys.argv[5]) if len(sys.argv) > 5 else 1
tmp = tempfile.mkdtemp(prefix="dftrace_")
clips = []
for c in range(NCLIPS):
    clip = os.path.join(tmp, f"clip{c:03d}.y4m")
    frames = SynthClip(W, H, 2 + c).frames_torch(NF, torch.device("cuda", 0)).cpu().numpy()
    with open(clip, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F30:1 Ip A1:1 Cmono\n".encode())
        for fr in frames:
            f.write(b"FRAME\n")
            f.write(fr.tobytes())
    clips.append(clip)
open(os.path.join(tmp, "list.txt"), "w").write("".join(c + "\n" for c in clips))
r = subprocess.run([os.path.join(ROOT, "build", "denseflow"), os.path.join(tmp, "list.txt"), "-o=" + os.path.join(tmp, "out"),
                    "-a=" + algo, "-s=1", "-b=20"], capture_output=True, text=True, env={**os.environ, "DF_TRACE": "1"})
print(r.stdout[-300:])
print(r.stderr[-6000:])
