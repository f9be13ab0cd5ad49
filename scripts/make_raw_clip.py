#!/usr/bin/env python3
"""Write N frames of the bench's synthetic clip (denseflow_amd.synth.SynthClip) as a raw u8 file for
tools/dfx_prof (the torch-free process rocprofv3 profiles).   make_raw_clip.py W H seed N out.raw [clips]
With clips > 1: `clips` clips of N frames each, seeds seed, seed + 1, ... back to back (the videolist shape).
Uses the GPU generator the bench uses when a GPU is visible (a 1080p frame takes ~1 s in numpy)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from denseflow_amd.synth import SynthClip  # noqa: E402

w, h, seed, n, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
clips = int(sys.argv[6]) if len(sys.argv) > 6 else 1
try:
    import torch

    dev = torch.device("cuda", 0) if torch.cuda.is_available() else None
except ImportError:
    dev = None
with open(out, "wb") as f:
    for c in range(clips):
        clip = SynthClip(w, h, seed + c)
        if dev is not None:
            f.write(clip.frames_torch(n, dev).cpu().numpy().tobytes())
        else:
            for t in range(n):
                f.write(clip.frame(t).tobytes())
