#!/usr/bin/env python3
"""Within-process A/B sweep of engine knobs on one resident clip.
Usage: SWEEP="impl:K:B[:variant[:math]],..." ALGO=tvl1|farn|brox python scripts/sweep_tvl1.py [W H NF]
(variant = dfx_params.variant, DFX_VAR_* bits; math = dfx_params.tvl1_math); prints one line per configuration and
whether its flows are bit-identical to the first configuration's."""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import denseflow_amd  # noqa: E402
from denseflow_amd.synth import SynthClip  # noqa: E402

W, H, NF = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1920, 1080, 33)))
ALGO = os.environ.get("ALGO", "tvl1")
configs = os.environ.get("SWEEP", "0:4:0,0:4:4,0:2:4,0:6:4,0:8:4,0:4:8,0:4:1,1:1:2").split(",")
dev = torch.device("cuda", 0)
clip = SynthClip(W, H, 2)
d_frames = clip.frames_torch(NF, dev)
d_flows = torch.empty((NF - 1, H, W, 2), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ref = None
for cfg in configs:
    parts = [int(v) for v in cfg.split(":")]
    impl, k, b = parts[:3]
    variant = parts[3] if len(parts) > 3 else 0
    math = parts[4] if len(parts) > 4 else 0
    extra = {"tvl1_nscales": int(os.environ["NSCALES"])} if os.environ.get("NSCALES") else {}
    if os.environ.get("ITERS"):
        extra["tvl1_iterations"] = int(os.environ["ITERS"])
    if os.environ.get("EPS"):
        extra["tvl1_epsilon"] = float(os.environ["EPS"])
    eng = denseflow_amd.FlowEngine(W, H, ALGO, impl=impl, tvl1_fuse_k=k, max_batch=b, variant=variant, tvl1_math=math, **extra)
    run = lambda: eng.calc_optflows_device(d_frames.data_ptr(), W, W * H, NF, 1, d_flows.data_ptr(), W * H * 2)
    run()
    eng.reset_stats()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        run()
    dt = time.perf_counter() - t0
    st = eng.stats()
    out = d_flows.clone()
    same = "ref" if ref is None else ("bit-identical" if torch.equal(out, ref) else "DIFFERENT max|d|=%g" % float((out - ref).abs().max()))
    if ref is None:
        ref = out
    print(f"impl={impl} K={k} B={b} variant={variant} math={math}: {reps*(NF-1)/dt:8.1f} pairs/s  dev_ms/pair={st.device_ms/st.pairs:7.3f} step_ms/pair={st.step_ms/st.pairs:7.3f} "
          f"launches/pair={st.kernel_launches/st.pairs:7.1f} noop={st.noop_steps/max(st.step_launches,1):.3f} "
          f"alg_GB/s(step)={st.step_algorithmic_bytes/(st.step_ms*1e-3)/1e9:8.1f}  [{same}]", flush=True)
    if os.environ.get("SWEEP_LEVELS"):
        print("    per-level step ms/pair:", " ".join(f"L{l}={st.level_ms[l]/st.pairs:.3f}({st.level_launches[l]/ (st.pairs/ max(1,(b or 1))) :.0f} launches/batch, {st.level_ms[l]*1e3/max(st.level_launches[l],1):.1f}us/launch)" for l in range(st.levels)), flush=True)
    eng.close()
