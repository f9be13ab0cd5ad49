#!/usr/bin/env python3
"""Within-process A/B sweep of engine knobs (fuse_k, max_batch, impl) on one resident clip.
Usage: python scripts/sweep_tvl1.py [W H NF] ; prints one line per configuration."""
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
    th = parts[3] if len(parts) > 3 else 0
    if len(parts) > 4:  # step-kernel tile geometry (tvl1) — read by dfx_create
        os.environ["DFX_TVL1_GEOM"] = str(parts[4])
    if len(parts) > 5:  # fused-SOR barrier scheme (brox) — read per launch
        os.environ["DFX_BROX_SOR"] = str(parts[5])
    if len(parts) > 6:  # zero-weight pyramid taps skipped (farn) — read per launch
        os.environ["DFX_FARN_SKIP0"] = str(parts[6])
    if len(parts) > 7:  # rows per workgroup of the polynomial expansion (farn) — read per launch
        os.environ["DFX_FARN_POLYROWS"] = str(parts[7])
    extra = {"tvl1_nscales": int(os.environ["NSCALES"])} if os.environ.get("NSCALES") else {}
    if os.environ.get("ITERS"):
        extra["tvl1_iterations"] = int(os.environ["ITERS"])
    if os.environ.get("EPS"):
        extra["tvl1_epsilon"] = float(os.environ["EPS"])
    eng = denseflow_amd.FlowEngine(W, H, ALGO, impl=impl, tvl1_fuse_k=k, max_batch=b, tvl1_tile_h=th, **extra)
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
    print(f"impl={impl} K={k} B={b} TH={th} cfg={cfg}: {reps*(NF-1)/dt:8.1f} pairs/s  dev_ms/pair={st.device_ms/st.pairs:7.3f} step_ms/pair={st.step_ms/st.pairs:7.3f} "
          f"launches/pair={st.kernel_launches/st.pairs:7.1f} noop={st.noop_steps/max(st.step_launches,1):.3f} "
          f"alg_GB/s(step)={st.step_algorithmic_bytes/(st.step_ms*1e-3)/1e9:8.1f}  [{same}]", flush=True)
    if os.environ.get("SWEEP_LEVELS"):
        print("    per-level step ms/pair:", " ".join(f"L{l}={st.level_ms[l]/st.pairs:.3f}({st.level_launches[l]/ (st.pairs/ max(1,(b or 1))) :.0f} launches/batch, {st.level_ms[l]*1e3/max(st.level_launches[l],1):.1f}us/launch)" for l in range(st.levels)), flush=True)
    eng.close()
