"""The batch geometry bench.py times, held against the oracle (VERDICT r4 next #1b).

The other 1080p parity tests run one pair or `max_batch=2`; the engines size their launches from the batch (Farneback's
row segments: 216 rows x 5 at 129 pairs, 54 x 20 at 2; TVL1's grid.z and step groups; Brox's patch schedule), so the
automatic batch of a 1080p FlowBuffer — 129 pairs (256 Mpx of frames) — is a configuration of its own.
One FlowBuffer of batch + 1 frames through the device-resident entry point (the one bench.py times): the first, a middle
and the LAST flow of the batch against the oracle, bit for bit, plus TVL1's executed iteration table of the last pair.

And the two workloads bench.py added in round 5 for content that does not converge at once (VERDICT r4 missing #4):
`tvl1_epsilon = 0` (300 x 5 x 5 inner iterations) and denseflow_amd.synth.HardClip, at sizes the oracle finishes in
seconds.  Reference call sites: /root/reference/src/denseflow_gpu.cpp:299-303 (create), :307-342 (the FlowBuffer loop)."""
import numpy as np
import pytest

from denseflow_amd.synth import HardClip, SynthClip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo,frames", [("tvl1", 130), ("farn", 130), ("brox", 130)])
def test_full_automatic_batch_at_1080p_first_middle_last_flow(dfx, oracle, algo, frames):
    import torch

    w, h = 1920, 1080
    clip = SynthClip(w, h, 2)
    d_frames = clip.frames_torch(frames, torch.device("cuda", 0))  # what bench.py feeds the engine (GPU-minted clip)
    n = frames - 1
    d_flows = torch.empty((n, h, w, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with dfx.FlowEngine(w, h, algo) as eng:
        eng.calc_optflows_device(d_frames.data_ptr(), w, w * h, frames, 1, d_flows.data_ptr(), w * h * 2)
        st = eng.stats()
    assert st.batch == n, f"the automatic batch of a 1080p {algo} handle is expected to be {n}, got {st.batch}"
    calc = {"tvl1": oracle.tvl1_calc, "farn": oracle.farneback_calc, "brox": oracle.brox_calc}[algo]
    for i in (0, n // 2, n - 1):
        f0, f1 = d_frames[i].cpu().numpy(), d_frames[i + 1].cpu().numpy()
        if algo == "tvl1":
            ref, tr = calc(f0, f1, want_trace=True)
            if i == n - 1:  # dfx_stats holds the last pair processed
                assert [r[:5] for r in st.iters_table()] == [r[:5] for r in tr.iters_table()][:st.levels]
        else:
            ref = calc(f0, f1)
        got = d_flows[i].cpu().numpy()
        assert np.array_equal(got, ref), f"{algo} flow {i} of {n}: max-abs {np.max(np.abs(got - ref))}"


@pytest.mark.parametrize("w,h,seed", [(224, 224, 2), (97, 61, 9)])
def test_no_early_exit_runs_every_iteration_and_matches_the_oracle(dfx, oracle, w, h, seed):
    """tvl1_epsilon = 0: the convergence test `error <= epsilon^2 * area` can only pass on an exactly stationary
    iteration, so every warp runs its 300 iterations (bench.py's tvl1_1080p_noexit leg; BASELINE.md section 3's
    no-early-exit ceilings are priced on exactly this schedule)."""
    clip = SynthClip(w, h, seed)
    f0, f1 = clip.frame(0), clip.frame(1)
    p = oracle.tvl1_default_params()
    p.epsilon = 0.0
    ref, tr = oracle.tvl1_calc(f0, f1, p, want_trace=True)
    with dfx.FlowEngine(w, h, "tvl1", tvl1_epsilon=0.0) as eng:
        out = eng.calc(f0, f1)
        st = eng.stats()
    table = [r[:5] for r in st.iters_table()]
    assert table == [r[:5] for r in tr.iters_table()][:st.levels]
    assert all(v == 300 for row in table for v in row), table
    assert st.tvl1_total_iters == 300 * 5 * st.levels
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("w,h,pairs", [(224, 224, 3), (640, 360, 1)])
def test_hard_clip_keeps_the_fine_levels_iterating_and_matches_the_oracle(dfx, oracle, w, h, pairs):
    """Two independently moving layers + 2 % noise: levels 0-2 run hundreds of iterations in their first warp and tens in
    the later ones (the plain clip: 16-40 and 2), and flows and iteration tables are still the oracle's bit for bit."""
    frames = HardClip(w, h, 2).frames(pairs + 1)
    with dfx.FlowEngine(w, h, "tvl1", max_batch=2) as eng:
        flows = eng.calc_optflows(frames, 1)
        st = eng.stats()
    plain_total = None
    for i in range(pairs):
        ref, tr = oracle.tvl1_calc(frames[i], frames[i + 1], want_trace=True)
        assert np.array_equal(flows[i], ref), i
        if i == pairs - 1:
            table = [r[:5] for r in tr.iters_table()][:tr.nscales]
            assert [r[:5] for r in st.iters_table()] == table
            assert all(sum(row) >= 150 for row in table[:3]), table  # the fine levels keep iterating
            plain = SynthClip(w, h, 2).frames(2)
            _, tp = oracle.tvl1_calc(plain[0], plain[1], want_trace=True)
            plain_total = sum(sum(r[:5]) for r in tp.iters_table()[:tp.nscales])
            assert sum(sum(r) for r in table) > 2 * plain_total
