"""dfx_submit_batch / dfx_submit_batch_u8 / dfx_wait (include/dfx.h): a submitted FlowBuffer's last download and
hand-over run beside the next FlowBuffer's uploads and compute.  Results must be exactly those of the synchronous
entry points — for small frames (page-locked bounce buffers, hand-over on the helper thread) and for large frames
(direct downloads), for float flows and for bounded planes, with one or several batches per FlowBuffer."""
import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,algo,max_batch", [(96, 64, "tvl1", 0), (96, 64, "farn", 3), (640, 480, "farn", 0),
                                                (640, 480, "tvl1", 2)])
def test_two_flowbuffers_in_flight_give_the_synchronous_results(dfx, w, h, algo, max_batch):
    fa = SynthClip(w, h, 5).frames(7)
    fb = SynthClip(w, h, 9).frames(5)
    with dfx.FlowEngine(w, h, algo, max_batch=max_batch) as eng:
        ref_a, ref_b = eng.calc_optflows(fa, 1), eng.calc_optflows(fb, -2)
        ta, out_a = eng.submit_optflows(fa, 1)
        tb, out_b = eng.submit_optflows(fb, -2)  # issued while the tail of A is still in flight
        assert ta > 0 and tb > ta
        eng.wait(ta)
        for r, o in zip(ref_a, out_a):
            assert np.array_equal(r, o)
        eng.wait(tb)
        for r, o in zip(ref_b, out_b):
            assert np.array_equal(r, o)
        eng.wait(ta)  # waiting again is a no-op
        eng.wait(0)


@pytest.mark.parametrize("w,h", [(80, 56), (704, 480)])
def test_bounded_planes_and_mixing_with_synchronous_calls(dfx, w, h):
    frames = SynthClip(w, h, 21).frames(9)
    with dfx.FlowEngine(w, h, "farn", max_batch=4) as eng:  # 8 flows: two batches per FlowBuffer
        ref_x, ref_y = eng.calc_optflows_u8(frames, 1, 20)
        tickets = []
        for _ in range(3):  # three FlowBuffers back to back, collected afterwards in order
            tickets.append(eng.submit_optflows(frames, 1, bound=20))
        one = eng.calc(frames[0], frames[1])  # a synchronous call first completes everything outstanding
        for t, (ox, oy) in tickets:
            eng.wait(t)
            for i in range(len(ref_x)):
                assert np.array_equal(ref_x[i], ox[i]) and np.array_equal(ref_y[i], oy[i])
        assert np.array_equal(one, eng.calc_optflows(frames[:2], 1)[0])


def test_empty_flowbuffer_has_no_ticket(dfx):
    with dfx.FlowEngine(64, 48, "tvl1") as eng:
        t, flows = eng.submit_optflows(SynthClip(64, 48, 1).frames(1), 1)
        assert t == 0 and flows == []
        eng.wait(0)


def test_one_helper_thread_per_handle_across_many_batches_and_calls(dfx):
    """the host-side work beside the calling thread (hand-over from the bounce buffer, JPEG assembly, the
    gather of the next batch's small frames) runs on ONE persistent thread per handle (dfx_helper.h) instead of a
    std::thread per batch.  Small frames (bounce buffers in both directions), max_batch = 2 -> 6 batches per FlowBuffer,
    every output kind, repeated calls on one handle and two handles side by side: the results of the one-batch handle."""
    import threading

    w, h, n = 112, 80, 13
    frames = SynthClip(w, h, 33).frames(n)
    with dfx.FlowEngine(w, h, "farn") as one:
        ref_f = one.calc_optflows(frames, 1)
        ref_x, ref_y = one.calc_optflows_u8(frames, 1, 20)
        ref_jx, ref_jy = one.calc_optflows_jpeg(frames, 1, 20)
        ref_px, ref_py, ref_b = one.calc_optflows_png(frames, 1)

    def work(errors):
        try:
            with dfx.FlowEngine(w, h, "farn", max_batch=2) as eng:
                for _ in range(3):  # the same thread serves every call of the handle
                    f = eng.calc_optflows(frames, 1)
                    x, y = eng.calc_optflows_u8(frames, 1, 20)
                    jx, jy = eng.calc_optflows_jpeg(frames, 1, 20)
                    px, py, b = eng.calc_optflows_png(frames, 1, submit=True)
                    assert all(np.array_equal(a, r) for a, r in zip(f, ref_f))
                    assert all(np.array_equal(a, r) for a, r in zip(x, ref_x)) and all(np.array_equal(a, r) for a, r in zip(y, ref_y))
                    assert jx == ref_jx and jy == ref_jy
                    assert all(np.array_equal(a, r) for a, r in zip(px, ref_px)) and np.array_equal(b, ref_b)
                    assert all(np.array_equal(a, r) for a, r in zip(py, ref_py))
        except Exception as e:  # pragma: no cover
            errors.append(e)

    errors = []
    threads = [threading.Thread(target=work, args=(errors,)) for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
