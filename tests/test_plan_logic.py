"""CPU test of the FlowBuffer plan (denseflow_amd/csrc/dfx_plan.h, the header dfx_api.cpp compiles): the pairs of a
FlowBuffer are the reference's (/root/reference/src/denseflow_gpu.cpp:307-316: M = max(N - |step|, 0); flow i is
(i, i + step) for step > 0, (i - step, i) otherwise) inside every clip and never across a clip boundary; the device batches
cover every pair once, bring every needed frame in exactly once and in order, and never need more frame slots than
dfx_frames_needed says (a frame lives in slot id % F)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def plan():
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libplan_harness.%d.so" % os.getpid())
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "plan_harness.cpp")],
                   check=True, capture_output=True)
    L = C.CDLL(so)
    os.unlink(so)
    return L


def _pairs(L, seg, step):
    s = (C.c_int * max(len(seg), 1))(*seg)
    cap = sum(seg) + 1
    a, b = (C.c_int * cap)(), (C.c_int * cap)()
    m = L.ph_pairs(s, len(seg), step, a, b, cap)
    return [(a[i], b[i]) for i in range(m)]


def _plan(L, seg, step, batch):
    s = (C.c_int * max(len(seg), 1))(*seg)
    cap = sum(seg) + 1
    out = (C.c_longlong * (4 * cap))()
    n = L.ph_plan(s, len(seg), step, batch, out, cap)
    return [tuple(out[4 * k:4 * k + 4]) for k in range(n)], L.ph_frames_needed(s, len(seg), step, batch)


def _reference_pairs(seg, step):
    """The reference's loop, clip by clip (src/denseflow_gpu.cpp:307-316)."""
    out, off = [], 0
    for n in seg:
        m = max(n - abs(step), 0)
        for i in range(m):
            out.append((off + i, off + i + step) if step > 0 else (off + i - step, off + i))
        off += n
    return out


@pytest.mark.parametrize("seg,step", [([7], 1), ([7], -1), ([7], 3), ([7], -3), ([3], 3), ([0], 1), ([5, 1, 7, 2, 4], 1),
                                      ([5, 1, 7, 2, 4], -2), ([300] * 16, 1), ([2, 2, 2], 2), ([], 1)])
def test_pairs_are_the_references_inside_every_clip(plan, seg, step):
    assert _pairs(plan, seg, step) == _reference_pairs(seg, step)


@settings(max_examples=300, deadline=None)
@given(seg=st.lists(st.integers(0, 40), min_size=1, max_size=12), step=st.integers(-5, 5).filter(lambda s: s != 0),
       batch=st.integers(1, 64))
def test_batches_cover_every_pair_and_every_needed_frame_once(plan, seg, step, batch):
    pairs = _reference_pairs(seg, step)
    assert _pairs(plan, seg, step) == pairs
    batches, f_need = _plan(plan, seg, step, batch)
    if not pairs:
        assert batches == [] and f_need == 0
        return
    # pairs: consecutive ranges of at most `batch`, all of them
    assert batches[0][0] == 0 and sum(b[1] for b in batches) == len(pairs)
    for k, (i0, nb, first_new, n_new) in enumerate(batches):
        assert 1 <= nb <= batch and (k == 0 or i0 == batches[k - 1][0] + batches[k - 1][1])
    # frames: prepared once, in increasing order; resident when their batch runs; slots never collide within a batch
    F = max(f_need, 1)
    prepared, slot_of = set(), {}
    last = -1
    for i0, nb, first_new, n_new in batches:
        for f in range(first_new, first_new + n_new):
            assert f not in prepared and f > last
            prepared.add(f)
            last = f
            slot_of[f % F] = f  # what the engines do: frame id f lives in slot f % F
        used = {f for p in pairs[i0:i0 + nb] for f in p}
        assert used <= prepared
        assert max(used) - min(used) + 1 <= f_need
        for f in used:
            assert slot_of[f % F] == f, "a frame this batch needs was evicted"
    # no frame is prepared that no pair of its own or a later batch could need... except the frames BETWEEN needed ones
    # (clips without a pair inside a batch's range): they are inside [first needed, last needed] of some batch
    assert prepared <= set(range(sum(seg)))


def test_one_clip_needs_batch_plus_step_frames(plan):
    for n, step, batch in [(300, 1, 129), (300, -2, 64), (34, 2, 32), (10, 1, 100)]:
        _, f_need = _plan(plan, [n], step, batch)
        assert f_need == min(batch, n - abs(step)) + abs(step)
    # sixteen 300-frame clips, 2048-pair batches: a batch spans up to eight clips and needs their boundary frames too
    _, f_need = _plan(plan, [300] * 16, 1, 2048)
    assert 2048 + 7 <= f_need <= 2048 + 8


@given(w=st.integers(1, 8192), h=st.integers(1, 8192), n_pairs=st.integers(1, 2048))
@settings(max_examples=300, deadline=None)
def test_farneback_stream_segments(plan, w, h, n_pairs):
    """Segments of the Farneback row-stream iteration kernel (denseflow_amd/csrc/farneback_plan.h): whole 6-row steps;
    ceil(h / rows) segments partition the rows with no empty segment; never shorter than 48 rows unless the level is (12
    warm-up rows per segment); and as many segments as it takes for a launch to be >= 16 generations of the machine's 1024
    workgroup slots, where the level's height allows."""
    rows = plan.ph_farn_seg_rows(w, h, n_pairs)
    assert rows > 0 and rows % 6 == 0
    nseg = -(-h // rows)
    assert (nseg - 1) * rows < h <= nseg * rows  # a partition, the last segment not empty
    assert rows >= min(48, -(-h // 6) * 6)
    cols = -(-w // 64)
    wgs = cols * n_pairs * nseg
    if wgs < 0.85 * 16 * 1024:  # (rounding a segment up to whole steps costs a few segments) fewer generations only where
        # the 48-row floor or the level's height stops the cutting
        assert rows >= h or rows < 2 * 48 + 6, (w, h, n_pairs, rows, nseg)
