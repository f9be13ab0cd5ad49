"""Property tests (hypothesis) for the two integer-output oracles against their NumPy restatements: flow bounding
(reference src/common.cpp:4-16) and frame preparation (cvtColor + cv::resize as load_frames_batch calls them)."""
import numpy as np
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as hnp

from tests import numpy_restatement as NR

finite_or_special = st.one_of(
    st.floats(width=32, allow_nan=True, allow_infinity=True),
    st.floats(min_value=-64, max_value=64, width=32),
)


@settings(max_examples=60, deadline=None)
@given(flow=hnp.arrays(np.float32, st.tuples(st.integers(1, 9), st.integers(1, 17), st.just(2)), elements=finite_or_special),
       bound=st.sampled_from([1.0, 2.0, 15.0, 20.0, 32.0, 127.5]), shift=st.sampled_from([0.0, 3.0, -7.5]))
def test_flow_bounding_oracle_equals_numpy(oracle, flow, bound, shift):
    lo, hi = -bound + shift, bound + shift
    ox, oy = oracle.flow_to_u8(flow, lo, hi)
    nx, ny = NR.flow_to_u8(flow, lo, hi)
    assert np.array_equal(ox, nx) and np.array_equal(oy, ny)
    v = flow.astype(np.float64)
    inside = (v >= lo) & (v <= hi)
    q = np.stack([ox, oy], -1).astype(np.float64)
    # inside the interval the 8-bit value is within half a step of the exact affine map
    assert np.all(np.abs(q[inside] - 255 * (v[inside] - lo) / (hi - lo)) <= 0.5 + 1e-9)
    assert np.all(q[v > hi] == 255) and np.all(q[v < lo] == 0)


@settings(max_examples=60, deadline=None)
@given(data=st.data(), sw=st.integers(1, 40), sh=st.integers(1, 30), dw=st.integers(1, 50), dh=st.integers(1, 40),
       ch=st.sampled_from([1, 3]))
def test_frame_preparation_oracle_equals_numpy(oracle, data, sw, sh, dw, dh, ch):
    shape = (sh, sw) if ch == 1 else (sh, sw, 3)
    src = data.draw(hnp.arrays(np.uint8, shape))
    out = oracle.prepare_frame(src, dw, dh)
    assert out.shape == (dh, dw)
    assert np.array_equal(out, NR.prepare_frame(src, dw, dh))
    gray = NR.bgr2gray(src) if ch == 3 else src
    # bilinear interpolation (and the 2x2 mean) never leaves the range of its inputs by more than the rounding
    assert int(out.min()) >= int(gray.min()) - 1 and int(out.max()) <= int(gray.max()) + 1
