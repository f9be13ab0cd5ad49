"""GPU parity tests for the -st=png scheme on the device (second half): dfx_flow_to_png_device /
dfx_calc_batch_png* against the reference's convertFlowToPngImage (/root/reference/src/common.cpp:18-46).
Integer output and double bounds: bit-exact.

The golden images were produced by the reference's own source lines (tests/golden/make_png_planes_golden.py);
the oracle (oracle/quant_oracle.c: orc_flow_to_png_planes) is the restatement that travels to the GPU box."""
import os

import numpy as np
import pytest
import torch

from denseflow_amd.synth import SynthClip
from tests.golden.make_png_planes_golden import CASES, flow_with_extrema

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "png_planes_golden.npz")


def _png_on_device(dfx, flows: np.ndarray, img_pitch=None, img_stride=None):
    """flows: (n, H, W, 2) float32 on the host -> (x, y) uint8 (n, H, W), bounds (n, 2) through dfx_flow_to_png_device."""
    n, h, w, _ = flows.shape
    pitch = img_pitch or w
    stride = img_stride or pitch * h
    dev = torch.device("cuda", 0)
    d_flow = torch.from_numpy(np.ascontiguousarray(flows)).to(dev)
    d_x = torch.full((n * stride,), 7, dtype=torch.uint8, device=dev)
    d_y = torch.full((n * stride,), 9, dtype=torch.uint8, device=dev)
    d_b = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    with dfx.FlowEngine(w, h, "farn") as eng:
        eng.flow_to_png_device(d_flow.data_ptr(), h * w * 2, n, d_x.data_ptr(), d_y.data_ptr(), pitch, stride, d_b.data_ptr())
    torch.cuda.synchronize()
    X = d_x.cpu().numpy().reshape(n, stride)
    Y = d_y.cpu().numpy().reshape(n, stride)
    x = np.stack([X[i, : pitch * h].reshape(h, pitch)[:, :w] for i in range(n)])
    y = np.stack([Y[i, : pitch * h].reshape(h, pitch)[:, :w] for i in range(n)])
    pad = np.stack([X[i, : pitch * h].reshape(h, pitch)[:, w:] for i in range(n)])
    assert np.all(pad == 7), "bytes between rows were overwritten"
    return x, y, d_b.cpu().numpy()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_golden_images(dfx, case):
    name = case[0]
    g = np.load(GOLDEN)
    want = g[name + "_bgr"]
    x, y, b = _png_on_device(dfx, g[name + "_flow"][None])
    assert np.array_equal(dfx.FlowEngine.png_bgr(x[0], y[0], b[0, 0], b[0, 1]), want)


@pytest.mark.parametrize("w,h", [(64, 48), (61, 37), (130, 33), (257, 5), (3, 2), (1, 1), (1920, 1080)])
def test_matches_oracle_on_flows_built_for_the_bound_rule(dfx, oracle, w, h):
    rng = np.random.default_rng(w * 131 + h)
    n = 5 if w * h < 10 ** 6 else 2
    flows = np.stack([flow_with_extrema(w, h, float(rng.choice([0.0, 0.3, 7.9, 15.5, 31.7, 300.0, 1500.0])),
                                        float(rng.choice([0.0, 1.0, 3.9, 23.6, 64.0])), 50 + i, bool(i & 1)) for i in range(n)])
    x, y, b = _png_on_device(dfx, flows)
    for i in range(n):
        ox, oy, ob, _ = oracle.flow_to_png_planes(flows[i])
        assert tuple(b[i]) == ob, (w, h, i)
        assert np.array_equal(x[i], ox) and np.array_equal(y[i], oy), (w, h, i)


def test_nan_rules(dfx, oracle):
    """NaNs never win minMaxLoc wherever they stand; a plane of NaNs only locates nothing: 0 / 0, bound 4 (the
    device used to report a NaN bound there)."""
    from tests.test_oracle_quant import _nan_flows

    flows = _nan_flows()
    x, y, b = _png_on_device(dfx, np.stack(list(flows.values())))
    for i, (name, f) in enumerate(flows.items()):
        ox, oy, ob, _ = oracle.flow_to_png_planes(f)
        assert tuple(b[i]) == ob and np.isfinite(b[i]).all(), (name, tuple(b[i]), ob)
        assert np.array_equal(x[i], ox) and np.array_equal(y[i], oy), name


def test_padded_rows_and_strided_planes(dfx, oracle):
    w, h = 50, 21
    flows = np.stack([flow_with_extrema(w, h, 6.0 + i, 2.5, 70 + i) for i in range(2)])
    x, y, b = _png_on_device(dfx, flows, img_pitch=67, img_stride=67 * h + 13)
    for i in range(2):
        ox, oy, ob, _ = oracle.flow_to_png_planes(flows[i])
        assert np.array_equal(x[i], ox) and np.array_equal(y[i], oy) and tuple(b[i]) == ob


@pytest.mark.parametrize("algo", ["tvl1", "farn", "brox"])
@pytest.mark.parametrize("step,submit", [(1, False), (-2, True)])
def test_flowbuffer_to_png_planes(dfx, oracle, algo, step, submit):
    """dfx_calc_batch_png / dfx_submit_batch_png: the flows of dfx_calc_batch through the scheme, in ragged batches; the
    bounds array is complete when the call returns."""
    w, h, n = 96, 64, 8
    frames = SynthClip(w, h, 11).frames(n)
    with dfx.FlowEngine(w, h, algo, max_batch=3) as eng:
        flows = eng.calc_optflows(frames, step)
        x, y, b = eng.calc_optflows_png(frames, step, submit=submit)
    assert len(x) == len(flows) == n - abs(step) and b.shape == (len(flows), 2)
    for i, f in enumerate(flows):
        ox, oy, ob, bgr = oracle.flow_to_png_planes(f)
        assert tuple(b[i]) == ob, i
        assert np.array_equal(x[i], ox) and np.array_equal(y[i], oy), i
        assert np.array_equal(dfx.FlowEngine.png_bgr(x[i], y[i], *b[i]), bgr)


def test_device_resident_png_planes_at_1080p(dfx, oracle):
    w, h, n = 1920, 1080, 4
    dev = torch.device("cuda", 0)
    d_frames = SynthClip(w, h, 2).frames_torch(n, dev)
    d_flows = torch.empty((n - 1, h, w, 2), dtype=torch.float32, device=dev)
    d_x = torch.empty((n - 1, h, w), dtype=torch.uint8, device=dev)
    d_y = torch.empty_like(d_x)
    d_b = torch.zeros((n - 1, 2), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with dfx.FlowEngine(w, h, "farn", max_batch=2) as eng:
        eng.calc_optflows_device(d_frames.data_ptr(), w, w * h, n, 1, d_flows.data_ptr(), w * h * 2)
        eng.calc_optflows_png_device(d_frames.data_ptr(), w, w * h, n, 1, d_x.data_ptr(), d_y.data_ptr(), w, w * h,
                                     d_b.data_ptr())
    torch.cuda.synchronize()
    for i in range(n - 1):
        ox, oy, ob, _ = oracle.flow_to_png_planes(d_flows[i].cpu().numpy())
        assert tuple(d_b[i].cpu().numpy()) == ob
        assert np.array_equal(d_x[i].cpu().numpy(), ox) and np.array_equal(d_y[i].cpu().numpy(), oy)


def test_argument_checks(dfx):
    import ctypes as C

    L = dfx.load_library()
    with dfx.FlowEngine(64, 48, "farn") as eng:
        frames = SynthClip(64, 48, 1).frames(3)
        fp = (C.c_void_p * 3)(*[f.ctypes.data for f in frames])
        assert L.dfx_calc_batch_png(eng._h, fp, 64, 3, 1, None, None, 64, None) != 0  # NULL outputs with M > 0
        assert L.dfx_calc_batch_png(eng._h, fp, 64, 1, 1, None, None, 64, None) == 0  # M = 0: nothing to write
        assert L.dfx_submit_batch_png(eng._h, fp, 64, 3, 1, None, None, 64, None, None) != 0  # NULL ticket
