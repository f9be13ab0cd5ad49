"""GPU parity tests for the flow bounding: dfx_flow_to_u8_device / dfx_calc_batch_u8*
against the reference's convertFlowToImage (src/common.cpp:4-16).  Integer output: bit-exact.

The golden vectors were produced by the reference's own source lines (tests/golden/make_quant_golden.py);
the oracle (oracle/quant_oracle.c) is the restatement that travels to the GPU box.
"""
import os

import numpy as np
import pytest
import torch

from denseflow_amd.synth import SynthClip
from tests.golden.make_quant_golden import CASES, adversarial_flow

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "quant_golden.npz")


def _bound_on_device(dfx, flows: np.ndarray, lo: float, hi: float, img_pitch=None, img_stride=None):
    """flows: (n, H, W, 2) float32 on the host -> (x, y) uint8 (n, H, W) through dfx_flow_to_u8_device."""
    n, h, w, _ = flows.shape
    pitch = img_pitch or w
    stride = img_stride or pitch * h
    dev = torch.device("cuda", 0)
    d_flow = torch.from_numpy(np.ascontiguousarray(flows)).to(dev)
    d_x = torch.full((n * stride,), 7, dtype=torch.uint8, device=dev)
    d_y = torch.full((n * stride,), 9, dtype=torch.uint8, device=dev)
    with dfx.FlowEngine(w, h, "farn") as eng:
        eng.flow_to_u8_device(d_flow.data_ptr(), h * w * 2, n, lo, hi, d_x.data_ptr(), d_y.data_ptr(), pitch, stride)
    torch.cuda.synchronize()
    X = d_x.cpu().numpy().reshape(n, stride)
    Y = d_y.cpu().numpy().reshape(n, stride)
    x = np.stack([X[i, : pitch * h].reshape(h, pitch)[:, :w] for i in range(n)])
    y = np.stack([Y[i, : pitch * h].reshape(h, pitch)[:, :w] for i in range(n)])
    pad_x = np.stack([X[i, : pitch * h].reshape(h, pitch)[:, w:] for i in range(n)])
    assert np.all(pad_x == 7), "bytes between rows were overwritten"
    return x, y


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_golden_vectors(dfx, case):
    name = case[0]
    g = np.load(GOLDEN)
    bound = float(g[name + "_bound"][0])
    x, y = _bound_on_device(dfx, g[name + "_flow"][None], -bound, bound)
    assert np.array_equal(x[0], g[name + "_x"]) and np.array_equal(y[0], g[name + "_y"])


@pytest.mark.parametrize("w,h", [(64, 48), (61, 37), (130, 33), (257, 5), (3, 2), (1, 1), (1920, 1080)])
@pytest.mark.parametrize("lo,hi", [(-20.0, 20.0), (-5.0, 32.0)])
def test_matches_oracle_on_adversarial_flows(dfx, oracle, w, h, lo, hi):
    n = 3 if w * h < 10 ** 6 else 1
    flows = np.stack([adversarial_flow(hi, w, h, 50 + i) if w * h * 2 >= 1806 else
                      (np.random.default_rng(i).standard_normal((h, w, 2)) * hi).astype(np.float32) for i in range(n)])
    x, y = _bound_on_device(dfx, flows, lo, hi)
    for i in range(n):
        ox, oy = oracle.flow_to_u8(flows[i], lo, hi)
        assert np.array_equal(x[i], ox) and np.array_equal(y[i], oy), (w, h, i)


def test_degenerate_bounds_follow_the_formula(dfx, oracle):
    flows = adversarial_flow(1.0, 64, 64, 5)[None]
    for lo, hi in [(0.0, 0.0), (3.0, -3.0), (-1e-30, 1e-30)]:
        x, y = _bound_on_device(dfx, flows, lo, hi)
        ox, oy = oracle.flow_to_u8(flows[0], lo, hi)
        assert np.array_equal(x[0], ox) and np.array_equal(y[0], oy), (lo, hi)


def test_padded_rows_and_strided_planes(dfx, oracle):
    w, h = 50, 21
    flows = np.stack([adversarial_flow(20.0, w, h, 70 + i) for i in range(2)])
    x, y = _bound_on_device(dfx, flows, -20.0, 20.0, img_pitch=67, img_stride=67 * h + 13)
    for i in range(2):
        ox, oy = oracle.flow_to_u8(flows[i], -20.0, 20.0)
        assert np.array_equal(x[i], ox) and np.array_equal(y[i], oy)


@pytest.mark.parametrize("algo", ["tvl1", "farn", "brox"])
@pytest.mark.parametrize("step", [1, -2])
def test_calc_batch_u8_is_bounding_of_calc_batch(dfx, oracle, algo, step):
    """encodeFlowMap(calc(a, b), bound) with both halves on the device == the oracle's bounding of the flows."""
    w, h, n, bound = 96, 72, 8, 20
    frames = SynthClip(w, h, 12).frames(n)
    with dfx.FlowEngine(w, h, algo, max_batch=3) as eng:  # 3 batches through the copy pipeline
        flows = eng.calc_optflows(frames, step)
        img_x, img_y = eng.calc_optflows_u8(frames, step, bound)
    assert len(img_x) == len(img_y) == len(flows) == n - abs(step)
    for i, f in enumerate(flows):
        ox, oy = oracle.flow_to_u8(f, -bound, bound)
        assert np.array_equal(img_x[i], ox) and np.array_equal(img_y[i], oy), (algo, step, i)
    # small synthetic motion: the planes are far from saturated, i.e. the test compares something
    assert np.mean((img_x[0] > 0) & (img_x[0] < 255)) > 0.9


def test_calc_batch_u8_matches_oracle_end_to_end(dfx, oracle):
    w, h, bound = 128, 96, 2  # a tight bound so that clamping to 0 / 255 happens as well
    clip = SynthClip(w, h, 3)
    frames = clip.frames(4)
    refs = [oracle.flow_to_u8(oracle.tvl1_calc(frames[i], frames[i + 1]), -bound, bound) for i in range(3)]
    with dfx.FlowEngine(w, h, "tvl1") as eng:
        img_x, img_y = eng.calc_optflows_u8(frames, 1, bound)
    for i in range(3):
        assert np.array_equal(img_x[i], refs[i][0]) and np.array_equal(img_y[i], refs[i][1])
    allv = np.concatenate([v.ravel() for v in img_x + img_y])
    assert (allv == 0).any() or (allv == 255).any()


def test_device_resident_u8_output(dfx, oracle):
    w, h, n, bound = 160, 90, 6, 20
    clip = SynthClip(w, h, 8)
    dev = torch.device("cuda", 0)
    d_frames = clip.frames_torch(n, dev)
    d_flows = torch.empty((n - 1, h, w, 2), dtype=torch.float32, device=dev)
    d_x = torch.zeros((n - 1, h, w), dtype=torch.uint8, device=dev)
    d_y = torch.zeros((n - 1, h, w), dtype=torch.uint8, device=dev)
    with dfx.FlowEngine(w, h, "farn", max_batch=4) as eng:
        eng.calc_optflows_device(d_frames.data_ptr(), w, w * h, n, 1, d_flows.data_ptr(), w * h * 2)
        eng.calc_optflows_u8_device(d_frames.data_ptr(), w, w * h, n, 1, -bound, bound, d_x.data_ptr(), d_y.data_ptr(),
                                    w, w * h)
    torch.cuda.synchronize()
    flows = d_flows.cpu().numpy()
    for i in range(n - 1):
        ox, oy = oracle.flow_to_u8(flows[i], -bound, bound)
        assert np.array_equal(d_x[i].cpu().numpy(), ox) and np.array_equal(d_y[i].cpu().numpy(), oy)


def test_argument_errors(dfx):
    with dfx.FlowEngine(32, 32, "farn") as eng:
        with pytest.raises(dfx.DfxError):
            eng.flow_to_u8_device(0, 32 * 32 * 2, 1, -1, 1, 0, 0, 32, 32 * 32)
        dev = torch.device("cuda", 0)
        t = torch.zeros(32 * 32 * 2, device=dev)
        o = torch.zeros(32 * 32, dtype=torch.uint8, device=dev)
        with pytest.raises(dfx.DfxError):  # pitch smaller than a row
            eng.flow_to_u8_device(t.data_ptr(), 32 * 32 * 2, 1, -1, 1, o.data_ptr(), o.data_ptr(), 16, 32 * 32)
        eng.flow_to_u8_device(t.data_ptr(), 32 * 32 * 2, 0, -1, 1, 0, 0, 32, 32 * 32)  # n = 0 is a no-op
