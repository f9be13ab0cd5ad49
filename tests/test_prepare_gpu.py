"""GPU parity tests for the frame preparation: dfx_prepare_frames* and dfx_set_source_format
against the oracle's cvtColor(BGR2GRAY) + cv::resize restatement.  Integer output: bit-exact."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_prepare_golden import CASES
from tests.test_oracle_prepare import SIZES

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "prepare_golden.npz")


def _frames(n, sw, sh, ch, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:sh, 0:sw]
    out = []
    for i in range(n):
        base = 128 + 80 * np.sin((xx + 1.7 * i) / 6.0) * np.cos((yy - 0.9 * i) / 5.0)
        img = np.stack([np.clip(base + rng.normal(0, 12, (sh, sw)) + 15 * k, 0, 255) for k in range(ch)], -1).astype(np.uint8)
        out.append(img[..., 0].copy() if ch == 1 else img)
    return out


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_golden_vectors(dfx, case):
    name, sw, sh, ch, dw, dh = case
    g = np.load(GOLDEN)
    with dfx.FlowEngine(dw, dh, "farn") as eng:
        out = eng.prepare_frames([g[name + "_src"]])
    assert np.array_equal(out[0], g[name + "_dst"])


@pytest.mark.parametrize("sw,sh,dw,dh", SIZES + [(1920, 1080, 455, 256), (1920, 1080, 960, 540), (454, 256, 1920, 1080)])
@pytest.mark.parametrize("ch", [1, 3])
def test_matches_oracle(dfx, oracle, sw, sh, dw, dh, ch):
    rng = np.random.default_rng(sw + 3 * dh + ch)
    frames = [rng.integers(0, 256, (sh, sw) if ch == 1 else (sh, sw, 3), dtype=np.uint8) for _ in range(2)]
    frames[1][...] = _frames(1, sw, sh, ch, 5)[0]
    with dfx.FlowEngine(dw, dh, "farn") as eng:
        out = eng.prepare_frames(frames)
    for f, o in zip(frames, out):
        assert np.array_equal(o, oracle.prepare_frame(f, dw, dh)), (sw, sh, dw, dh, ch)


def test_device_resident_and_padded(dfx, oracle):
    sw, sh, ch, dw, dh, n = 97, 61, 3, 40, 30, 3
    frames = _frames(n, sw, sh, ch, 2)
    src_pitch, src_stride = sw * ch + 5, (sw * ch + 5) * sh + 11
    buf = np.zeros(n * src_stride, np.uint8)
    for i, f in enumerate(frames):
        view = buf[i * src_stride: i * src_stride + src_pitch * sh].reshape(sh, src_pitch)
        view[:, : sw * ch] = f.reshape(sh, sw * ch)
    dev = torch.device("cuda", 0)
    d_src = torch.from_numpy(buf).to(dev)
    gp, gs = dw + 3, (dw + 3) * dh + 7
    d_out = torch.full((n * gs,), 9, dtype=torch.uint8, device=dev)
    with dfx.FlowEngine(dw, dh, "farn") as eng:
        eng.prepare_frames_device(d_src.data_ptr(), src_pitch, src_stride, sw, sh, ch, n, d_out.data_ptr(), gp, gs)
    out = d_out.cpu().numpy()
    for i, f in enumerate(frames):
        got = out[i * gs: i * gs + gp * dh].reshape(dh, gp)
        assert np.array_equal(got[:, :dw], oracle.prepare_frame(f, dw, dh))
        assert np.all(got[:, dw:] == 9)


@pytest.mark.parametrize("algo", ["tvl1", "farn"])
@pytest.mark.parametrize("ch,src", [(1, (150, 100)), (3, (192, 144)), (1, (96, 72))])
def test_source_format_feeds_the_flow_path(dfx, oracle, algo, ch, src):
    """Flows from source-format frames == flows from frames prepared by the oracle (the reference's loader)."""
    sw, sh = src
    dw, dh, n = 96, 72, 7
    frames = _frames(n, sw, sh, ch, 11)
    prepared = [oracle.prepare_frame(f, dw, dh) for f in frames]
    with dfx.FlowEngine(dw, dh, algo, max_batch=3) as eng:
        ref = eng.calc_optflows(prepared, 2)
        eng.set_source_format(sw, sh, ch)
        got = eng.calc_optflows(frames, 2)
        gx, gy = eng.calc_optflows_u8(frames, 2, 20)
        one = eng.calc(frames[0], frames[2])
        eng.set_source_format()
        again = eng.calc_optflows(prepared, 2)
    assert len(got) == n - 2
    for a, b, c in zip(ref, got, again):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert np.array_equal(one, ref[0])
    ox, oy = oracle.flow_to_u8(ref[1], -20, 20)
    assert np.array_equal(gx[1], ox) and np.array_equal(gy[1], oy)


def test_source_format_device_resident(dfx, oracle):
    sw, sh, ch, dw, dh, n = 128, 96, 1, 64, 48, 5  # exact 2x: the INTER_AREA switch
    frames = _frames(n, sw, sh, ch, 4)
    prepared = [oracle.prepare_frame(f, dw, dh) for f in frames]
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    d_flows = torch.empty((n - 1, dh, dw, 2), dtype=torch.float32, device=dev)
    with dfx.FlowEngine(dw, dh, "farn", max_batch=2) as eng:
        ref = eng.calc_optflows(prepared, 1)
        eng.set_source_format(sw, sh, ch)
        eng.calc_optflows_device(d_frames.data_ptr(), sw, sw * sh, n, 1, d_flows.data_ptr(), dw * dh * 2)
    got = d_flows.cpu().numpy()
    for i in range(n - 1):
        assert np.array_equal(got[i], ref[i])


def test_argument_errors(dfx):
    with dfx.FlowEngine(32, 24, "farn") as eng:
        with pytest.raises(dfx.DfxError):
            eng.set_source_format(64, 48, 2)
        with pytest.raises(dfx.DfxError):
            eng.set_source_format(-1, 48, 1)
        eng.set_source_format(64, 48, 3)
        with pytest.raises(ValueError):  # frames must now be 48 x 64 x 3
            eng.calc(np.zeros((24, 32), np.uint8), np.zeros((24, 32), np.uint8))
        eng.set_source_format(32, 24, 1)  # the engine's own format: back to the default
        eng.calc(np.zeros((24, 32), np.uint8), np.zeros((24, 32), np.uint8))
        assert eng.prepare_frames([]) == []
