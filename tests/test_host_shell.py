"""Host shell (C++): the reference's CLI / dense_flow.h surface on the MI355X build.  CPU tests cover the
command line (help text golden from the reference README, exit codes, error texts), quantisation, the
decoder-free codecs and frame extraction; gpu tests run the operator and the whole CLI on the device."""
import ctypes as C
import io
import os
import subprocess

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "denseflow")

HELP_GOLDEN = """GPU optical flow extraction.
Usage: denseflow [params] input

\t-a, --algorithm (value:tvl1)
\t\toptical flow algorithm (nv/tvl1/farn/brox)
\t-b, --bound (value:32)
\t\tmaximum of optical flow
\t--cf, --classFolder
\t\toutputDir/class/video/flow.jpg
\t-f, --force
\t\tregardless of the marked .done file
\t-g, --gpus (value:1)
\t\tnumber of GPUs to shard the input list over
\t-h, --help (value:true)
\t\tprint help message
\t--if, --inputFrames
\t\tinputs are frames
\t--newHeight, --nh (value:0)
\t\tnew height
\t--newShort, --ns (value:0)
\t\tshort side length
\t--newWidth, --nw (value:0)
\t\tnew width
\t-o, --outputDir (value:.)
\t\troot dir of output
\t-s, --step (value:0)
\t\tright - left (0 for img, non-0 for flow)
\t--saveType, --st (value:jpg)
\t\tsave format type (png/h5/jpg)
\t-v, --verbose
\t\tverbose

\tinput
\t\tfilename of video or folder of frames or a list.txt of those
"""


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", ROOT, "host"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return BIN


@pytest.fixture(scope="module")
def harness(built):
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhost_harness.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", so,
           os.path.join(ROOT, "tests", "host_harness.cpp"), os.path.join(ROOT, "build", "libzzdenseflow.a"),
           "-L" + os.path.join(ROOT, "denseflow_amd", "lib"), "-ldfx", "-lpthread", "-lz",
           "-Wl,-rpath," + os.path.join(ROOT, "denseflow_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return C.CDLL(so)


def write_y4m(path, frames):
    h, w = frames[0].shape
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 Cmono\n".encode())
        for fr in frames:
            f.write(b"FRAME\n")
            f.write(np.ascontiguousarray(fr, np.uint8).tobytes())


def test_help_text_matches_reference_readme(built):
    # README.md:109-143 of the reference, plus the one added key (-g)
    for args in ([], ["-h"], ["--help", "x.mp4"]):
        r = subprocess.run([built] + args, capture_output=True, text=True)
        assert r.returncode == 0
        assert r.stdout == HELP_GOLDEN


def test_cli_error_behaviour(built, tmp_path):
    clip = tmp_path / "a.y4m"
    write_y4m(clip, SynthClip(32, 24, 1).frames(2))
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path), "-s=1", "-a=lk"], capture_output=True, text=True)
    assert r.returncode == 1 and "lk not supported!" in r.stdout and "check init param error." in r.stdout
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path), "-s=1", "-b=0"], capture_output=True, text=True)
    assert r.returncode == 1 and "bound should > 0!" in r.stdout
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path), "-s=1", "-st=gif"], capture_output=True, text=True)
    assert "only jpg/png/h5 are supported (no gif) for output" in r.stdout
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path), "-s=1", "-nw=2000000000", "-nh=2000000000"],
                       capture_output=True, text=True)  # beyond the engine's limit: refused before anything is sized by it
    assert r.returncode == 1 and "height and width cannot > 32768!" in r.stdout
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path), "-s=-2147483648"], capture_output=True, text=True)
    assert r.returncode == 1 and "step out of range!" in r.stdout
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path), "-s=abc"], capture_output=True, text=True)
    assert r.returncode == 0 and "can not convert" in r.stdout  # parse errors: print, exit 0 (tools/denseflow.cpp:30-33)
    r = subprocess.run([built, str(tmp_path / "missing.y4m"), "-o=" + str(tmp_path), "-s=1"], capture_output=True,
                       text=True)
    assert r.returncode == 1 and "does not exist!" in r.stdout
    if not os.path.exists("/dev/kfd"):  # no GPU: the flow mode must fail loudly, never compute on the CPU
        r = subprocess.run([built, str(clip), "-o=" + str(tmp_path), "-s=1"], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stdout


def test_frame_extraction_mode_writes_decodable_jpegs(built, tmp_path):
    from PIL import Image

    frames = SynthClip(96, 64, 5).frames(3)
    clip = tmp_path / "vid.y4m"
    write_y4m(clip, frames)
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path / "out"), "-s=0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "1 videos (3 frames, 0 tvl1 flows) processed" in r.stdout
    assert "-s=0 in this build writes GRAY frames" in r.stdout  # the deviation from the reference is announced
    for i, fr in enumerate(frames):
        img = np.array(Image.open(tmp_path / "out" / "vid" / f"img_{i:05d}.jpg"))
        assert img.shape == fr.shape
        mse = np.mean((img.astype(np.float64) - fr) ** 2)
        assert 10 * np.log10(255 ** 2 / mse) > 38  # quality 95


def test_quantisation_cast_formula(harness):
    rng = np.random.default_rng(0)
    w, h, bound = 37, 11, 20
    fx = rng.uniform(-30, 30, (h, w)).astype(np.float32)
    fy = rng.uniform(-30, 30, (h, w)).astype(np.float32)
    fx[0, :4] = [-20, 20, 0, 20.0001]
    ox = np.zeros((h, w), np.uint8)
    oy = np.zeros((h, w), np.uint8)
    harness.hh_quantise(fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p), w, h, bound,
                        ox.ctypes.data_as(C.c_void_p), oy.ctypes.data_as(C.c_void_p))

    def cast(v):  # src/common.cpp:6 of the reference: double arithmetic, cvRound = round-half-even
        v = v.astype(np.float64)
        q = np.rint(255 * (v + bound) / (2 * bound))
        return np.where(v > bound, 255, np.where(v < -bound, 0, q)).astype(np.uint8)

    assert np.array_equal(ox, cast(fx)) and np.array_equal(oy, cast(fy))
    assert list(ox[0, :4]) == [0, 255, 128, 255]


@pytest.mark.parametrize("w,h", [(70, 45), (64, 64), (257, 131), (8, 8), (5, 3)])
def test_jpeg_transform_paths_agree_and_decode(harness, w, h):
    """The AVX2 and the portable forward DCT feed the same entropy coder; both must decode to the image."""
    from PIL import Image

    rng = np.random.default_rng(w * 1000 + h)
    yy, xx = np.mgrid[0:h, 0:w]
    gray = np.clip(128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
    buf = np.zeros(1 << 20, np.uint8)
    dec = {}
    for name, portable in (("simd", 0), ("portable", 1)):
        harness.hh_jpeg_force_portable(portable)
        for quality in (95, 50):
            n = harness.hh_encode_jpeg(gray.ctypes.data_as(C.c_void_p), w, h, quality, buf.ctypes.data_as(C.c_void_p),
                                       buf.size)
            assert n > 0
            dec[name, quality] = np.array(Image.open(io.BytesIO(buf[:n].tobytes()))).astype(int)
            assert dec[name, quality].shape == (h, w)
    harness.hh_jpeg_force_portable(0)
    assert np.abs(dec["simd", 95] - gray).mean() < 2.0 and np.abs(dec["portable", 95] - gray).mean() < 2.0
    for quality in (95, 50):  # same coefficients up to rounding ties -> essentially the same picture
        assert np.abs(dec["simd", quality] - dec["portable", quality]).mean() < 0.1


def test_png_is_compressed_and_lossless_on_a_smooth_flow(harness):
    from PIL import Image

    w, h = 320, 200
    yy, xx = np.mgrid[0:h, 0:w]
    fx = (2.5 * np.sin(xx / 40.0) * np.cos(yy / 35.0)).astype(np.float32)
    fy = (1.5 * np.cos(xx / 50.0)).astype(np.float32)
    buf = np.zeros(1 << 20, np.uint8)
    n = harness.hh_encode_flow_png(fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p), w, h,
                                   buf.ctypes.data_as(C.c_void_p), buf.size)
    assert 0 < n < 0.35 * w * h * 3  # filtered + deflated, not stored
    png = np.array(Image.open(io.BytesIO(buf[:n].tobytes())))
    bx = int(png[0, 0, 0]) * 4
    x_rec = (png[..., 2].astype(np.float64) - 128) * (bx / 128.0)
    assert np.abs(x_rec - fx).max() <= bx / 128.0


def test_png_and_jpeg_encoders_round_trip(harness):
    from PIL import Image

    rng = np.random.default_rng(1)
    w, h = 70, 45
    gray = (rng.uniform(0, 1, (h, w)) * 255).astype(np.uint8)
    buf = np.zeros(1 << 20, np.uint8)
    n = harness.hh_encode_jpeg(gray.ctypes.data_as(C.c_void_p), w, h, 100, buf.ctypes.data_as(C.c_void_p), buf.size)
    assert n > 0
    dec = np.array(Image.open(io.BytesIO(buf[:n].tobytes())))
    assert dec.shape == (h, w) and np.abs(dec.astype(int) - gray).max() <= 3
    fx = rng.uniform(-3, 3, (h, w)).astype(np.float32)
    fy = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    n = harness.hh_encode_flow_png(fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p), w, h,
                                   buf.ctypes.data_as(C.c_void_p), buf.size)
    assert n > 0
    png = np.array(Image.open(io.BytesIO(buf[:n].tobytes())))  # RGB; the shell wrote B=x, G=y, R=bound/4
    assert png.shape == (h, w, 3)
    bx = int(png[0, 0, 0]) * 4
    assert bx in (4, 12)  # ceil(3*128/127/4)*4 = 4, bumped by 4 only if divisible by 8
    x_rec = (png[..., 2].astype(np.float64) - 128) * (bx / 128.0)
    assert np.abs(x_rec - fx).max() <= bx / 128.0


@pytest.mark.gpu
def test_operator_matches_oracle_on_gpu(harness, oracle):
    w, h, n, step = 96, 64, 5, 1
    frames = np.stack(SynthClip(w, h, 5).frames(n))
    flows = np.zeros((n - 1, h, w, 2), np.float32)
    err = C.create_string_buffer(512)
    harness.hh_calc_optflows_imp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p,
                                             C.c_char_p, C.c_int]
    refs = {algo: [fn(frames[i], frames[i + 1]) for i in range(n - 1)]
            for algo, fn in (("tvl1", oracle.tvl1_calc), ("farn", oracle.farneback_calc))}
    for algo in ("tvl1", "farn"):
        m = harness.hh_calc_optflows_imp(frames.ctypes.data, n, w, h, algo.encode(), step, flows.ctypes.data, err, 512)
        assert m == n - 1, err.value
        for i in range(m):
            assert np.max(np.abs(flows[i] - refs[algo][i])) <= 1e-3
    m = harness.hh_calc_optflows_imp(frames.ctypes.data, n, w, h, b"nv", step, flows.ctypes.data, err, 512)
    assert m == -1 and err.value == b"NV hardware flow not enabled, pls recompile"


@pytest.mark.parametrize("sw,sh,dw,dh", [(64, 48, 32, 24), (340, 256, 224, 224), (100, 70, 224, 224), (33, 17, 64, 64),
                                         (640, 360, 455, 256), (7, 5, 3, 2), (64, 48, 64, 48)])
def test_host_resize_is_the_oracles_cv_resize(harness, oracle, sw, sh, dw, dh):
    src = np.random.default_rng(sw + dh).integers(0, 256, (sh, sw), dtype=np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    harness.hh_resize(src.ctypes.data_as(C.c_void_p), sw, sh, dst.ctypes.data_as(C.c_void_p), dw, dh)
    assert np.array_equal(dst, oracle.prepare_frame(src, dw, dh))


def test_concurrent_clip_opens_and_encoders_are_thread_safe(harness, tmp_path):
    """Regression: the Y4M header parser used strtok, so two loader threads (one per device in a multi-GPU run)
    corrupted each other's parse and clips came out with one frame.  Also: parallel JPEG encoders share only
    immutable tables."""
    w, h, n = 48, 32, 5
    frames = SynthClip(w, h, 3).frames(n)
    clip = tmp_path / "c.y4m"
    write_y4m(clip, frames)
    assert harness.hh_parallel_open_y4m(str(clip).encode(), 8, 300, w, h, n) == 0
    g = np.ascontiguousarray(frames[0])
    assert harness.hh_parallel_jpeg(g.ctypes.data_as(C.c_void_p), w, h, 8) == 1


def test_flow_buffer_queue_orders_blocks_and_closes(harness):
    harness.hh_queue_roundtrip.restype = C.c_long
    for n, depth in [(1, 1), (10, 1), (200, 3), (50, 64)]:
        assert harness.hh_queue_roundtrip(n, depth) == n * (n - 1) // 2
    assert harness.hh_queue_close_unblocks() == 1
    # byte budget (short clips may queue up for the joining flow stage): depth 1, but buffers are accepted while fewer
    # than `budget` bytes and fewer than `hard_max` buffers are queued; try_pop drains in order without blocking
    assert harness.hh_queue_budget(1000, 4500, 48) == 5   # the fifth push still sees 4000 < 4500 queued
    assert harness.hh_queue_budget(1000, 10 ** 6, 7) == 7  # hard_max
    assert harness.hh_queue_budget(1000, 0, 48) == 1       # no budget: the reference's bound alone


def test_parallel_for_covers_every_index_and_propagates_errors(harness):
    harness.hh_parallel_sum.restype = C.c_long
    for n, threads in [(0, 4), (1, 8), (100, 1), (1000, 7), (5, 64)]:
        assert harness.hh_parallel_sum(n, threads, -1) == n * (n - 1) // 2
    assert harness.hh_parallel_sum(200, 6, 57) == -1  # the exception reaches the caller, nothing hangs
    assert harness.hh_parallel_sum(200, 1, 199) == -1


@pytest.mark.gpu
def test_bounded_operator_matches_oracle_on_gpu(harness, oracle):
    """save_type "jpg": calc_optflows_imp hands the save stage planes that were bounded on the device."""
    w, h, n, step, bound = 96, 64, 5, 1, 20
    frames = np.stack(SynthClip(w, h, 5).frames(n))
    planes = np.zeros((n - 1, 2, h, w), np.uint8)
    err = C.create_string_buffer(512)
    harness.hh_calc_optflows_imp_bounded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                                     C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    refs = [oracle.flow_to_u8(oracle.tvl1_calc(frames[i], frames[i + 1]), -bound, bound) for i in range(n - 1)]
    m = harness.hh_calc_optflows_imp_bounded(frames.ctypes.data, n, w, h, b"tvl1", step, bound, planes.ctypes.data,
                                             err, 512)
    assert m == n - 1, err.value
    for i in range(m):
        assert np.array_equal(planes[i, 0], refs[i][0]) and np.array_equal(planes[i, 1], refs[i][1])


@pytest.mark.gpu
def test_cli_device_bounding_writes_the_same_files_as_host_bounding(built, tmp_path):
    """DF_HOST_BOUND=1 runs the reference's host-side convertFlowToImage; the default bounds on the GPU.
    Both feed the same encoder, so the JPEG files must be identical byte for byte."""
    w, h, n = 160, 120, 6
    frames = SynthClip(w, h, 4).frames(n)
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, frames)
    lst = tmp_path / "list.txt"
    lst.write_text(str(clip) + "\n")
    outs = {}
    for tag, env in (("dev", {"DF_HOST_JPEG": "1"}), ("host", {"DF_HOST_BOUND": "1", "DF_ENCODE_THREADS": "1"})):
        r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / tag), "-a=farn", "-s=2", "-b=8"],
                           capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stdout + r.stderr
        files = sorted(p.name for p in (tmp_path / tag / "clip").iterdir())
        outs[tag] = {f: (tmp_path / tag / "clip" / f).read_bytes() for f in files}
    assert len(outs["dev"]) == 2 * (n - 2) and outs["dev"].keys() == outs["host"].keys()
    assert all(outs["dev"][f] == outs["host"][f] for f in outs["dev"])


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,algo,step", [(160, 120, "farn", 2), (227, 131, "tvl1", 1), (640, 360, "farn", -1)])
def test_cli_device_jpeg_writes_the_same_files_as_the_host_encoders(built, tmp_path, w, h, algo, step):
    """The default save path for -st=jpg: flows are bounded AND JPEG-coded on the GPU (dfx_submit_batch_jpeg), the save
    stage only writes the files.  DF_HOST_JPEG=1 keeps the encoders on the host (device bounding only) and
    DF_HOST_BOUND=1 is the reference's all-host encodeFlowMap: all three must write the same bytes — across several
    short FlowBuffers (DF_BATCH_MAXSIZE) so that batches, tails and buffer reuse are all exercised."""
    n = 9
    frames = SynthClip(w, h, 14).frames(n)
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, frames)
    lst = tmp_path / "list.txt"
    lst.write_text(str(clip) + "\n")
    outs = {}
    for tag, env in (("gpu_jpeg", {"DF_BATCH_MAXSIZE": "4"}), ("host_jpeg", {"DF_HOST_JPEG": "1"}),
                     ("all_host", {"DF_HOST_BOUND": "1", "DF_ENCODE_THREADS": "2"})):
        r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / tag), "-a=" + algo, "-s=%d" % step, "-b=20"],
                           capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stdout + r.stderr
        files = sorted(p.name for p in (tmp_path / tag / "clip").iterdir())
        outs[tag] = {f: (tmp_path / tag / "clip" / f).read_bytes() for f in files}
    assert len(outs["gpu_jpeg"]) == 2 * (n - abs(step))
    for other in ("host_jpeg", "all_host"):
        assert outs["gpu_jpeg"].keys() == outs[other].keys()
        assert all(outs["gpu_jpeg"][f] == outs[other][f] for f in outs["gpu_jpeg"]), other


@pytest.mark.gpu
@pytest.mark.parametrize("st,env", [("jpg", {}), ("jpg", {"DF_HOST_JPEG": "1"}), ("png", {}), ("h5", {}), ("jpg", {"DF_TRACE": "1"})])
def test_cli_list_of_short_clips_is_joined_without_changing_a_byte(built, tmp_path, st, env):
    """A list of short clips (BASELINE configs[3] in small): the flow stage joins the clips that are already queued into one
    library call (dfx_next_segments) — every file must be what the unjoined run (DF_NO_JOIN=1) writes.  Clips of unequal
    lengths, one with no pair at all, one of another size in the middle (it ends a group), .done records on."""
    shapes = [(64, 48, 7), (64, 48, 5), (64, 48, 2), (64, 48, 9), (96, 64, 6), (64, 48, 4), (64, 48, 8)]
    lines = []
    for i, (w, h, n) in enumerate(shapes):
        clip = tmp_path / f"clip{i}.y4m"
        write_y4m(clip, SynthClip(w, h, 40 + i).frames(n))
        lines.append(str(clip))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(lines) + "\n")
    outs = {}
    for tag, extra in (("joined", {}), ("single", {"DF_NO_JOIN": "1"})):
        out = tmp_path / tag
        r = subprocess.run([built, str(lst), "-o=" + str(out), "-a=farn", "-s=2", "-b=20", "-st=" + st],
                           capture_output=True, text=True, env={**os.environ, **env, **extra})
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = {str(p.relative_to(out)): p.read_bytes() for p in sorted(out.rglob("*")) if p.is_file()}
        if "DF_TRACE" in env:  # the trace shows how many FlowBuffers went into each library call
            import re

            sizes = [int(m) for m in re.findall(r"frames of (\d+) FlowBuffer", r.stderr)]
            assert sizes and (max(sizes) > 1 if tag == "joined" else max(sizes) == 1), sizes
    assert outs["joined"].keys() == outs["single"].keys() and len(outs["joined"]) > 0
    if st == "jpg":
        assert sum(1 for f in outs["joined"] if f.endswith(".jpg")) == 2 * sum(max(n - 2, 0) for _, _, n in shapes)
    for f in outs["joined"]:
        assert outs["joined"][f] == outs["single"][f], f


@pytest.mark.gpu
@pytest.mark.parametrize("algo,step,size", [("tvl1", 1, (96, 64)), ("farn", -2, (130, 71)), ("brox", 1, (64, 48))])
def test_cli_png_files_do_not_depend_on_where_the_scheme_runs(built, tmp_path, algo, step, size):
    """-st=png on the real library: convertFlowToPngImage's arithmetic (/root/reference/src/common.cpp:18-46) on the device
    (dfx_submit_batch_png, the default: 2 bytes per pixel + the bounds come back) or on the host from the float flows
    (DF_HOST_PNG=1, the reference's place, 8 bytes per pixel): the same .png files, byte for byte — including FlowBuffers
    cut short (ragged batches)."""
    w, h = size
    n = 9
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, SynthClip(w, h, 23).frames(n))
    outs = {}
    for tag, env in (("device", {}), ("device_cut", {"DF_BATCH_MAXSIZE": "4"}), ("host", {"DF_HOST_PNG": "1"})):
        r = subprocess.run([built, str(clip), "-o=" + str(tmp_path / tag), "-a=" + algo, "-s=%d" % step, "-st=png"],
                           capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = {p.name: p.read_bytes() for p in sorted((tmp_path / tag / "clip").iterdir())}
    assert len(outs["device"]) == n - abs(step) and all(f.endswith(".png") for f in outs["device"])
    assert outs["device"] == outs["host"] and outs["device_cut"] == outs["host"]


def _write_pgm_dir(d, frames):
    d.mkdir(parents=True)
    for i, fr in enumerate(frames):
        with open(d / f"img_{i:05d}.pgm", "wb") as f:
            f.write(f"P5\n{fr.shape[1]} {fr.shape[0]}\n255\n".encode())
            f.write(np.ascontiguousarray(fr, np.uint8).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["video", "frames"])
@pytest.mark.parametrize("step", [1, 2])
def test_buffer_boundaries_do_not_change_the_output(built, tmp_path, source, step):
    """FlowBuffers are a pipelining unit only: cutting a video into short buffers (carrying |step| frames over,
    reference :204-207) must give the files of a single-buffer run, for clips and for image directories."""
    w, h, n = 64, 48, 11
    frames = SynthClip(w, h, 9).frames(n)
    if source == "video":
        src = tmp_path / "clip.y4m"
        write_y4m(src, frames)
        extra = []
    else:
        src = tmp_path / "clip"
        _write_pgm_dir(src, frames)
        extra = ["-if"]
    lst = tmp_path / "list.txt"
    lst.write_text(str(src) + "\n")
    outs = {}
    for tag, bm in (("one", "512"), ("cut", "4"), ("cut3", "3")):
        r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / tag), "-a=farn", f"-s={step}", "-b=8"] + extra,
                           capture_output=True, text=True, env={**os.environ, "DF_BATCH_MAXSIZE": bm})
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"({n} frames, {n - step} farn flows)" in r.stdout, r.stdout
        d = tmp_path / tag / "clip"
        outs[tag] = {p.name: p.read_bytes() for p in sorted(d.iterdir())}
    assert len(outs["one"]) == 2 * (n - step)
    assert outs["cut"] == outs["one"] and outs["cut3"] == outs["one"]
    # file names as the reference writes them (src/common.cpp:84-100): "_p<step>_" for step > 1
    first = "flow_x_00000.jpg" if step == 1 else f"flow_x_p{step}_00000.jpg"
    last = f"flow_y_{n - step - 1:05d}.jpg" if step == 1 else f"flow_y_p{step}_{n - step - 1:05d}.jpg"
    assert first in outs["one"] and last in outs["one"]


@pytest.mark.gpu
@pytest.mark.parametrize("resize_args", [["-nw=96", "-nh=72"], ["-ns=60"], ["-nw=100"]])
def test_cli_device_resize_writes_the_same_files_as_host_resize(built, tmp_path, resize_args):
    """-nw/-nh/-ns: by default the loader hands source-size frames to the GPU, which resizes them (dfx_set_source_format);
    DF_HOST_RESIZE=1 resizes on the loader thread like the reference.  Same arithmetic, so identical files."""
    w, h, n = 192, 144, 7
    frames = SynthClip(w, h, 6).frames(n)
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, frames)
    lst = tmp_path / "list.txt"
    lst.write_text(str(clip) + "\n")
    outs = {}
    for tag, env in (("dev", {}), ("host", {"DF_HOST_RESIZE": "1"})):
        r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / tag), "-a=tvl1", "-s=1", "-b=20"] + resize_args,
                           capture_output=True, text=True, env={**os.environ, "DF_BATCH_MAXSIZE": "4", **env})
        assert r.returncode == 0, r.stdout + r.stderr
        d = tmp_path / tag / "clip"
        outs[tag] = {p.name: p.read_bytes() for p in sorted(d.iterdir())}
    assert len(outs["dev"]) == 2 * (n - 1)
    assert outs["dev"] == outs["host"]
    from PIL import Image

    img = Image.open(tmp_path / "dev" / "clip" / "flow_x_00000.jpg")
    expect = {"-nw=96": (96, 72), "-ns=60": (80, 60), "-nw=100": (100, 75)}[resize_args[0]]
    assert img.size == expect


@pytest.mark.gpu
def test_cli_two_pipelines_in_one_process_give_the_single_pipeline_files(built, tmp_path):
    """The multi-GPU mode of the shell (one DenseFlow + one handle per device, videos dealt round-robin, no collective)
    exercised on one GPU: DF_DEVICES=0,0 runs two complete pipelines concurrently on device 0."""
    w, h, n, clips = 96, 72, 9, 5
    lst = tmp_path / "list.txt"
    names = []
    for c in range(clips):
        clip = tmp_path / f"v{c}.y4m"
        write_y4m(clip, SynthClip(w, h, 30 + c).frames(n))
        names.append(str(clip))
    lst.write_text("\n".join(names) + "\n")
    outs = {}
    for tag, env in (("one", {}), ("two", {"DF_DEVICES": "0,0"})):
        r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / tag), "-a=tvl1", "-s=1", "-b=20"],
                           capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"{clips} videos ({clips * n} frames, {clips * (n - 1)} tvl1 flows) processed" in r.stdout, (tag, r.stdout)
        outs[tag] = {str(p.relative_to(tmp_path / tag)): p.read_bytes()
                     for p in sorted((tmp_path / tag).rglob("*")) if p.is_file()}
    assert len(outs["one"]) == clips * (2 * (n - 1) + 1)  # flow_x / flow_y per pair + the .done marker per clip
    assert outs["one"] == outs["two"]


@pytest.mark.gpu
def test_cli_end_to_end_on_gpu(built, oracle, tmp_path):
    """BASELINE config 1 shape: a 224x224 pair sequence, -a=tvl1 -s=1 -b=20, files named like the reference's."""
    from PIL import Image

    w, h, n = 224, 224, 4
    frames = SynthClip(w, h, 1).frames(n)
    refs = [oracle.tvl1_calc(frames[i], frames[i + 1]) for i in range(n - 1)]
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, frames)
    lst = tmp_path / "list.txt"
    lst.write_text(str(clip) + "\n")
    r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / "out"), "-a=tvl1", "-s=1", "-b=20"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"1 videos ({n} frames, {n - 1} tvl1 flows) processed" in r.stdout and "done video" in r.stdout
    assert (tmp_path / "out" / ".done" / "clip").is_file()
    for i in range(n - 1):
        ref = refs[i]
        for c, name in enumerate(("flow_x", "flow_y")):
            img = np.array(Image.open(tmp_path / "out" / "clip" / f"{name}_{i:05d}.jpg")).astype(np.float64)
            q = np.rint(255 * (np.clip(ref[..., c].astype(np.float64), -20, 20) + 20) / 40)
            assert np.abs(img - q).mean() < 1.0  # JPEG q95 of the quantised oracle flow
    # second run: the .done marker makes it a no-op (resume)
    r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / "out"), "-a=tvl1", "-s=1", "-b=20", "-v"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "skip" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("step,algo", [(1, "tvl1"), (-2, "farn")])
def test_cli_h5_output_holds_the_unbounded_float_flows(built, oracle, tmp_path, step, algo):
    """-st=h5 (reference src/denseflow_gpu.cpp:223-243, :429-441, src/common.cpp:121-149): `<out>/<stem>.h5` (with the
    `_m2` infix for -s=-2) holds /flow_x_%05d and /flow_y_%05d, rank-2 float datasets = the two channels of each
    CV_32FC2 flow, unbounded.  Read back with the independent parser and compared with the oracle bit for bit."""
    from tests import h5_min_reader

    w, h, n = 96, 64, 7
    frames = SynthClip(w, h, 6).frames(n)
    calc = oracle.tvl1_calc if algo == "tvl1" else oracle.farneback_calc
    a = abs(step)
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, frames)
    env = dict(os.environ, DF_BATCH_MAXSIZE="4")  # several FlowBuffers: the file is re-opened and appended to
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path / "out"), f"-a={algo}", f"-s={step}", "-st=h5"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    name = "clip.h5" if step == 1 else ("clip_p%d.h5" % step if step > 1 else "clip_m%d.h5" % a)
    got = h5_min_reader.read(str(tmp_path / "out" / name))
    base = 0 if step > 0 else a
    infix = "" if step == 1 else ("_p%d" % step if step > 1 else "_m%d" % a)
    assert len(got) == 2 * (n - a)
    for i in range(n - a):
        fa, fb = (i, i + step) if step > 0 else (i - step, i)
        ref = calc(frames[fa], frames[fb])
        assert np.array_equal(got["flow_x%s_%05d" % (infix, i + base)], ref[..., 0]), i
        assert np.array_equal(got["flow_y%s_%05d" % (infix, i + base)], ref[..., 1]), i


@pytest.mark.gpu
def test_cli_videolist_sharded_over_device_pipelines_matches_the_oracle(built, harness, oracle, tmp_path):
    """BASELINE config 4 shape (a list of 224x224 clips, -a=tvl1, sharded over the GPUs of a node) at test size:
    8 clips through `-g` with two pipelines (on a 1-GPU box both on device 0, DF_DEVICES=0,0).  Every flow file of
    every clip must be the JPEG of the ORACLE's bounded flow: byte for byte what the shell's encoder (quality 95) makes
    of the oracle's plane, and a decodable image close to it."""
    from PIL import Image

    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]

    w, h, n, clips = 224, 224, 4, 8
    lst = tmp_path / "list.txt"
    refs = {}
    with open(lst, "w") as f:
        for c in range(clips):
            frames = SynthClip(w, h, 1000 + c).frames(n)  # seed 1000 + clip
            p = tmp_path / f"v{c:02d}.y4m"
            write_y4m(p, frames)
            f.write(str(p) + "\n")
            refs[c] = [oracle.flow_to_u8(oracle.tvl1_calc(frames[i], frames[i + 1]), -20, 20) for i in range(n - 1)]
    env = dict(os.environ, DF_DEVICES="0,0")
    r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / "out"), "-a=tvl1", "-s=1", "-b=20", "-g=2"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{clips} videos" in r.stdout
    for c in range(clips):
        assert (tmp_path / "out" / ".done" / f"v{c:02d}").is_file()
        for i in range(n - 1):
            for k, name in enumerate(("flow_x", "flow_y")):
                f = tmp_path / "out" / f"v{c:02d}" / f"{name}_{i:05d}.jpg"
                plane = np.ascontiguousarray(refs[c][i][k])
                buf = np.empty(plane.size * 2 + 4096, np.uint8)
                nb = harness.hh_encode_jpeg(plane.ctypes.data, w, h, 95, buf.ctypes.data, buf.size)
                assert nb > 0 and f.read_bytes() == buf[:nb].tobytes(), (c, i, name)
                img = np.array(Image.open(f)).astype(np.int32)
                assert img.shape == plane.shape and np.abs(img - plane).mean() < 0.6, (c, i, name)


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["video", "frames"])
@pytest.mark.parametrize("step", [1, -2, 3])
def test_cli_level2_split_of_one_clip_over_device_pipelines(built, tmp_path, source, step):
    """SURVEY.md §8e Level 2 (BASELINE configs 2/3/5 on a multi-GPU node): ONE clip, more devices than videos: every
    pipeline computes a contiguous range of the clip's flows (|step| overlap frames loaded twice, as the reference pads
    its own batches, src/denseflow_gpu.cpp:204-208) and writes them under their GLOBAL indices.  Three pipelines on one
    GPU (DF_DEVICES=0,0,0) must produce the single pipeline's files byte for byte, `.done` included."""
    w, h, n = 96, 72, 23
    frames = SynthClip(w, h, 40).frames(n)
    if source == "video":
        src = tmp_path / "clip.y4m"
        write_y4m(src, frames)
        extra = []
    else:
        src = tmp_path / "clip"
        _write_pgm_dir(src, frames)
        extra = ["-if"]
    lst = tmp_path / "list.txt"  # a list input records `.done/<stem>` (tools/denseflow.cpp:54-82)
    lst.write_text(str(src) + "\n")
    outs = {}
    for tag, env in (("one", {}), ("split", {"DF_DEVICES": "0,0,0", "DF_BATCH_MAXSIZE": "5"})):
        r = subprocess.run([built, str(lst), "-o=" + str(tmp_path / tag), "-a=farn", f"-s={step}", "-b=20"] + extra,
                           capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"{n - abs(step)} farn flows) processed" in r.stdout, (tag, r.stdout)
        outs[tag] = {str(p.relative_to(tmp_path / tag)): p.read_bytes()
                     for p in sorted((tmp_path / tag).rglob("*")) if p.is_file()}
    assert len(outs["one"]) == 2 * (n - abs(step)) + 1
    assert outs["one"] == outs["split"]


def test_shell_and_bench_split_a_clip_the_same_way(harness):
    """The host shell (DenseFlow::shard_range, -g with fewer videos than devices) and bench.py --split clip
    (denseflow_amd/shard.py) must cut a clip into the same contiguous flow ranges: disjoint, covering every flow
    once, each shard loading |step| overlap frames (reference padding logic src/denseflow_gpu.cpp:204-208)."""
    from denseflow_amd.shard import shard_pairs

    b, e = C.c_int(), C.c_int()
    for n in (0, 1, 2, 5, 23, 300, 513):
        for step in (1, -1, 2, -3, 7):
            for world in (1, 2, 3, 8):
                covered = []
                for rank in range(world):
                    harness.hh_shard_range(n, step, rank, world, C.byref(b), C.byref(e))
                    sh = shard_pairs(n, step, world, rank)
                    assert (b.value, e.value) == (sh.flow_begin, sh.flow_end), (n, step, world, rank)
                    if sh.n_flows:
                        assert sh.frame_begin == sh.flow_begin and sh.frame_end == sh.flow_end + abs(step)
                    covered += list(range(b.value, e.value))
                assert covered == list(range(max(n - abs(step), 0)))


@pytest.mark.parametrize("h", [7, 8, 9, 11, 191])
def test_png_bound_channel_split_and_float_scaling_follow_the_reference(harness, h):
    """convertFlowToPngImage (reference src/common.cpp:18-46): rows 0 .. int(h/2) carry bound_x/4, the rest bound_y/4
    (Point(w-1, half_h) truncates the double; ADVICE r1 found 5 rows instead of 4 at h = 7), and the two flow channels
    are Mat::convertTo(CV_8U, 1/(bound/128), 128) evaluated in FLOAT."""
    from PIL import Image

    rng = np.random.default_rng(h)
    w = 12
    fx = rng.uniform(-9, 9, (h, w)).astype(np.float32)
    fy = rng.uniform(-2, 2, (h, w)).astype(np.float32)
    buf = np.zeros(1 << 16, np.uint8)
    n = harness.hh_encode_flow_png(fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p), w, h,
                                   buf.ctypes.data_as(C.c_void_p), buf.size)
    png = np.array(Image.open(io.BytesIO(buf[:n].tobytes())))  # RGB order: R = third channel (bound), G = y, B = x

    def bound(f, lim):
        b = min(255.0 * 4, np.ceil((min(lim, float(np.abs(f).max())) * 128.0 / 127.0) / 4) * 4)
        return b + 4 if int(b) % 8 == 0 else b

    bx, by = bound(fx, w), bound(fy, h)
    rows_x = int(h / 2) + 1  # inclusive rectangle up to the truncated half height
    assert np.all(png[:rows_x, :, 0] == int(np.rint(bx / 4))) and np.all(png[rows_x:, :, 0] == int(np.rint(by / 4)))
    for f, b, ch in ((fx, bx, 2), (fy, by, 1)):
        inv = np.float32(1.0 / ((1.0 / 128.0) * b))
        v = f * inv + np.float32(128.0)  # float32 arithmetic, as cv::Mat::convertTo
        assert v.dtype == np.float32
        assert np.array_equal(png[..., ch], np.clip(np.rint(v.astype(np.float64)), 0, 255).astype(np.uint8))
