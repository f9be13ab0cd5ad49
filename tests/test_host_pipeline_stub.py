"""The host shell's own logic on the CPU: tools/denseflow.cpp + src/*.cpp linked against tests/stub_dfx.cpp — a FAKE of the
C ABI (include/dfx.h) that is test infrastructure only: its "flows" are a made-up function of the two frames.  What is tested
is everything around the library call: the loader / flow / collector / save threads, FlowBuffer boundaries, the joining of
short clips (dfx_next_segments), device pipelines (-g), frame-to-pair selection and output indices, naming, .done records,
the jpg / png / h5 writers, the resize hand-off.  The GPU suite runs the same CLI against the real library and the oracle
(tests/test_host_shell.py); the real CLI must refuse to run without a device."""
import ctypes as C
import io
import os
import subprocess

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip
from tests.test_host_shell import BIN, ROOT, _write_pgm_dir, built, harness, write_y4m  # noqa: F401

STUB = os.path.join(ROOT, "tests", "_build", "denseflow_stub")


@pytest.fixture(scope="module")
def stub(built):
    os.makedirs(os.path.dirname(STUB), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", STUB,
           os.path.join(ROOT, "tools", "denseflow.cpp"), os.path.join(ROOT, "tests", "stub_dfx.cpp"),
           os.path.join(ROOT, "build", "libzzdenseflow.a"), "-lpthread", "-lz", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return STUB


def _run(exe, args, env=None):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env={**os.environ, **(env or {})})
    assert r.returncode == 0, r.stdout + r.stderr
    return r


def _files(root):
    return {str(p.relative_to(root)): p.read_bytes() for p in sorted(root.rglob("*")) if p.is_file()}


def _fake_flow(a, b):
    """tests/stub_dfx.cpp: fake_flow, in float32 like the stub."""
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    h, w = a.shape
    x = np.arange(w, dtype=np.float32)[None, :]
    y = (np.arange(h) % 7).astype(np.float32)[:, None]
    u = (b - a) * np.float32(0.125) + np.float32(0.01) * x - np.float32(0.3)
    v = (np.roll(a, -1, axis=1) - b) * np.float32(0.0625) + np.float32(0.02) * y
    return np.stack([u, v], -1).astype(np.float32)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU box: the real CLI works here")
def test_the_real_cli_refuses_to_run_without_a_device(built, tmp_path):
    clip = tmp_path / "c.y4m"
    write_y4m(clip, SynthClip(64, 48, 1).frames(3))
    r = subprocess.run([built, str(clip), "-o=" + str(tmp_path / "o"), "-a=farn", "-s=1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device available" in r.stdout + r.stderr
    assert not any((tmp_path / "o").rglob("*.jpg"))


@pytest.mark.parametrize("algo", ["tvl1", "farn"])
def test_baseline_config_1_the_cli_on_a_cpu_with_the_oracle_as_its_backend(stub, harness, oracle, tmp_path, algo):
    """BASELINE.json configs[0] / SURVEY.md section 8d Config 1: a single 224x224 synthetic pair (seed 1), -s=1 -b=20, no GPU —
    plumbing only: the CLI, the file names flow_x_00000.jpg / flow_y_00000.jpg, and "oracle == backend".  The backend here is
    tests/stub_dfx.cpp handing the pair to the parity oracle (STUB_ORACLE); the files must be the pinned encoder's JPEGs of
    the oracle's flow bounded by the reference's own convertFlowToImage."""
    frames = SynthClip(224, 224, 1).frames(2)
    clip = tmp_path / "pair.y4m"
    write_y4m(clip, frames)
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    _run(stub, [clip, "-o=" + str(tmp_path / "out"), "-a=" + algo, "-s=1", "-b=20"], {"STUB_ORACLE": so, "OMP_NUM_THREADS": "8"})
    got = _files(tmp_path / "out" / "pair")
    assert sorted(got) == ["flow_x_00000.jpg", "flow_y_00000.jpg"]
    flow = (oracle.tvl1_calc if algo == "tvl1" else oracle.farneback_calc)(frames[0], frames[1])
    planes = oracle.ref_flow_to_u8(flow, -20.0, 20.0) if oracle.ref_quant_available() else oracle.flow_to_u8(flow, -20.0, 20.0)
    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    buf = np.zeros(1 << 20, np.uint8)
    for c, plane in zip("xy", planes):
        plane = np.ascontiguousarray(plane)
        k = harness.hh_encode_jpeg(plane.ctypes.data, 224, 224, 95, buf.ctypes.data, buf.size)
        assert got[f"flow_{c}_00000.jpg"] == buf[:k].tobytes(), c


@pytest.mark.parametrize("step", [1, 2, -1, -3])
def test_every_file_is_the_right_pair_under_the_right_name(stub, harness, oracle, tmp_path, step):
    """Frame-to-pair selection, bounding and naming through the whole pipeline: file flow_{x,y}[_pS|_mS]_%05d.jpg number i
    must hold the (made-up) flow of the reference's pair i — (i, i + s) for s > 0, (i - s, i) otherwise
    (/root/reference/src/denseflow_gpu.cpp:307-316), bounded by convertFlowToImage and encoded at quality 95 — across several
    FlowBuffers (DF_BATCH_MAXSIZE=4: |step| frames are carried over, reference :204-207)."""
    w, h, n = 64, 48, 11
    frames = SynthClip(w, h, 3).frames(n)
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, frames)
    _run(stub, [clip, "-o=" + str(tmp_path / "out"), "-a=tvl1", "-s=%d" % step, "-b=20"], {"DF_BATCH_MAXSIZE": "4"})
    got = _files(tmp_path / "out" / "clip")
    m = n - abs(step)
    tag = "" if step == 1 else ("_p%d" % step if step > 0 else "_m%d" % -step)
    base = 0 if step > 0 else -step  # writeFlowImages (reference src/common.cpp:73-100): names count from the LATER frame
    assert sorted(got) == sorted(f"flow_{c}{tag}_{i + base:05d}.jpg" for c in "xy" for i in range(m))
    harness.hh_encode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    buf = np.zeros(1 << 20, np.uint8)
    for i in range(m):
        a, b = (frames[i], frames[i + step]) if step > 0 else (frames[i - step], frames[i])
        for c, plane in zip("xy", oracle.flow_to_u8(_fake_flow(a, b), -20.0, 20.0)):
            plane = np.ascontiguousarray(plane)
            k = harness.hh_encode_jpeg(plane.ctypes.data, w, h, 95, buf.ctypes.data, buf.size)
            assert got[f"flow_{c}{tag}_{i + base:05d}.jpg"] == buf[:k].tobytes(), (c, i)


@pytest.mark.parametrize("st,env", [("jpg", {}), ("jpg", {"DF_HOST_JPEG": "1"}), ("jpg", {"STUB_JPEG_UNSUPPORTED": "1"}),
                                    ("jpg", {"DF_HOST_BOUND": "1"}), ("png", {}), ("png", {"DF_HOST_PNG": "1"}), ("h5", {})])
def test_a_list_of_short_clips_is_joined_without_changing_a_byte(stub, tmp_path, st, env):
    """The flow stage joins the queued FlowBuffers of one geometry into one library call; DF_NO_JOIN=1 does not.  Clips of
    unequal lengths, one without a pair, one of another size in the middle; every save type and every bounding / encoding
    place (device JPEG, host JPEG, the UNSUPPORTED fallback, all-host)."""
    import re

    shapes = [(64, 48, 7), (64, 48, 5), (64, 48, 2), (64, 48, 9), (96, 64, 6), (64, 48, 4), (64, 48, 8)]
    lines = []
    for i, (w, h, n) in enumerate(shapes):
        write_y4m(tmp_path / f"clip{i}.y4m", SynthClip(w, h, 40 + i).frames(n))
        lines.append(str(tmp_path / f"clip{i}.y4m"))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    outs, groups = {}, {}
    for tag, extra in (("joined", {}), ("single", {"DF_NO_JOIN": "1"})):
        r = _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / tag), "-a=farn", "-s=2", "-b=20", "-st=" + st],
                 {**env, **extra, "DF_TRACE": "1", "STUB_DELAY_MS": "40"})  # a slow "device": the loader runs ahead
        outs[tag] = _files(tmp_path / tag)
        groups[tag] = [int(g) for g in re.findall(r"frames of (\d+) FlowBuffer", r.stderr)]
    assert max(groups["single"]) == 1 and max(groups["joined"]) > 1, groups
    assert outs["joined"].keys() == outs["single"].keys() and outs["joined"]
    for f in outs["joined"]:
        assert outs["joined"][f] == outs["single"][f], f
    if st == "jpg":
        assert sum(f.endswith(".jpg") for f in outs["joined"]) == 2 * sum(max(n - 2, 0) for _, _, n in shapes)


@pytest.mark.parametrize("oracle_backed", [False, True])
def test_png_files_do_not_depend_on_where_the_scheme_runs(stub, oracle, tmp_path, oracle_backed):
    """-st=png: convertFlowToPngImage's arithmetic (src/common.cpp:18-46) on the "device" (dfx_submit_batch_png: two planes
    + the adaptive bounds come back, the save stage interleaves and encodes) or on the host from float flows
    (DF_HOST_PNG=1, the reference's place): the same files, byte for byte.  With the oracle as the fake's backend the
    device side is oracle/quant_oracle.c's restatement (pinned to the reference's own lines), so this also holds the
    host shell's own convertFlowToPngImage to the reference."""
    write_y4m(tmp_path / "clip.y4m", SynthClip(96, 64, 17).frames(7))
    env = {"STUB_ORACLE": os.path.join(ROOT, "oracle", "liboracle.so")} if oracle_backed else {}
    outs = {}
    for tag, extra in (("device", {}), ("host", {"DF_HOST_PNG": "1"})):
        _run(stub, [tmp_path / "clip.y4m", "-o=" + str(tmp_path / tag), "-a=farn", "-s=1", "-st=png"], {**env, **extra})
        outs[tag] = _files(tmp_path / tag)
    assert outs["device"].keys() == outs["host"].keys() and len(outs["device"]) == 6
    for f in outs["device"]:
        assert outs["device"][f] == outs["host"][f], f


def test_joining_really_happens_when_clips_are_waiting(stub, tmp_path):
    """With the loader ahead of the flow stage at least one library call must carry several FlowBuffers — otherwise the
    byte-for-byte test above would compare two unjoined runs — and every FlowBuffer is in exactly one call."""
    import re

    lines = []
    for i in range(24):
        write_y4m(tmp_path / f"c{i}.y4m", SynthClip(32, 24, i).frames(6))
        lines.append(str(tmp_path / f"c{i}.y4m"))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    # a "device" that takes 60 ms per call: while the first call runs the loader queues the other clips, so the next
    # call must take several of them along
    r = _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / "o"), "-a=farn", "-s=1"], {"DF_TRACE": "1", "STUB_DELAY_MS": "60"})
    sizes = [int(g) for g in re.findall(r"frames of (\d+) FlowBuffer", r.stderr)]
    assert max(sizes) > 1 and sum(sizes) == 24, sizes


@pytest.mark.parametrize("source", ["video", "frames"])
def test_flowbuffer_boundaries_and_device_pipelines_do_not_change_the_files(stub, tmp_path, source):
    """FlowBuffers are a pipelining unit and -g a sharding of independent work: short buffers, and one clip split over three
    device pipelines (Level 2: every pipeline loads its |step| overlap frames itself), must write the single-buffer,
    single-pipeline files."""
    w, h, n, step = 64, 48, 23, 2
    frames = SynthClip(w, h, 9).frames(n)
    if source == "video":
        src = tmp_path / "clip.y4m"
        write_y4m(src, frames)
        extra = []
    else:
        src = tmp_path / "clip"
        _write_pgm_dir(src, frames)
        extra = ["--if"]
    base = [src, "-a=farn", "-s=%d" % step, "-b=20"] + extra
    _run(stub, base + ["-o=" + str(tmp_path / "one")])
    _run(stub, base + ["-o=" + str(tmp_path / "short")], {"DF_BATCH_MAXSIZE": "5"})
    _run(stub, base + ["-o=" + str(tmp_path / "split"), "-g=3"], {"STUB_DEVICES": "3", "DF_BATCH_MAXSIZE": "4"})
    one = _files(tmp_path / "one")
    assert len([f for f in one if f.endswith(".jpg")]) == 2 * (n - step)
    assert _files(tmp_path / "short") == one
    assert _files(tmp_path / "split") == one


@pytest.mark.parametrize("what", ["one clip split by pair ranges", "a list dealt round-robin"])
def test_pipelines_as_processes_write_the_same_files(stub, tmp_path, what):
    """DF_PROCESSES=1: every device pipeline is a PROCESS (fork before any library call) instead of a thread set — Level 2
    (one clip, contiguous pair ranges) and Level 1 (a list) — and the parent adds up the children's counts for the
    reference's summary line.  Same files as one pipeline; a failing child ends the run with a non-zero status."""
    if what.startswith("one"):
        write_y4m(tmp_path / "clip.y4m", SynthClip(64, 48, 3).frames(11))
        src, n_frames, n_flows = tmp_path / "clip.y4m", 11, 10
    else:
        lines = []
        for i in range(5):
            write_y4m(tmp_path / f"c{i}.y4m", SynthClip(64, 48, 60 + i).frames(6 + i))
            lines.append(str(tmp_path / f"c{i}.y4m"))
        (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
        src, n_frames, n_flows = tmp_path / "list.txt", sum(6 + i for i in range(5)), sum(5 + i for i in range(5))
    outs = {}
    # "procs -g": -g=3 with DF_PROCESSES and NO device list — the parent counts the devices in a child of their own and
    # stays free of the runtime it is about to fork away from (the stub offers STUB_DEVICES of them)
    for tag, env in (("one", {}), ("threads", {"DF_DEVICES": "0,0,0"}), ("procs", {"DF_DEVICES": "0,0,0", "DF_PROCESSES": "1"}),
                     ("procs -g", {"DF_PROCESSES": "1", "STUB_DEVICES": "3", "STUB_TRACE_DEVICE_COUNT": "1"})):
        extra = ["-g=3"] if tag == "procs -g" else []
        r = _run(stub, [src, "-o=" + str(tmp_path / tag), "-a=farn", "-s=1", "-b=20"] + extra, env)
        if tag == "procs -g":  # the library was asked for its devices — never by the CLI's own process (children: the
            pids = [ln.split()[-1] for ln in r.stderr.splitlines() if ln.startswith("stub: dfx_device_count in pid")]  # counter + pipelines)
            main = [ln.split()[-1] for ln in r.stderr.splitlines() if ln.startswith("stub: main pid")]
            assert pids and len(main) == 1 and main[0] not in pids, r.stderr
        outs[tag] = {k: v for k, v in _files(tmp_path / tag).items() if ".done" not in k}
        assert f"{n_flows} farn flows) processed" in r.stdout, r.stdout
    assert outs["one"] and outs["threads"] == outs["one"] and outs["procs"] == outs["one"] and outs["procs -g"] == outs["one"]
    r = subprocess.run([stub, str(src), "-o=" + str(tmp_path / "bad"), "-a=farn", "-s=1"], capture_output=True, text=True,
                       env={**os.environ, "DF_DEVICES": "0,0", "DF_PROCESSES": "1", "STUB_FAIL_SUBMIT": "1"})
    assert r.returncode != 0 and "a pipeline process failed" in r.stdout, r.stdout + r.stderr


def test_a_list_sharded_over_device_pipelines_and_done_records(stub, tmp_path):
    """Level 1: the videos of a list are dealt to the device pipelines; the files and the .done records are those of one
    pipeline, and a second run skips every video that is marked done (no -f)."""
    lines = []
    for i in range(5):
        write_y4m(tmp_path / f"v{i}.y4m", SynthClip(48, 32, 60 + i).frames(4 + i))
        lines.append(str(tmp_path / f"v{i}.y4m"))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / "g1"), "-a=farn", "-s=1"])
    _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / "g2"), "-a=farn", "-s=1", "-g=2"], {"STUB_DEVICES": "2"})
    one, two = _files(tmp_path / "g1"), _files(tmp_path / "g2")
    assert one == two
    assert sum(f.endswith(".jpg") for f in one) == 2 * sum(3 + i for i in range(5))
    assert sorted(f for f in one if ".done" in f) == [f".done/v{i}" for i in range(5)]
    before = {f: os.stat(tmp_path / "g1" / f).st_mtime_ns for f in one}
    r = _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / "g1"), "-a=farn", "-s=1"])
    assert "processed" not in r.stdout  # every video is marked done: nothing is read, computed or written again
    assert {f: os.stat(tmp_path / "g1" / f).st_mtime_ns for f in one} == before
    r = _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / "g1"), "-a=farn", "-s=1", "-f"])  # -f: regardless
    assert "5 videos" in r.stdout and _files(tmp_path / "g1") == one


def test_resize_on_the_flow_stage_equals_resize_on_the_loader(stub, tmp_path):
    """-nw / -nh: by default the frames go to the library at their source size (dfx_set_source_format; the stub resizes with
    the host's resizeLinear), DF_HOST_RESIZE=1 resizes on the loader thread: the same frames reach the pair function."""
    clip = tmp_path / "clip.y4m"
    write_y4m(clip, SynthClip(80, 60, 5).frames(6))
    args = [clip, "-a=farn", "-s=1", "-nw=48", "-nh=32"]
    _run(stub, args + ["-o=" + str(tmp_path / "dev")])
    _run(stub, args + ["-o=" + str(tmp_path / "host")], {"DF_HOST_RESIZE": "1"})
    dev = _files(tmp_path / "dev")
    assert dev == _files(tmp_path / "host") and len(dev) >= 10
    from PIL import Image

    assert Image.open(io.BytesIO(next(v for k, v in dev.items() if k.endswith(".jpg")))).size == (48, 32)


def test_class_folder_layout_and_frame_extraction(stub, tmp_path):
    """--cf: outputDir/class/video/flow.jpg and .done/class/video (reference tools/denseflow.cpp:54-82, src/denseflow_gpu.cpp:
    456-470); -s=0: the frames themselves as img_%05d.jpg (reference :82-144 — gray here, the notice says so)."""
    lines = []
    for i, cls in enumerate(["catA", "catB", "catA"]):
        (tmp_path / cls).mkdir(exist_ok=True)
        write_y4m(tmp_path / cls / f"v{i}.y4m", SynthClip(48, 32, 60 + i).frames(4))
        lines.append(str(tmp_path / cls / f"v{i}.y4m"))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    r = _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / "o"), "-a=farn", "-s=1", "--cf"])
    assert all(f'done video "{c}/v{i}"' in r.stdout for i, c in enumerate(["catA", "catB", "catA"]))
    got = sorted(_files(tmp_path / "o"))
    assert got == sorted([f".done/{c}/v{i}" for i, c in enumerate(["catA", "catB", "catA"])] +
                         [f"{c}/v{i}/flow_{k}_{j:05d}.jpg" for i, c in enumerate(["catA", "catB", "catA"]) for k in "xy"
                          for j in range(3)])
    r = _run(stub, [tmp_path / "catA" / "v0.y4m", "-o=" + str(tmp_path / "f"), "-s=0"])
    assert "GRAY frames" in r.stdout
    assert sorted(_files(tmp_path / "f")) == [f"v0/img_{j:05d}.jpg" for j in range(4)]


def test_an_unreadable_video_ends_the_run_with_the_references_message(stub, tmp_path):
    """The reference throws "cannot open video_path stream:<path>" on its loader thread (src/denseflow_gpu.cpp:226-228), which
    ends the process.  Here: the same message, a non-zero exit status, no hang of the other stages, and no .done record for
    the video that could not be read."""
    write_y4m(tmp_path / "good.y4m", SynthClip(48, 32, 1).frames(4))
    (tmp_path / "bad.y4m").write_bytes(b"not a video")
    (tmp_path / "list.txt").write_text(f"{tmp_path / 'good.y4m'}\n{tmp_path / 'bad.y4m'}\n{tmp_path / 'good.y4m'}\n")
    r = subprocess.run([stub, str(tmp_path / "list.txt"), "-o=" + str(tmp_path / "o"), "-a=farn", "-s=1"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
    assert "cannot open video_path stream:" + str(tmp_path / "bad.y4m") in r.stdout + r.stderr
    assert not (tmp_path / "o" / ".done" / "bad").exists()


def test_degenerate_inputs_end_cleanly(stub, tmp_path):
    """Nothing to do must not hang a stage: an empty list, a header-only clip, a one-frame clip, a step longer than every clip
    of a list (the .done records still fire: an empty FlowBuffer travels the pipeline, reference :358), a truncated last
    frame (dropped), more device pipelines than pairs, `-a=nv` (the reference's message)."""
    def run(args, env=None):
        return subprocess.run([stub] + [str(a) for a in args], capture_output=True, text=True, timeout=60,
                              env={**os.environ, **(env or {})})

    (tmp_path / "empty.txt").write_text("")
    assert run([tmp_path / "empty.txt", "-o=" + str(tmp_path / "o1"), "-s=1"]).returncode == 0
    (tmp_path / "h.y4m").write_bytes(b"YUV4MPEG2 W64 H48 F30:1 Ip A1:1 Cmono\n")
    r = run([tmp_path / "h.y4m", "-o=" + str(tmp_path / "o2"), "-s=1", "-a=farn"])
    assert r.returncode == 0 and "(0 frames, 0 farn flows)" in r.stdout
    write_y4m(tmp_path / "one.y4m", SynthClip(64, 48, 1).frames(1))
    write_y4m(tmp_path / "three.y4m", SynthClip(64, 48, 1).frames(3))
    r = run([tmp_path / "one.y4m", "-o=" + str(tmp_path / "o3"), "-s=1", "-a=farn"])
    assert r.returncode == 0 and "(1 frames, 0 farn flows)" in r.stdout
    (tmp_path / "l.txt").write_text(f"{tmp_path / 'three.y4m'}\n{tmp_path / 'one.y4m'}\n")
    r = run([tmp_path / "l.txt", "-o=" + str(tmp_path / "o4"), "-s=5", "-a=farn"])
    assert r.returncode == 0 and sorted(_files(tmp_path / "o4")) == [".done/one", ".done/three"]
    data = (tmp_path / "three.y4m").read_bytes()
    (tmp_path / "trunc.y4m").write_bytes(data[:-100])
    r = run([tmp_path / "trunc.y4m", "-o=" + str(tmp_path / "o5"), "-s=1", "-a=farn"])
    assert r.returncode == 0 and sorted(_files(tmp_path / "o5")) == ["trunc/flow_x_00000.jpg", "trunc/flow_y_00000.jpg"]
    r = run([tmp_path / "three.y4m", "-o=" + str(tmp_path / "o6"), "-s=1", "-a=farn", "-g=4"], {"STUB_DEVICES": "4"})
    one = run([tmp_path / "three.y4m", "-o=" + str(tmp_path / "o7"), "-s=1", "-a=farn"])
    assert r.returncode == 0 and one.returncode == 0 and _files(tmp_path / "o6") == _files(tmp_path / "o7")
    r = run([tmp_path / "three.y4m", "-o=" + str(tmp_path / "o8"), "-s=1", "-a=nv"])
    assert r.returncode != 0 and "NV hardware flow not enabled, pls recompile" in r.stdout + r.stderr


@pytest.mark.parametrize("ticket", [1, 2, 3])
def test_a_failed_tail_stops_the_run_without_garbage_or_done_records(stub, tmp_path, ticket):
    """dfx_wait reports a failed deferred download for one FlowBuffer: nothing of that FlowBuffer or of a later
    one may reach the disk, its video must not be marked done, the process ends with the error and does not hang."""
    lines = []
    for i in range(4):
        write_y4m(tmp_path / f"v{i}.y4m", SynthClip(48, 32, 70 + i).frames(9))
        lines.append(str(tmp_path / f"v{i}.y4m"))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    good = tmp_path / "good"
    _run(stub, [tmp_path / "list.txt", "-o=" + str(good), "-a=farn", "-s=1"], {"DF_NO_JOIN": "1", "DF_BATCH_MAXSIZE": "5"})
    want = _files(good)
    r = subprocess.run([stub, str(tmp_path / "list.txt"), "-o=" + str(tmp_path / "bad"), "-a=farn", "-s=1"],
                       capture_output=True, text=True, timeout=60,
                       env={**os.environ, "DF_NO_JOIN": "1", "DF_BATCH_MAXSIZE": "5", "STUB_FAIL_WAIT": str(ticket)})
    assert r.returncode != 0 and "deferred download failed" in r.stdout + r.stderr
    got = _files(tmp_path / "bad")
    # every file that was written is a correct one (nothing of the failed FlowBuffer, no garbage) ...
    assert all(f in want and got[f] == want[f] for f in got), sorted(set(got) - set(want))
    # ... the run did not finish, and the video of the failed FlowBuffer (two FlowBuffers per video here) is not marked done
    assert len(got) < len(want)
    assert f".done/v{(ticket - 1) // 2}" not in got


def test_a_failed_library_call_ends_the_run_with_its_message(stub, tmp_path):
    """dfx_submit_batch* returns an error for the third FlowBuffer: the message reaches the user, the exit status is non-zero,
    what was written before is intact, the video being processed is not marked done, nothing hangs."""
    lines = []
    for i in range(3):
        write_y4m(tmp_path / f"v{i}.y4m", SynthClip(48, 32, 80 + i).frames(9))
        lines.append(str(tmp_path / f"v{i}.y4m"))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    env = {"DF_NO_JOIN": "1", "DF_BATCH_MAXSIZE": "5"}
    _run(stub, [tmp_path / "list.txt", "-o=" + str(tmp_path / "good"), "-a=farn", "-s=1"], env)
    want = _files(tmp_path / "good")
    r = subprocess.run([stub, str(tmp_path / "list.txt"), "-o=" + str(tmp_path / "bad"), "-a=farn", "-s=1"],
                       capture_output=True, text=True, timeout=60, env={**os.environ, **env, "STUB_FAIL_SUBMIT": "3"})
    assert r.returncode != 0 and "hipMemcpyAsync failed" in r.stdout + r.stderr
    got = _files(tmp_path / "bad")
    assert all(f in want and got[f] == want[f] for f in got) and len(got) < len(want)
    assert ".done/v1" not in got and ".done/v2" not in got


def test_the_stub_is_test_infrastructure_only():
    """Nothing the product builds or loads may know the fake: no CPU path hides behind the C ABI."""
    hits = []
    for top in ("denseflow_amd", "src", "tools", "include", "oracle"):
        for d, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".so", ".o", ".a", ".pyc", ".npz")):
                    continue
                with open(os.path.join(d, f), errors="ignore") as fh:
                    if "stub_dfx" in fh.read():
                        hits.append(os.path.join(d, f))
    for f in ("Makefile", "bench.py", "__graft_entry__.py"):
        with open(os.path.join(ROOT, f)) as fh:
            if "stub_dfx" in fh.read():
                hits.append(f)
    assert hits == []
