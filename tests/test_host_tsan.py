"""ThreadSanitizer pass over the multi-threaded host-shell pieces (CPU only).  It found two real races while the
multi-device mode was being tested on one GPU: strtok in the Y4M header parser and a lazily built CRC table shared by
parallel PNG encoders."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_shell_pieces_are_race_free_under_tsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    lib = os.path.join(ROOT, "denseflow_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libdfx.so")):
        pytest.skip("libdfx.so not built")
    exe = str(tmp_path / "tsan_host")
    srcs = [os.path.join(ROOT, "tests", "tsan_host.cpp")] + [os.path.join(ROOT, "src", f) for f in
                                                              ("common.cpp", "image_io.cpp", "utils.cpp", "h5mini.cpp",
                                                               "denseflow_gpu.cpp")]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include")] + srcs +
                       ["-L" + lib, "-ldfx", "-lpthread", "-lz", "-Wl,-rpath," + lib, "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("ThreadSanitizer build not available here: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]  # any other build failure is a failure (a missing source once hid this test)
    w, h, n = 64, 48, 4
    clip = tmp_path / "c.y4m"
    with open(clip, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 Cmono\n".encode())
        for i in range(n):
            f.write(b"FRAME\n")
            f.write(((np.arange(w * h, dtype=np.uint32) * 7 + i) & 0xFF).astype(np.uint8).tobytes())
    r = subprocess.run([exe, str(clip)], capture_output=True, text=True, env={**os.environ, "TSAN_OPTIONS": "halt_on_error=0"})
    assert "bad 0" in r.stdout, r.stdout + r.stderr[-2000:]
    assert "sum 124750 " in r.stdout, r.stdout  # 0 + ... + 499 through the queue, and the joining pattern kept its order
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0


def test_whole_pipeline_is_race_free_under_tsan(tmp_path):
    """The complete shell — loader, flow stage (joining queued clips), collector, save stage, two device pipelines — under
    ThreadSanitizer on the CPU, against the fake C ABI of tests/stub_dfx.cpp (test infrastructure; see
    tests/test_host_pipeline_stub.py)."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "denseflow_stub_tsan")
    srcs = [os.path.join(ROOT, "tools", "denseflow.cpp"), os.path.join(ROOT, "tests", "stub_dfx.cpp")] + [
        os.path.join(ROOT, "src", f) for f in ("common.cpp", "image_io.cpp", "utils.cpp", "h5mini.cpp", "denseflow_gpu.cpp")]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include")] + srcs +
                       ["-lpthread", "-lz", "-ldl", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("ThreadSanitizer build not available here: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = []
    for i in range(12):
        clip = tmp_path / f"c{i}.y4m"
        w, h, n = (48, 32, 5 + i % 4) if i != 7 else (64, 40, 6)
        with open(clip, "wb") as f:
            f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 Cmono\n".encode())
            for k in range(n):
                f.write(b"FRAME\n")
                f.write(((np.arange(w * h, dtype=np.uint32) * (3 + i) + 11 * k) & 0xFF).astype(np.uint8).tobytes())
        lines.append(str(clip))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    for extra_args, env in (([], {}), (["-g=2"], {"STUB_DEVICES": "2"}), (["-st=png"], {"DF_BATCH_MAXSIZE": "3"}),
                            (["-st=h5"], {}), ([], {"DF_HOST_JPEG": "1", "DF_ENCODE_THREADS": "4"})):
        out = tmp_path / ("o" + "".join(extra_args).replace("=", "").replace("-", "") + "".join(env))
        r = subprocess.run([exe, str(tmp_path / "list.txt"), "-o=" + str(out), "-a=farn", "-s=1"] + extra_args,
                           capture_output=True, text=True, env={**os.environ, "TSAN_OPTIONS": "halt_on_error=0", **env})
        assert "WARNING: ThreadSanitizer" not in r.stderr, (extra_args, env, r.stderr[-4000:])
        assert r.returncode == 0 and "12 videos" in r.stdout, r.stdout + r.stderr[-2000:]


def test_whole_pipeline_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The same pipeline under ASan + UBSan + LeakSanitizer (it found a left shift of a negative int in the shared DCT
    header).  Lists of clips of unequal sizes and lengths through jpg / png / h5, short FlowBuffers, two device pipelines,
    host bounding, the resize hand-off and the JPEG fallback."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "denseflow_stub_asan")
    srcs = [os.path.join(ROOT, "tools", "denseflow.cpp"), os.path.join(ROOT, "tests", "stub_dfx.cpp")] + [
        os.path.join(ROOT, "src", f) for f in ("common.cpp", "image_io.cpp", "utils.cpp", "h5mini.cpp", "denseflow_gpu.cpp")]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-I" + os.path.join(ROOT, "include")] + srcs + ["-lpthread", "-lz", "-ldl", "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("AddressSanitizer build not available here: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = []
    for i, (w, h, n) in enumerate([(64, 48, 7), (64, 48, 5), (64, 48, 2), (64, 48, 9), (96, 64, 6), (64, 48, 4), (45, 27, 8)]):
        clip = tmp_path / f"c{i}.y4m"
        with open(clip, "wb") as f:
            f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 Cmono\n".encode())
            for k in range(n):
                f.write(b"FRAME\n")
                f.write(((np.arange(w * h, dtype=np.uint32) * (3 + i) + 11 * k) & 0xFF).astype(np.uint8).tobytes())
        lines.append(str(clip))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    runs = (([], {}), (["-st=png"], {"DF_BATCH_MAXSIZE": "3"}), (["-st=h5"], {}), (["-g=2"], {"STUB_DEVICES": "2"}),
            ([], {"DF_HOST_BOUND": "1"}), (["-nw=40", "-nh=30"], {}), ([], {"STUB_JPEG_UNSUPPORTED": "1", "STUB_DELAY_MS": "10"}))
    for k, (extra_args, env) in enumerate(runs):
        r = subprocess.run([exe, str(tmp_path / "list.txt"), "-o=" + str(tmp_path / f"o{k}"), "-a=farn", "-s=2"] + extra_args,
                           capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1", **env})
        for mark in ("ERROR: AddressSanitizer", "runtime error", "LeakSanitizer"):
            assert mark not in r.stderr, (extra_args, env, r.stderr[-3000:])
        assert r.returncode == 0 and "7 videos" in r.stdout, r.stdout + r.stderr[-2000:]
    # malformed inputs: refused or read as far as they go, never a sanitizer report, never a hang
    rng = np.random.default_rng(3)
    bad = {
        "huge": b"YUV4MPEG2 W2000000000 H2000000000 F30:1 Ip A1:1 Cmono\nFRAME\n" + b"\0" * 100,
        "digits": b"YUV4MPEG2 W" + b"9" * 40 + b" H48 Cmono\nFRAME\n",
        "neg": b"YUV4MPEG2 W-5 H48 F30:1 Cmono\nFRAME\n" + b"\0" * 100,
        "zero": b"YUV4MPEG2 W0 H0 Cmono\nFRAME\n",
        "nofields": b"YUV4MPEG2\nFRAME\n",
        "longline": b"YUV4MPEG2 " + b"X" * 5000 + b"\nFRAME\n",
        "c420trunc": b"YUV4MPEG2 W64 H48 F30:1 C420\nFRAME\n" + b"\1" * (64 * 48) + b"\2" * 100,
        "frameparams": b"YUV4MPEG2 W8 H8 Cmono\n" + (b"FRAME Ip\n" + b"\1" * 64) * 3,
        "garbage": rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(),
    }
    for name, data in bad.items():
        (tmp_path / f"{name}.y4m").write_bytes(data)
        r = subprocess.run([exe, str(tmp_path / f"{name}.y4m"), "-o=" + str(tmp_path / "bad"), "-a=farn", "-s=1"],
                           capture_output=True, text=True, timeout=60, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
        for mark in ("AddressSanitizer", "runtime error"):
            assert mark not in r.stderr, (name, r.stderr[-3000:])
        assert r.returncode == 0 or "cannot open video_path stream" in r.stdout + r.stderr, (name, r.stdout, r.stderr[-500:])
    pg = tmp_path / "pgms"
    pg.mkdir()
    (pg / "img_00000.pgm").write_bytes(b"P5\n" + b"9" * 30 + b" 4\n255\n" + b"\0" * 16)
    r = subprocess.run([exe, str(pg), "--if", "-o=" + str(tmp_path / "badp"), "-a=farn", "-s=1"], capture_output=True,
                       text=True, timeout=60, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_the_library_helper_thread_is_race_free_under_tsan(tmp_path):
    """denseflow_amd/csrc/dfx_helper.h — the one persistent helper thread of a handle (it used to be a
    std::thread per batch) — under ThreadSanitizer, driven the way calc_batch_body's host-pointer path drives it with
    bounce buffers and device JPEG over 3 … 11 batches: hand-over of batch k-1 and gather of batch k+1 beside the owner's
    batch k, finish() before every reuse, a failing job, destruction with a job in flight."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "helper_tsan")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", os.path.join(ROOT, "tests", "helper_tsan.cpp"),
                        "-lpthread", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("ThreadSanitizer build not available here: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env={**os.environ, "TSAN_OPTIONS": "halt_on_error=0"})
    assert r.stdout.startswith("bad 0 "), r.stdout + r.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0
