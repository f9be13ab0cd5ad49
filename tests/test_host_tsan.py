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
