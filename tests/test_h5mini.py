"""-st=h5 writer (src/h5mini.cpp; replaces the libhdf5 calls of /root/reference/src/common.cpp:121-149,
src/utils.cpp:28-41, src/denseflow_gpu.cpp:223-243): files are read back with an independent parser
(tests/h5_min_reader.py) and, where a libhdf5 exists (this container: /opt/conda/lib), with libhdf5 itself;
a file written by libhdf5's own H5Fcreate can be appended to."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import h5_min_reader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBHDF5 = next((p for p in ("/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
                            "/usr/lib/x86_64-linux-gnu/libhdf5.so") if os.path.exists(p)), None)


@pytest.fixture(scope="module")
def h5h():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libh5mini_harness.so")
    srcs = [os.path.join(ROOT, "tests", "h5mini_harness.cpp"), os.path.join(ROOT, "src", "h5mini.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", so]
                       + srcs, check=True)
    L = C.CDLL(so)
    L.h5h_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.h5h_append.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_char_p, C.c_int]
    L.h5h_count.argtypes = [C.c_char_p]
    return L


def _append(L, path, names, arr, pitch=None):
    n, rows, cols = arr.shape[0], arr.shape[1], arr.shape[2] if pitch is None else arr.shape[2]
    a = np.ascontiguousarray(arr, dtype=np.float32)
    err = C.create_string_buffer(512)
    nm = (C.c_char_p * n)(*[s.encode() for s in names])
    rc = L.h5h_append(path.encode(), n, nm, rows, pitch and cols or cols, a.shape[2], a.ctypes.data, err, 512)
    return rc, err.value.decode()


def _libhdf5_read(path):
    h5 = C.CDLL(LIBHDF5)
    h5.H5open()
    for f in ("H5Fopen", "H5Dopen2", "H5Dget_space"):
        getattr(h5, f).restype = C.c_int64
    h5.H5Fopen.argtypes = [C.c_char_p, C.c_uint, C.c_int64]
    h5.H5Dopen2.argtypes = [C.c_int64, C.c_char_p, C.c_int64]
    h5.H5Dget_space.argtypes = [C.c_int64]
    h5.H5Sget_simple_extent_dims.argtypes = [C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    h5.H5Dread.argtypes = [C.c_int64] * 5 + [C.c_void_p]
    h5.H5Gget_num_objs.argtypes = [C.c_int64, C.POINTER(C.c_uint64)]
    h5.H5Gget_objname_by_idx.argtypes = [C.c_int64, C.c_uint64, C.c_char_p, C.c_size_t]
    h5.H5Gget_objname_by_idx.restype = C.c_ssize_t
    for f in ("H5Fclose", "H5Dclose", "H5Sclose"):
        getattr(h5, f).argtypes = [C.c_int64]
    f32 = C.c_int64.in_dll(h5, "H5T_NATIVE_FLOAT_g").value
    fid = h5.H5Fopen(path.encode(), 0, 0)
    assert fid >= 0, "libhdf5 cannot open the file"
    n = C.c_uint64()
    assert h5.H5Gget_num_objs(fid, C.byref(n)) >= 0
    out = {}
    for i in range(n.value):
        buf = C.create_string_buffer(256)
        assert h5.H5Gget_objname_by_idx(fid, i, buf, 256) > 0
        did = h5.H5Dopen2(fid, b"/" + buf.value, 0)
        assert did >= 0
        sp = h5.H5Dget_space(did)
        dims = (C.c_uint64 * 8)()
        rank = h5.H5Sget_simple_extent_dims(sp, dims, None)
        a = np.empty(tuple(dims[k] for k in range(rank)), np.float32)
        assert h5.H5Dread(did, f32, 0, 0, 0, a.ctypes.data) >= 0
        out[buf.value.decode()] = a
        h5.H5Sclose(sp)
        h5.H5Dclose(did)
    assert h5.H5Fclose(fid) >= 0
    return out


def _check(path, expect):
    got = h5_min_reader.read(path)
    assert list(got) == sorted(expect), "names"
    for k, v in expect.items():
        assert np.array_equal(got[k], v), k
    if LIBHDF5:
        ref = _libhdf5_read(path)
        assert sorted(ref) == sorted(expect)
        for k, v in expect.items():
            assert np.array_equal(ref[k], v), k


def test_create_gives_an_empty_valid_file(h5h, tmp_path):
    p = str(tmp_path / "v.h5")
    err = C.create_string_buffer(512)
    assert h5h.h5h_create(p.encode(), err, 512) == 0, err.value
    assert h5h.h5h_count(p.encode()) == 0
    _check(p, {})


@pytest.mark.parametrize("n_buffers,per", [(1, 3), (3, 5), (2, 300)])
def test_flowbuffer_appends_accumulate(h5h, tmp_path, n_buffers, per):
    """The reference re-opens the file for every FlowBuffer (src/common.cpp:131) and adds flow_x / flow_y sets."""
    p = str(tmp_path / "v.h5")
    err = C.create_string_buffer(512)
    assert h5h.h5h_create(p.encode(), err, 512) == 0
    rng = np.random.default_rng(7)
    expect = {}
    for b in range(n_buffers):
        for phase in ("flow_x", "flow_y"):
            names = ["%s_%05d" % (phase, b * per + i) for i in range(per)]
            arr = rng.standard_normal((per, 6, 10)).astype(np.float32)
            rc, msg = _append(h5h, p, names, arr)
            assert rc == 0, msg
            expect.update({n: arr[i] for i, n in enumerate(names)})
    assert h5h.h5h_count(p.encode()) == 2 * n_buffers * per
    _check(p, expect)


def test_more_links_than_one_btree_node_holds(h5h, tmp_path):
    """2 * 16 children x 2 * 256 entries fit a single-level tree; 20000 links need a second B-tree level."""
    p = str(tmp_path / "big.h5")
    err = C.create_string_buffer(512)
    assert h5h.h5h_create(p.encode(), err, 512) == 0
    names = ["flow_x_%05d" % i for i in range(20000)]
    arr = np.arange(20000 * 2, dtype=np.float32).reshape(20000, 1, 2)
    rc, msg = _append(h5h, p, names, arr)
    assert rc == 0, msg
    _check(p, {n: arr[i] for i, n in enumerate(names)})


def test_duplicate_name_is_an_error_like_h5ltmake_dataset(h5h, tmp_path):
    p = str(tmp_path / "v.h5")
    err = C.create_string_buffer(512)
    assert h5h.h5h_create(p.encode(), err, 512) == 0
    a = np.zeros((1, 2, 2), np.float32)
    assert _append(h5h, p, ["flow_x_00000"], a)[0] == 0
    rc, msg = _append(h5h, p, ["flow_x_00000"], a)
    assert rc != 0 and "Failed to save hdf5 file" in msg


@pytest.mark.skipif(LIBHDF5 is None, reason="no libhdf5 on this machine")
def test_appending_to_a_file_created_by_libhdf5(h5h, tmp_path):
    """H5Fcreate by the real library (leaf K 4, so several symbol table nodes), datasets added here."""
    h5 = C.CDLL(LIBHDF5)
    h5.H5open()
    h5.H5Fcreate.restype = C.c_int64
    h5.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, C.c_int64, C.c_int64]
    h5.H5Fclose.argtypes = [C.c_int64]
    p = str(tmp_path / "lib.h5")
    fid = h5.H5Fcreate(p.encode(), 2, 0, 0)
    assert fid >= 0 and h5.H5Fclose(fid) >= 0
    rng = np.random.default_rng(3)
    arr = rng.standard_normal((40, 4, 7)).astype(np.float32)
    names = ["flow_y_p2_%05d" % i for i in range(40)]
    rc, msg = _append(h5h, p, names, arr)
    assert rc == 0, msg
    _check(p, {n: arr[i] for i, n in enumerate(names)})


def test_many_flowbuffers_grow_the_file_linearly(h5h, tmp_path):
    """Every append rewrites the group's index (heap + symbol table nodes + B-tree).  The old index is overwritten by
    the new datasets instead of being abandoned (the file grew roughly quadratically in the number of
    FlowBuffers): after 60 appends the file holds the data, ONE index and nothing else — and still reads back."""
    p = str(tmp_path / "long.h5")
    err = C.create_string_buffer(512)
    assert h5h.h5h_create(p.encode(), err, 512) == 0
    rng = np.random.default_rng(9)
    expect, n_buffers, per = {}, 60, 8
    sizes = []
    for b in range(n_buffers):
        for phase in ("flow_x", "flow_y"):
            names = ["%s_%05d" % (phase, b * per + i) for i in range(per)]
            arr = rng.standard_normal((per, 5, 9)).astype(np.float32)
            rc, msg = _append(h5h, p, names, arr)
            assert rc == 0, msg
            expect.update({n: arr[i] for i, n in enumerate(names)})
        sizes.append(os.path.getsize(p))
    n = 2 * n_buffers * per
    data_bytes = n * (5 * 9 * 4 + 4 + 144)           # data (padded to 8) + one 144-byte object header per dataset
    index_bytes = 32 + 8 + n * 24 + 16 + ((n + 511) // 512) * (8 + 512 * 40) + 24 + 32 * 8 + 33 * 8
    assert sizes[-1] <= 96 + 40 + data_bytes + index_bytes + 64, (sizes[-1], data_bytes, index_bytes)
    growth = np.diff(sizes)
    assert growth.max() <= 2 * (2 * per * (5 * 9 * 4 + 4 + 144 + 24)) + 8 + 512 * 40 + 64  # no term that grows with b
    _check(p, expect)


def test_failed_append_leaves_the_file_as_it_was(h5h, tmp_path):
    """an append that fails part-way must not cost the file its previous contents.  Two ways to fail:
    a duplicate name that is not the first dataset of the call (every name is now checked before the first write), and a
    write error in the middle of the call (RLIMIT_FSIZE in a child process: the bytes of the old index that the new
    datasets had already overwritten are put back)."""
    p = str(tmp_path / "v.h5")
    err = C.create_string_buffer(512)
    assert h5h.h5h_create(p.encode(), err, 512) == 0
    rng = np.random.default_rng(3)
    first = rng.standard_normal((2, 5, 7)).astype(np.float32)
    assert _append(h5h, p, ["a", "b"], first)[0] == 0
    expect = {"a": first[0], "b": first[1]}
    before = open(p, "rb").read()
    rc, msg = _append(h5h, p, ["c", "a"], rng.standard_normal((2, 5, 7)).astype(np.float32))
    assert rc != 0 and "dataset exists: a" in msg
    assert open(p, "rb").read() == before  # not a byte was written
    _check(p, expect)

    # a write error after the old index has been overwritten: the child may grow the file by 4 KB only, the append needs ~1 MB
    child = f"""
import ctypes as C, resource, signal, sys
import numpy as np
signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
L = C.CDLL({os.path.join(ROOT, "tests", "_build", "libh5mini_harness.so")!r})
L.h5h_append.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
n = 4
a = np.ones((n, 256, 256), np.float32)
nm = (C.c_char_p * n)(*[b"big%d" % i for i in range(n)])
err = C.create_string_buffer(512)
resource.setrlimit(resource.RLIMIT_FSIZE, ({len(before)} + 4096, resource.RLIM_INFINITY))
rc = L.h5h_append({p!r}.encode(), n, nm, 256, 256, 256, a.ctypes.data, err, 512)
print(rc, err.value.decode())
sys.exit(0 if rc != 0 else 3)
"""
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Failed to save hdf5 file" in r.stdout
    assert open(p, "rb").read()[:len(before)] == before  # the old index and the patched words are back
    _check(p, expect)
    # and the file still takes appends
    more = rng.standard_normal((1, 5, 7)).astype(np.float32)
    assert _append(h5h, p, ["c"], more)[0] == 0
    _check(p, dict(expect, c=more[0]))
