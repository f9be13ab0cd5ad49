"""The device-to-host leg of the host-pointer entry points (denseflow_amd/csrc/egress_kernels.hip).

Results leave the device through a copy kernel of a few persistent workgroups whenever every destination of a batch is
page-locked memory the device can write with 16-byte stores, and through hipMemcpyAsync otherwise.  The reference's
blocking `flow_gpu.download(flows[i])` (/root/reference/src/denseflow_gpu.cpp:339) into page-locked cv::Mat buffers
(tools/denseflow.cpp:49) is the page-locked case.  Every route must deliver the same bytes: page-locked buffers at odd
offsets inside one allocation, padded rows, both output kinds, the bounce-buffer path of small frames, pageable buffers,
widths that are not a multiple of 16 bytes, and the hipMemcpyAsync variant (DFX_VAR_D2H_MEMCPY)."""
import ctypes as C

import numpy as np
import pytest

from denseflow_amd.synth import SynthClip

pytestmark = pytest.mark.gpu


class Pinned:
    """A block of page-locked memory from dfx_host_alloc, carved into numpy views."""

    def __init__(self, L, nbytes):
        self.L, self.p = L, C.c_void_p()
        assert L.dfx_host_alloc(C.byref(self.p), nbytes) == 0
        self.buf = (C.c_uint8 * nbytes).from_address(self.p.value)
        self.arr = np.frombuffer(self.buf, dtype=np.uint8)
        self.arr[:] = 0xCD

    def view(self, offset, shape, dtype, pitch_bytes=None):
        item = np.dtype(dtype).itemsize
        rows, row_elems = shape[0], int(np.prod(shape[1:]))
        pitch = pitch_bytes or row_elems * item
        v = np.lib.stride_tricks.as_strided(self.arr[offset:].view(dtype), shape=(rows, row_elems), strides=(pitch, item))
        return v, self.p.value + offset, pitch

    def free(self):
        self.arr = self.buf = None
        self.L.dfx_host_free(self.p)


@pytest.mark.parametrize("algo,w,h,n,pad", [("farn", 640, 360, 7, 0), ("tvl1", 224, 224, 6, 0), ("farn", 320, 200, 5, 64),
                                            ("farn", 100, 64, 4, 0), ("brox", 352, 288, 3, 32)])
def test_every_route_delivers_the_same_bytes(dfx, algo, w, h, n, pad):
    from denseflow_amd import engine as E

    L = dfx.load_library()
    frames = SynthClip(w, h, 12).frames(n)
    m = n - 1
    with dfx.FlowEngine(w, h, algo, max_batch=3) as eng:  # pageable numpy outputs: hipMemcpyAsync (staged by the runtime)
        want_f = eng.calc_optflows(frames, 1)
        want_x, want_y = eng.calc_optflows_u8(frames, 1, 20)
    fpitch, ipitch = w * 8 + pad, w + pad
    blk = Pinned(L, 4096 + m * (fpitch * h + 2 * ipitch * h) + n * w * h + 4096)
    try:
        off = 48  # 16-byte aligned, not page aligned: the device view of an INTERIOR pointer is what has to be right
        fr_ptrs = []
        for f in frames:
            v, ptr, _ = blk.view(off, (h, w), np.uint8)
            v[:] = f
            fr_ptrs.append(ptr)
            off += w * h
        off = (off + 15) & ~15
        flows, fptrs, xs, xptrs, ys, yptrs = [], [], [], [], [], []
        for _ in range(m):
            v, ptr, _ = blk.view(off, (h, w * 2), np.float32, fpitch)
            flows.append(v), fptrs.append(ptr)
            off += fpitch * h
        for lst, ptrs in ((xs, xptrs), (ys, yptrs)):
            for _ in range(m):
                v, ptr, _ = blk.view(off, (h, w), np.uint8, ipitch)
                lst.append(v), ptrs.append(ptr)
                off += ipitch * h
        fp = (C.c_void_p * n)(*fr_ptrs)
        for variant in (0, E.VAR_D2H_MEMCPY):
            for wgs in ((0, 3) if variant == 0 else (0,)):
                for v in flows + xs + ys:
                    v[:] = 0
                with dfx.FlowEngine(w, h, algo, max_batch=3, variant=variant, egress_workgroups=wgs) as eng:
                    rc = L.dfx_calc_batch(eng._h, fp, w, n, 1, (C.c_void_p * m)(*fptrs), fpitch)
                    assert rc == 0, L.dfx_last_error(eng._h)
                    rc = L.dfx_calc_batch_u8(eng._h, fp, w, n, 1, -20.0, 20.0, (C.c_void_p * m)(*xptrs),
                                             (C.c_void_p * m)(*yptrs), ipitch)
                    assert rc == 0, L.dfx_last_error(eng._h)
                    # one FlowBuffer in flight: the tail of the first is collected while the second is on its way
                    t1, t2 = C.c_uint64(0), C.c_uint64(0)
                for i in range(m):
                    assert np.array_equal(flows[i].reshape(h, w, 2).view(np.uint32), want_f[i].view(np.uint32)), (variant, wgs, i)
                    assert np.array_equal(xs[i], want_x[i]) and np.array_equal(ys[i], want_y[i]), (variant, wgs, i)
                # nothing outside the rows was touched (row padding keeps its fill pattern)
                if pad:
                    tail = blk.arr[fptrs[0] - blk.p.value + w * 8: fptrs[0] - blk.p.value + fpitch]
                    assert np.all(tail == 0xCD)
    finally:
        blk.free()


def test_submitted_flowbuffers_with_page_locked_outputs(dfx):
    """dfx_submit_batch_u8 / dfx_wait with the copy kernel as the tail: two FlowBuffers in flight, bytes as synchronous."""
    L = dfx.load_library()
    w, h, n = 448, 256, 6
    frames = SynthClip(w, h, 3).frames(n)
    m = n - 1
    with dfx.FlowEngine(w, h, "farn", max_batch=2) as eng:
        want_x, want_y = eng.calc_optflows_u8(frames, 1, 20)
    blk = Pinned(L, n * w * h + 4 * m * w * h + 1024)
    try:
        off, fr = 0, []
        for f in frames:
            v, ptr, _ = blk.view(off, (h, w), np.uint8)
            v[:] = f
            fr.append(ptr)
            off += w * h
        sets = []
        for _ in range(2):
            xs, ys = [], []
            for lst in (xs, ys):
                for _ in range(m):
                    v, ptr, _ = blk.view(off, (h, w), np.uint8)
                    lst.append((v, ptr))
                    off += w * h
            sets.append((xs, ys))
        fp = (C.c_void_p * n)(*fr)
        with dfx.FlowEngine(w, h, "farn", max_batch=2) as eng:
            tickets = []
            for xs, ys in sets:
                t = C.c_uint64(0)
                rc = L.dfx_submit_batch_u8(eng._h, fp, w, n, 1, -20.0, 20.0, (C.c_void_p * m)(*[p for _, p in xs]),
                                           (C.c_void_p * m)(*[p for _, p in ys]), w, C.byref(t))
                assert rc == 0, L.dfx_last_error(eng._h)
                tickets.append(t.value)
            for t in tickets:
                assert L.dfx_wait(eng._h, t) == 0
        for xs, ys in sets:
            for i in range(m):
                assert np.array_equal(xs[i][0], want_x[i]) and np.array_equal(ys[i][0], want_y[i])
    finally:
        blk.free()
