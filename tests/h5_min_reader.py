"""An independent reader for the subset of HDF5 that -st=h5 files use (tests only; written from the HDF5 File
Format Specification, shares no code with src/h5mini.cpp): version-0 superblock, symbol-table root group,
version-1 object headers, contiguous little-endian float32 datasets."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _u(b, off, n):
    return int.from_bytes(b[off:off + n], "little")


def read(path):
    """-> dict name -> float32 array, in on-disk (sorted) order; plus structural checks via assert."""
    d = open(path, "rb").read()
    assert d[:8] == b"\x89HDF\r\n\x1a\n"
    assert d[8] == 0 and d[13] == 8 and d[14] == 8, "superblock v0, 8-byte offsets/lengths"
    leaf_k, internal_k = _u(d, 16, 2), _u(d, 18, 2)
    assert _u(d, 24, 8) == 0 and _u(d, 32, 8) == UNDEF and _u(d, 48, 8) == UNDEF
    eof = _u(d, 40, 8)
    assert eof <= len(d), "file shorter than its end-of-file address"
    root_hdr = _u(d, 56 + 8, 8)
    assert _u(d, 56 + 16, 4) == 1, "root entry caches the group's B-tree / heap"
    btree, heap = _u(d, 56 + 24, 8), _u(d, 56 + 32, 8)
    # root object header must say the same
    assert d[root_hdr] == 1
    msgs = _messages(d, root_hdr)
    st = [m for m in msgs if m[0] == 0x11]
    assert len(st) == 1 and (_u(st[0][1], 0, 8), _u(st[0][1], 8, 8)) == (btree, heap)
    assert d[heap:heap + 4] == b"HEAP"
    seg_size, free_head, seg_addr = _u(d, heap + 8, 8), _u(d, heap + 16, 8), _u(d, heap + 24, 8)
    seg = d[seg_addr:seg_addr + seg_size]
    assert seg[0] == 0, "offset 0 of the heap is the empty name"
    if free_head != 1:  # walk the free list: blocks inside the segment, terminated by 1
        seen = 0
        while free_head != 1:
            assert free_head + 16 <= seg_size and seen < 1000
            nxt, size = _u(seg, free_head, 8), _u(seg, free_head + 8, 8)
            assert size >= 16 and free_head + size <= seg_size
            free_head, seen = nxt, seen + 1
    out = {}
    names = []
    _walk(d, btree, seg, leaf_k, internal_k, names, None)
    assert names == sorted(names, key=lambda kv: kv[0].encode()), "links are sorted by name"
    assert len({n for n, _ in names}) == len(names)
    for name, hdr in names:
        out[name] = _dataset(d, hdr)
    return out


def _messages(d, hdr):
    assert d[hdr] == 1, "object header version 1"
    nmsg, size = _u(d, hdr + 2, 2), _u(d, hdr + 8, 4)
    off, end, res = hdr + 16, hdr + 16 + size, []
    for _ in range(nmsg):
        assert off + 8 <= end
        t, s = _u(d, off, 2), _u(d, off + 2, 2)
        assert s % 8 == 0
        res.append((t, d[off + 8:off + 8 + s]))
        off += 8 + s
    assert off == end, "messages fill the header exactly"
    return res


def _walk(d, addr, seg, leaf_k, internal_k, names, bound):
    assert d[addr:addr + 4] == b"TREE" and d[addr + 4] == 0
    level, used = d[addr + 5], _u(d, addr + 6, 2)
    assert used <= 2 * internal_k
    keys = [_u(d, addr + 24 + 16 * i, 8) for i in range(used + 1)]
    kids = [_u(d, addr + 32 + 16 * i, 8) for i in range(used)]
    for i, kid in enumerate(kids):
        lo, hi = _name(seg, keys[i]), _name(seg, keys[i + 1])
        if level > 0:
            _walk(d, kid, seg, leaf_k, internal_k, names, (lo, hi))
            continue
        assert d[kid:kid + 4] == b"SNOD" and d[kid + 4] == 1
        n = _u(d, kid + 6, 2)
        assert n <= 2 * leaf_k
        for k in range(n):
            e = kid + 8 + 40 * k
            nm = _name(seg, _u(d, e, 8))
            assert lo.encode() < nm.encode() <= hi.encode(), "B-tree keys bracket the node's names"
            assert _u(d, e + 16, 4) == 0
            names.append((nm, _u(d, e + 8, 8)))


def _name(seg, off):
    end = seg.index(b"\0", off)
    return seg[off:end].decode()


def _dataset(d, hdr):
    shape = dtype_ok = addr = size = None
    for t, body in _messages(d, hdr):
        if t == 1:
            assert body[0] == 1
            rank, flags = body[1], body[2]
            shape = tuple(_u(body, 8 + 8 * i, 8) for i in range(rank))
            if flags & 1:
                assert tuple(_u(body, 8 + 8 * rank + 8 * i, 8) for i in range(rank)) == shape
        elif t == 3:
            assert body[0] == 0x11 and body[1] == 0x20 and body[2] == 0x1F and _u(body, 4, 4) == 4
            assert struct.unpack("<HHBBBBI", body[8:20]) == (0, 32, 23, 8, 0, 23, 127)
            dtype_ok = True
        elif t == 8:
            assert body[0] == 3 and body[1] == 1, "layout v3, contiguous"
            addr, size = _u(body, 2, 8), _u(body, 10, 8)
    assert shape and dtype_ok and addr is not None
    n = int(np.prod(shape))
    assert size == 4 * n and addr % 8 == 0
    return np.frombuffer(d[addr:addr + size], dtype="<f4").reshape(shape).copy()
