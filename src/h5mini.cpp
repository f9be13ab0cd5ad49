// src/h5mini.cpp — see include/h5mini.h.  Replaces the libhdf5 calls of the reference's -st=h5 path:
//   H5Fcreate / H5Fclose                 /root/reference/src/denseflow_gpu.cpp:236-237
//   H5Fopen / H5Fclose                   /root/reference/src/common.cpp:131, :146
//   H5LTmake_dataset_float               /root/reference/src/utils.cpp:35
// On-disk structures follow the HDF5 File Format Specification 1.x ("classic" groups); field layouts were checked
// byte for byte against a file written by libhdf5 1.10.6 (tests/test_h5mini.py keeps that comparison).
#include "h5mini.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <unordered_set>

namespace h5mini {
namespace {

constexpr uint64_t UNDEF = ~0ull;
constexpr uint64_t SUPERBLOCK_SIZE = 96; // version 0, 8-byte offsets and lengths, incl. the root symbol table entry
constexpr uint64_t ROOT_HEADER_ADDR = 96;
constexpr unsigned LEAF_K = 256;     // symbol table node capacity 2 * LEAF_K entries (libhdf5's default is 4)
constexpr unsigned INTERNAL_K = 16;  // B-tree node capacity 2 * INTERNAL_K children (libhdf5's default)

struct Buf {
    std::vector<uint8_t> b;
    void u8(unsigned v) { b.push_back((uint8_t)v); }
    void u16(unsigned v) {
        u8(v & 0xff);
        u8((v >> 8) & 0xff);
    }
    void u32(uint32_t v) {
        for (int i = 0; i < 4; ++i)
            u8((v >> (8 * i)) & 0xff);
    }
    void u64(uint64_t v) {
        for (int i = 0; i < 8; ++i)
            u8((unsigned)((v >> (8 * i)) & 0xff));
    }
    void bytes(const void *p, size_t n) {
        const uint8_t *q = (const uint8_t *)p;
        b.insert(b.end(), q, q + n);
    }
    void zeros(size_t n) { b.insert(b.end(), n, 0); }
    void pad8() {
        while (b.size() % 8)
            u8(0);
    }
};

struct FileCloser {
    FILE *f;
    ~FileCloser() {
        if (f)
            fclose(f);
    }
};

[[noreturn]] void fail(const std::string &path, const char *what) {
    throw std::runtime_error(std::string("Failed to save hdf5 file: ") + path + " (" + what + ")");
}

uint64_t rd(const uint8_t *p, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i)
        v |= (uint64_t)p[i] << (8 * i);
    return v;
}

void pread_exact(FILE *f, const std::string &path, uint64_t off, void *dst, size_t n) {
    if (fseek(f, (long)off, SEEK_SET) != 0 || fread(dst, 1, n, f) != n)
        fail(path, "short read");
}
void pwrite_exact(FILE *f, const std::string &path, uint64_t off, const void *src, size_t n) {
    if (fseek(f, (long)off, SEEK_SET) != 0 || fwrite(src, 1, n, f) != n)
        fail(path, "short write");
}

struct Entry {
    std::string name;
    uint64_t header_addr;
};

struct Index {
    unsigned leaf_k = LEAF_K, internal_k = INTERNAL_K;
    uint64_t eof = 0, root_header = 0, btree = UNDEF, heap = UNDEF;
    uint64_t root_msg_data_off = 0; // file offset of the symbol-table message's 16 data bytes in the root header
    std::vector<Entry> entries;
};

void walk_btree(FILE *f, const std::string &path, uint64_t addr, const std::vector<uint8_t> &heap_data, Index &ix,
                int depth) {
    if (depth > 8)
        fail(path, "B-tree too deep");
    uint8_t h[24];
    pread_exact(f, path, addr, h, 24);
    if (memcmp(h, "TREE", 4) != 0 || h[4] != 0)
        fail(path, "not a group B-tree node");
    const unsigned level = h[5], used = (unsigned)rd(h + 6, 2);
    std::vector<uint8_t> body((size_t)used * 16 + 8);
    pread_exact(f, path, addr + 24, body.data(), body.size());
    for (unsigned i = 0; i < used; ++i) {
        const uint64_t child = rd(body.data() + 8 + (size_t)i * 16, 8);
        if (level > 0) {
            walk_btree(f, path, child, heap_data, ix, depth + 1);
            continue;
        }
        uint8_t sh[8];
        pread_exact(f, path, child, sh, 8);
        if (memcmp(sh, "SNOD", 4) != 0)
            fail(path, "not a symbol table node");
        const unsigned n = (unsigned)rd(sh + 6, 2);
        std::vector<uint8_t> ents((size_t)n * 40);
        pread_exact(f, path, child + 8, ents.data(), ents.size());
        for (unsigned k = 0; k < n; ++k) {
            const uint64_t noff = rd(ents.data() + (size_t)k * 40, 8);
            if (noff >= heap_data.size())
                fail(path, "link name outside the heap");
            const char *s = (const char *)heap_data.data() + noff;
            const size_t len = strnlen(s, heap_data.size() - noff);
            ix.entries.push_back(Entry{std::string(s, len), rd(ents.data() + (size_t)k * 40 + 8, 8)});
        }
    }
}

Index read_index(FILE *f, const std::string &path) {
    Index ix;
    uint8_t sb[SUPERBLOCK_SIZE];
    pread_exact(f, path, 0, sb, sizeof sb);
    static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    if (memcmp(sb, sig, 8) != 0 || sb[8] != 0 || sb[13] != 8 || sb[14] != 8)
        fail(path, "unsupported superblock (expected version 0 with 8-byte offsets)");
    ix.leaf_k = (unsigned)rd(sb + 16, 2);
    ix.internal_k = (unsigned)rd(sb + 18, 2);
    if (rd(sb + 24, 8) != 0)
        fail(path, "non-zero base address");
    ix.eof = rd(sb + 40, 8);
    ix.root_header = rd(sb + 64, 8);
    // the symbol table message of the root object header (version 1, messages start 16 bytes in)
    uint8_t oh[16];
    pread_exact(f, path, ix.root_header, oh, 16);
    if (oh[0] != 1)
        fail(path, "unsupported root object header version");
    const unsigned nmsg = (unsigned)rd(oh + 2, 2);
    const uint64_t hsize = rd(oh + 8, 4);
    std::vector<uint8_t> msgs(hsize);
    pread_exact(f, path, ix.root_header + 16, msgs.data(), msgs.size());
    size_t off = 0;
    for (unsigned m = 0; m < nmsg && off + 8 <= msgs.size(); ++m) {
        const unsigned type = (unsigned)rd(msgs.data() + off, 2), size = (unsigned)rd(msgs.data() + off + 2, 2);
        if (type == 0x11 && size >= 16 && off + 8 + 16 <= msgs.size()) {
            ix.btree = rd(msgs.data() + off + 8, 8);
            ix.heap = rd(msgs.data() + off + 16, 8);
            ix.root_msg_data_off = ix.root_header + 16 + off + 8;
        }
        off += 8 + size;
    }
    if (ix.btree == UNDEF || ix.heap == UNDEF)
        fail(path, "root group is not a symbol table");
    uint8_t hh[32];
    pread_exact(f, path, ix.heap, hh, 32);
    if (memcmp(hh, "HEAP", 4) != 0)
        fail(path, "bad local heap");
    std::vector<uint8_t> heap_data(rd(hh + 8, 8));
    pread_exact(f, path, rd(hh + 24, 8), heap_data.data(), heap_data.size());
    walk_btree(f, path, ix.btree, heap_data, ix, 0);
    return ix;
}

// version-1 object header of a contiguous rank-2 IEEE float32 little-endian dataset (messages as libhdf5 writes them)
Buf dataset_header(uint64_t rows, uint64_t cols, uint64_t data_addr) {
    Buf m;
    // dataspace, version 1, rank 2, maximum dimensions present
    m.u16(0x0001), m.u16(40), m.u8(0), m.zeros(3);
    m.u8(1), m.u8(2), m.u8(1), m.u8(0), m.u32(0);
    m.u64(rows), m.u64(cols), m.u64(rows), m.u64(cols);
    // datatype, version 1 class 1 (floating point): little-endian, implied-msb mantissa, sign bit 31, 4 bytes;
    // bit offset 0, precision 32, exponent at 23 (8 bits), mantissa at 0 (23 bits), bias 127.  flag 1 = constant
    m.u16(0x0003), m.u16(24), m.u8(1), m.zeros(3);
    m.u8(0x11), m.u8(0x20), m.u8(0x1f), m.u8(0x00), m.u32(4);
    m.u16(0), m.u16(32), m.u8(23), m.u8(8), m.u8(0), m.u8(23), m.u32(127), m.u32(0);
    // fill value, version 2: allocate late, write if set, defined, size 0
    m.u16(0x0005), m.u16(8), m.u8(1), m.zeros(3);
    m.u8(2), m.u8(2), m.u8(2), m.u8(1), m.u32(0);
    // data layout, version 3, contiguous
    m.u16(0x0008), m.u16(24), m.u8(0), m.zeros(3);
    m.u8(3), m.u8(1), m.u64(data_addr), m.u64(rows * cols * 4), m.zeros(6);
    Buf h;
    h.u8(1), h.u8(0), h.u16(4), h.u32(1), h.u32((uint32_t)m.b.size()), h.u32(0);
    h.bytes(m.b.data(), m.b.size());
    return h;
}

struct Writer {
    FILE *f;
    const std::string &path;
    uint64_t pos; // next free file offset (8-byte aligned)
    uint64_t put(const Buf &b) {
        const uint64_t at = pos;
        pwrite_exact(f, path, at, b.b.data(), b.b.size());
        pos = (at + b.b.size() + 7) & ~7ull;
        return at;
    }
};

// Heap + symbol table nodes + B-tree for `entries` (sorted by name); returns (btree root, heap) addresses.
void write_group(Writer &w, const std::vector<Entry> &entries, unsigned leaf_k, unsigned internal_k, uint64_t &btree,
                 uint64_t &heap) {
    // local heap: offset 0 is the empty string, names are 8-byte aligned, one free block closes the segment
    Buf seg;
    seg.zeros(8);
    std::vector<uint64_t> name_off(entries.size());
    for (size_t i = 0; i < entries.size(); ++i) {
        name_off[i] = seg.b.size();
        seg.bytes(entries[i].name.c_str(), entries[i].name.size() + 1);
        seg.pad8();
    }
    const uint64_t free_off = seg.b.size();
    seg.u64(1), seg.u64(16); // H5HL_FREE_NULL terminates the free list; the block covers itself
    Buf hp;
    hp.bytes("HEAP", 4), hp.u8(0), hp.zeros(3);
    hp.u64(seg.b.size()), hp.u64(free_off), hp.u64(w.pos + 32);
    hp.bytes(seg.b.data(), seg.b.size());
    heap = w.put(hp);

    // symbol table nodes
    const size_t per = 2 * (size_t)leaf_k;
    struct Leaf {
        uint64_t addr, max_name_off;
    };
    std::vector<Leaf> leaves;
    for (size_t i = 0; i < entries.size() || leaves.empty(); i += per) {
        const size_t n = entries.empty() ? 0 : std::min(per, entries.size() - i);
        Buf s;
        s.bytes("SNOD", 4), s.u8(1), s.u8(0), s.u16((unsigned)n);
        for (size_t k = 0; k < n; ++k) {
            s.u64(name_off[i + k]), s.u64(entries[i + k].header_addr), s.u32(0), s.u32(0), s.zeros(16);
        }
        s.zeros((per - n) * 40);
        leaves.push_back(Leaf{w.put(s), n ? name_off[i + n - 1] : 0});
        if (entries.empty())
            break;
    }
    // B-tree levels, bottom up
    struct Node {
        uint64_t addr, max_name_off;
    };
    std::vector<Node> level;
    for (const Leaf &l : leaves)
        level.push_back(Node{l.addr, l.max_name_off});
    const size_t fan = 2 * (size_t)internal_k;
    const size_t node_bytes = 24 + fan * 8 + (fan + 1) * 8;
    for (unsigned lvl = 0;; ++lvl) {
        const size_t n_nodes = (level.size() + fan - 1) / fan;
        std::vector<Node> up;
        const uint64_t first_addr = w.pos;
        for (size_t j = 0; j < n_nodes; ++j) {
            const size_t lo = j * fan, n = std::min(fan, level.size() - lo);
            Buf t;
            t.bytes("TREE", 4), t.u8(0), t.u8(lvl), t.u16((unsigned)n);
            t.u64(j > 0 ? first_addr + (j - 1) * node_bytes : UNDEF);
            t.u64(j + 1 < n_nodes ? first_addr + (j + 1) * node_bytes : UNDEF);
            t.u64(lo > 0 ? level[lo - 1].max_name_off : 0); // key 0: everything in child 0 is greater than this name
            for (size_t k = 0; k < n; ++k) {
                t.u64(level[lo + k].addr), t.u64(level[lo + k].max_name_off);
            }
            t.zeros(node_bytes - t.b.size());
            up.push_back(Node{w.put(t), level[lo + n - 1].max_name_off});
        }
        if (up.size() == 1) {
            btree = up[0].addr;
            return;
        }
        level.swap(up);
    }
}

void finish(Writer &w, uint64_t root_header, uint64_t root_msg_data_off, uint64_t btree, uint64_t heap) {
    Buf a;
    a.u64(btree), a.u64(heap);
    pwrite_exact(w.f, w.path, root_msg_data_off, a.b.data(), 16); // symbol table message of the root group
    pwrite_exact(w.f, w.path, 56 + 24, a.b.data(), 16);           // scratch pad of the root symbol table entry
    Buf e;
    e.u64(w.pos);
    pwrite_exact(w.f, w.path, 40, e.b.data(), 8); // end-of-file address
    (void)root_header;
    // libhdf5 checks that the file is at least as long as the end-of-file address
    if (fseek(w.f, 0, SEEK_END) != 0)
        fail(w.path, "seek");
    const long len = ftell(w.f);
    if ((uint64_t)len < w.pos) {
        std::vector<uint8_t> z(w.pos - (uint64_t)len, 0);
        pwrite_exact(w.f, w.path, (uint64_t)len, z.data(), z.size());
    }
    if (fflush(w.f) != 0)
        fail(w.path, "flush");
}

} // namespace

void create(const std::string &path) {
    FileCloser fc{fopen(path.c_str(), "wb+")};
    if (!fc.f)
        fail(path, "cannot create");
    Buf sb;
    static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    sb.bytes(sig, 8);
    sb.u8(0), sb.u8(0), sb.u8(0), sb.u8(0), sb.u8(0), sb.u8(8), sb.u8(8), sb.u8(0);
    sb.u16(LEAF_K), sb.u16(INTERNAL_K), sb.u32(0);
    sb.u64(0), sb.u64(UNDEF), sb.u64(0) /* eof, patched */, sb.u64(UNDEF);
    sb.u64(0), sb.u64(ROOT_HEADER_ADDR), sb.u32(1), sb.u32(0), sb.u64(0), sb.u64(0); // root entry, scratch patched
    // root object header: one symbol table message
    Buf rh;
    rh.u8(1), rh.u8(0), rh.u16(1), rh.u32(1), rh.u32(24), rh.u32(0);
    rh.u16(0x0011), rh.u16(16), rh.u8(0), rh.zeros(3), rh.u64(0), rh.u64(0);
    Writer w{fc.f, path, 0};
    w.put(sb);
    if (w.pos != SUPERBLOCK_SIZE)
        fail(path, "internal: superblock size");
    w.put(rh);
    uint64_t btree, heap;
    write_group(w, {}, LEAF_K, INTERNAL_K, btree, heap);
    finish(w, ROOT_HEADER_ADDR, ROOT_HEADER_ADDR + 16 + 8, btree, heap);
}

std::vector<std::string> list(const std::string &path) {
    FileCloser fc{fopen(path.c_str(), "rb")};
    if (!fc.f)
        fail(path, "cannot open");
    const Index ix = read_index(fc.f, path);
    std::vector<std::string> names;
    for (const Entry &e : ix.entries)
        names.push_back(e.name);
    return names;
}

void append(const std::string &path, const std::vector<FloatDataset> &datasets) {
    FileCloser fc{fopen(path.c_str(), "rb+")};
    if (!fc.f)
        fail(path, "cannot open");
    Index ix = read_index(fc.f, path);
    // Where the new datasets go.  This writer lays a file out as [superblock, root header][datasets ...][index], the index
    // (local heap, then the symbol table nodes, then the B-tree whose root node is written last) being rewritten by every
    // append.  When the old index is that tail — heap first, above every dataset header, root node ending at the
    // end-of-file address — the new datasets overwrite it and the new index follows them: the file grows by the data
    // it holds, not quadratically by one abandoned index per FlowBuffer.  (A file whose group structures sit elsewhere,
    // e.g. one created by libhdf5 itself, is appended to at its end once and has the tail layout from then on.)
    uint64_t start = (ix.eof + 7) & ~7ull;
    {
        const uint64_t fan = 2 * (uint64_t)ix.internal_k;
        const uint64_t root_end = (ix.btree + 24 + fan * 8 + (fan + 1) * 8 + 7) & ~7ull;
        uint64_t hdr_hi = ix.root_header + 16 + 24; // end of the root object header written by create()
        for (const Entry &e : ix.entries)
            hdr_hi = std::max(hdr_hi, e.header_addr + 16 + 128); // a dataset header of dataset_header() is 136 bytes
        if (ix.heap < ix.btree && ix.heap >= hdr_hi && root_end == start)
            start = ix.heap;
    }
    // Every name is checked before the first byte is written (H5LTmake_dataset_float fails on an existing name; a
    // duplicate found half-way used to leave the old index overwritten: ADVICE r3).
    {
        std::unordered_set<std::string> names;
        for (const Entry &e : ix.entries)
            names.insert(e.name);
        for (const FloatDataset &d : datasets)
            if (!names.insert(d.name).second)
                fail(path, ("dataset exists: " + d.name).c_str());
    }
    // The append stays recoverable: the bytes of the old index that the new datasets overwrite, and the three words of
    // the superblock / root header that finish() patches, are kept in memory until the new index is complete; any
    // failure in between (ENOSPC, a short write) puts them back, so the file is the valid file it was before the call.
    std::vector<uint8_t> old_tail(start < ix.eof ? (size_t)(ix.eof - start) : 0);
    if (!old_tail.empty())
        pread_exact(fc.f, path, start, old_tail.data(), old_tail.size());
    uint8_t old_msg[16], old_scratch[16], old_eof[8];
    pread_exact(fc.f, path, ix.root_msg_data_off, old_msg, 16);
    pread_exact(fc.f, path, 56 + 24, old_scratch, 16);
    pread_exact(fc.f, path, 40, old_eof, 8);
    try {
        Writer w{fc.f, path, start};
        for (const FloatDataset &d : datasets) {
            const uint64_t data_addr = w.pos;
            if (d.pitch_bytes == d.cols * sizeof(float)) {
                pwrite_exact(fc.f, path, data_addr, d.data, d.rows * d.cols * sizeof(float));
            } else {
                for (size_t r = 0; r < d.rows; ++r)
                    pwrite_exact(fc.f, path, data_addr + r * d.cols * sizeof(float),
                                 (const uint8_t *)d.data + r * d.pitch_bytes, d.cols * sizeof(float));
            }
            w.pos = (data_addr + d.rows * d.cols * sizeof(float) + 7) & ~7ull;
            const uint64_t hdr = w.put(dataset_header(d.rows, d.cols, data_addr));
            ix.entries.push_back(Entry{d.name, hdr});
        }
        std::sort(ix.entries.begin(), ix.entries.end(),
                  [](const Entry &a, const Entry &b) { return strcmp(a.name.c_str(), b.name.c_str()) < 0; });
        uint64_t btree, heap;
        write_group(w, ix.entries, ix.leaf_k, ix.internal_k, btree, heap);
        finish(w, ix.root_header, ix.root_msg_data_off, btree, heap);
    } catch (...) {
        // best effort, no further throw from here: if the medium is gone nothing can be put back anyway
        clearerr(fc.f);
        bool ok = true;
        auto put_back = [&](uint64_t off, const void *src, size_t n) {
            ok = ok && fseek(fc.f, (long)off, SEEK_SET) == 0 && fwrite(src, 1, n, fc.f) == n;
        };
        if (!old_tail.empty())
            put_back(start, old_tail.data(), old_tail.size());
        put_back(ix.root_msg_data_off, old_msg, 16);
        put_back(56 + 24, old_scratch, 16);
        put_back(40, old_eof, 8);
        (void)fflush(fc.f);
        throw;
    }
}

} // namespace h5mini
