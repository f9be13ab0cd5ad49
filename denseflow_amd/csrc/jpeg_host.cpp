// jpeg_host.cpp — host half of the device JPEG encoder: tables, file header, byte stuffing.  Mirrors imencodeJpeg of the
// host shell (src/image_io.cpp) marker for marker so that both write the same files; the constants are shared
// (include/dfx_jpeg_tables.h).  Reference: imencode(".jpg") in encodeFlowMap, /root/reference/src/common.cpp:56-57.
#include <cstring>

#include "../../include/dfx_jpeg_tables.h"
#include "jpeg_kernels.h"

namespace {

void build_huff(const unsigned char *bits, const unsigned char *vals, unsigned short *code, unsigned char *len, int n) {
    std::memset(len, 0, (size_t)n);
    std::memset(code, 0, sizeof(unsigned short) * (size_t)n);
    unsigned code_v = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l]; ++i, ++k) {
            if (vals[k] < n) {
                code[vals[k]] = (unsigned short)code_v;
                len[vals[k]] = (unsigned char)l;
            }
            ++code_v;
        }
        code_v <<= 1;
    }
}

void put16(std::vector<unsigned char> &o, int v) {
    o.push_back((unsigned char)(v >> 8));
    o.push_back((unsigned char)v);
}

} // namespace

void jpeg_build_tables(int quality, JpegTables &t, unsigned char q_out[64]) {
    dfx_jpeg_quantiser(quality, q_out);
    for (int i = 0; i < 64; ++i) {
        t.div[i] = 8u * q_out[i];
        t.magic[i] = dfx_jpeg_divide_magic(t.div[i]);
    }
    build_huff(kDfxJpegDcBits, kDfxJpegDcVal, t.dc_code, t.dc_len, 12);
    build_huff(kDfxJpegAcBits, kDfxJpegAcVal, t.ac_code, t.ac_len, 256);
    for (int k = 0; k < 64; ++k)
        t.nat2zig[kDfxJpegZigzag[k]] = (unsigned char)k;
}

std::vector<unsigned char> jpeg_file_header(int w, int h, const unsigned char q[64]) {
    std::vector<unsigned char> out;
    const unsigned char soi_app0[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    out.insert(out.end(), soi_app0, soi_app0 + sizeof soi_app0);
    out.push_back(0xFF), out.push_back(0xDB), put16(out, 67), out.push_back(0);
    for (int i = 0; i < 64; ++i)
        out.push_back(q[kDfxJpegZigzag[i]]);
    out.push_back(0xFF), out.push_back(0xC0), put16(out, 11), out.push_back(8), put16(out, h), put16(out, w);
    out.push_back(1), out.push_back(1), out.push_back(0x11), out.push_back(0);
    out.push_back(0xFF), out.push_back(0xC4), put16(out, 2 + 1 + 16 + 12), out.push_back(0x00);
    out.insert(out.end(), kDfxJpegDcBits + 1, kDfxJpegDcBits + 17), out.insert(out.end(), kDfxJpegDcVal, kDfxJpegDcVal + 12);
    out.push_back(0xFF), out.push_back(0xC4), put16(out, 2 + 1 + 16 + 162), out.push_back(0x10);
    out.insert(out.end(), kDfxJpegAcBits + 1, kDfxJpegAcBits + 17), out.insert(out.end(), kDfxJpegAcVal, kDfxJpegAcVal + 162);
    const unsigned char sos[] = {0xFF, 0xDA, 0, 8, 1, 1, 0x00, 0, 63, 0};
    out.insert(out.end(), sos, sos + sizeof sos);
    return out;
}

size_t jpeg_assemble(const std::vector<unsigned char> &header, const unsigned char *src, unsigned long long bits,
                     unsigned char *dst, size_t capacity) {
    const size_t full = (size_t)(bits >> 3);
    const unsigned rem = (unsigned)(bits & 7);
    if (header.size() + 2 * (full + 1) + 2 > capacity) { // cheap bound first; exact check only when it fails
        size_t ff = 0;
        for (size_t i = 0; i < full; ++i)
            ff += src[i] == 0xFF;
        const size_t tail = rem ? ((unsigned char)(src[full] | ((1u << (8 - rem)) - 1u)) == 0xFF ? 2 : 1) : 0;
        if (header.size() + full + ff + tail + 2 > capacity)
            return 0;
    }
    unsigned char *p = dst;
    std::memcpy(p, header.data(), header.size());
    p += header.size();
    // byte stuffing: every 0xFF of the entropy-coded segment is followed by 0x00 (T.81 B.1.1.5); runs without one are
    // copied in one piece
    size_t i = 0;
    while (i < full) {
        const void *hit = std::memchr(src + i, 0xFF, full - i);
        const size_t run = hit ? (size_t)((const unsigned char *)hit - (src + i)) + 1 : full - i;
        std::memcpy(p, src + i, run);
        p += run;
        i += run;
        if (hit)
            *p++ = 0;
    }
    if (rem) { // the last, partial byte is padded with one bits (BitWriter::flush of the host encoder)
        const unsigned char b = (unsigned char)(src[full] | ((1u << (8 - rem)) - 1u));
        *p++ = b;
        if (b == 0xFF)
            *p++ = 0;
    }
    *p++ = 0xFF;
    *p++ = 0xD9;
    return (size_t)(p - dst);
}
