// src/image_io.cpp — decoder-free media I/O for the host shell: Y4M / PGM / PPM readers, bilinear
// resize, a baseline JPEG encoder (ITU-T T.81 Annex K tables) and a PNG writer (stored deflate).
// Host glue outside the hot path; nothing here runs on the GPU.
#include "image_io.h"

#include "dfx_jpeg_tables.h"

#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <cmath>

// ------------------------------------------------------------------------------------------------
// YUV4MPEG2

bool VideoCapture::open(const string &file) {
    FILE *fp = fopen(file.c_str(), "rb");
    if (!fp)
        return false;
    f_.reset(fp, fclose);
    char line[512];
    if (!fgets(line, sizeof line, fp) || strncmp(line, "YUV4MPEG2", 9) != 0) {
        f_.reset();
        return false;
    }
    string cs = "420";
    char *save = nullptr; // strtok_r: several loader threads (one per device) open clips concurrently
    for (char *tok = strtok_r(line + 9, " \n", &save); tok; tok = strtok_r(nullptr, " \n", &save)) {
        // strtol saturates where atoi is undefined; anything beyond the engine's limit is rejected below
        if (tok[0] == 'W')
            w_ = (int)std::min<long>(std::max<long>(strtol(tok + 1, nullptr, 10), -1), 1 << 20);
        else if (tok[0] == 'H')
            h_ = (int)std::min<long>(std::max<long>(strtol(tok + 1, nullptr, 10), -1), 1 << 20);
        else if (tok[0] == 'C')
            cs = tok + 1;
    }
    if (w_ <= 0 || h_ <= 0 || w_ > 32768 || h_ > 32768) { // the engine's own limit (dfx_create); also keeps w * h in range
        f_.reset();
        return false;
    }
    const size_t cw = (w_ + 1) / 2, ch = (h_ + 1) / 2;
    if (cs.rfind("mono", 0) == 0)
        chroma_bytes_ = 0;
    else if (cs.rfind("444", 0) == 0)
        chroma_bytes_ = 2 * (size_t)w_ * h_;
    else if (cs.rfind("422", 0) == 0)
        chroma_bytes_ = 2 * cw * h_;
    else
        chroma_bytes_ = 2 * cw * ch; // 420*
    // frame count from the file size (every frame is "FRAME\n" + planes when no frame parameters are used)
    const long here = ftell(fp);
    fseek(fp, 0, SEEK_END);
    const long end = ftell(fp);
    fseek(fp, here, SEEK_SET);
    const size_t per = 6 + (size_t)w_ * h_ + chroma_bytes_;
    frames_ = (int)((end - here) / (long)per);
    data_start_ = here;
    return true;
}

bool VideoCapture::seekFrame(int index) {
    if (!f_ || index < 0 || index > frames_)
        return false;
    const size_t per = 6 + (size_t)w_ * h_ + chroma_bytes_;
    return fseek(f_.get(), data_start_ + (long)((size_t)index * per), SEEK_SET) == 0;
}

bool VideoCapture::read(Mat &gray) {
    if (!f_)
        return false;
    FILE *fp = f_.get();
    char line[256];
    if (!fgets(line, sizeof line, fp) || strncmp(line, "FRAME", 5) != 0)
        return false;
    gray.create(Size(w_, h_), CV_8UC1);
    if (fread(gray.data(), 1, (size_t)w_ * h_, fp) != (size_t)w_ * h_)
        return false;
    if (chroma_bytes_)
        fseek(fp, (long)chroma_bytes_, SEEK_CUR);
    return true;
}

// ------------------------------------------------------------------------------------------------
// PGM / PPM

static bool pnm_token(FILE *fp, int &v) {
    int c = fgetc(fp);
    for (;;) {
        while (c == ' ' || c == '\n' || c == '\r' || c == '\t')
            c = fgetc(fp);
        if (c == '#') {
            while (c != '\n' && c != EOF)
                c = fgetc(fp);
            continue;
        }
        break;
    }
    if (c < '0' || c > '9')
        return false;
    v = 0;
    while (c >= '0' && c <= '9') {
        if (v > (1 << 24)) // no header field of a usable image is that large; keeps v * 10 in range
            return false;
        v = v * 10 + (c - '0');
        c = fgetc(fp);
    }
    return true; // exactly one whitespace byte after the token has been consumed
}

bool imreadGray(const string &file, Mat &gray) {
    FILE *fp = fopen(file.c_str(), "rb");
    if (!fp)
        return false;
    std::shared_ptr<FILE> guard(fp, fclose);
    char magic[3] = {0, 0, 0};
    if (fread(magic, 1, 2, fp) != 2 || magic[0] != 'P' || (magic[1] != '5' && magic[1] != '6'))
        return false;
    int w, h, maxv;
    if (!pnm_token(fp, w) || !pnm_token(fp, h) || !pnm_token(fp, maxv) || maxv != 255 || w <= 0 || h <= 0 || w > 32768 ||
        h > 32768)
        return false;
    gray.create(Size(w, h), CV_8UC1);
    if (magic[1] == '5')
        return fread(gray.data(), 1, (size_t)w * h, fp) == (size_t)w * h;
    vector<uchar> rgb((size_t)w * h * 3);
    if (fread(rgb.data(), 1, rgb.size(), fp) != rgb.size())
        return false;
    uchar *g = gray.data();
    for (size_t i = 0; i < (size_t)w * h; ++i) // cvtColor BGR2GRAY, 8-bit: 15-bit weights R 9798, G 19235, B 3735
        g[i] = (uchar)((rgb[3 * i] * 9798 + rgb[3 * i + 1] * 19235 + rgb[3 * i + 2] * 3735 + (1 << 14)) >> 15);
    return true;
}

// cv::resize(src, dst, size) for CV_8UC1 with the default INTER_LINEAR, in OpenCV's 8-bit integer arithmetic
// (the same arithmetic as the device path, denseflow_amd/csrc/prepare_kernels.hip): source coordinate
// (float)((d + 0.5)*scale - 0.5); 11-bit fixed-point weights; along x an out-of-range index is clamped and its
// fraction zeroed, along y the two row indices are clamped; an exact 2x2 decimation is a rounded 2x2 mean
// (cv::resize executes that INTER_LINEAR request as INTER_AREA).
void resizeLinear(const Mat &src, Mat &dst, Size size) {
    dst.create(size, CV_8UC1);
    const int sw = src.cols, sh = src.rows, dw = size.width, dh = size.height;
    if (sw == dw && sh == dh) {
        for (int y = 0; y < dh; ++y)
            std::memcpy(dst.ptr<uchar>(y), src.ptr<uchar>(y), dw);
        return;
    }
    if (sw == 2 * dw && sh == 2 * dh) {
        for (int y = 0; y < dh; ++y) {
            const uchar *r0 = src.ptr<uchar>(2 * y), *r1 = src.ptr<uchar>(2 * y + 1);
            uchar *d = dst.ptr<uchar>(y);
            for (int x = 0; x < dw; ++x)
                d[x] = (uchar)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
        }
        return;
    }
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    vector<int> xo(dw), a0(dw), a1(dw);
    for (int x = 0; x < dw; ++x) {
        float f = (float)((x + 0.5) * scale_x - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (s < 0)
            f = 0.f, s = 0;
        if (s >= sw - 1)
            f = 0.f, s = sw - 1;
        xo[x] = s;
        a0[x] = (int)std::lrintf((1.f - f) * 2048.f);
        a1[x] = (int)std::lrintf(f * 2048.f);
    }
    for (int y = 0; y < dh; ++y) {
        float f = (float)((y + 0.5) * scale_y - 0.5);
        const int s = (int)std::floor(f);
        f -= (float)s;
        const int b0 = (int)std::lrintf((1.f - f) * 2048.f), b1 = (int)std::lrintf(f * 2048.f);
        const uchar *r0 = src.ptr<uchar>(std::min(std::max(s, 0), sh - 1));
        const uchar *r1 = src.ptr<uchar>(std::min(std::max(s + 1, 0), sh - 1));
        uchar *d = dst.ptr<uchar>(y);
        for (int x = 0; x < dw; ++x) {
            const int x0 = xo[x], x1 = std::min(x0 + 1, sw - 1);
            const int h0 = r0[x0] * a0[x] + r0[x1] * a1[x], h1 = r1[x0] * a0[x] + r1[x1] * a1[x];
            d[x] = (uchar)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// baseline JPEG encoder, 8-bit gray (T.81): standard luminance quantisation and Huffman tables

namespace {

// tables shared with the device encoder (include/dfx_jpeg_tables.h): both encoders must produce the same bytes
const unsigned char (&kZigzag)[64] = kDfxJpegZigzag;
const unsigned char (&kDcBits)[17] = kDfxJpegDcBits;
const unsigned char (&kDcVal)[12] = kDfxJpegDcVal;
const unsigned char (&kAcBits)[17] = kDfxJpegAcBits;
const unsigned char (&kAcVal)[162] = kDfxJpegAcVal;

struct HuffTable {
    unsigned short code[256];
    uchar len[256];
    void build(const uchar *bits, const uchar *vals) {
        memset(len, 0, sizeof len);
        unsigned code_v = 0;
        int k = 0;
        for (int l = 1; l <= 16; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k) {
                code[vals[k]] = (unsigned short)code_v++;
                len[vals[k]] = (uchar)l;
            }
            code_v <<= 1;
        }
    }
};

// Entropy-coded segment writer: 64-bit accumulator, bytes appended to a raw buffer that the caller sized
// for the worst case; 0xFF bytes get their stuffed zero as they leave the accumulator.
struct BitWriter {
    uchar *p;
    unsigned long long acc = 0;
    int nbits = 0;
    explicit BitWriter(uchar *dst) : p(dst) {}
    inline void put(unsigned code, int len) { // len <= 27
        acc = (acc << len) | (code & ((1u << len) - 1));
        nbits += len;
        while (nbits >= 8) {
            const uchar b = (uchar)(acc >> (nbits - 8));
            *p++ = b;
            if (b == 0xFF)
                *p++ = 0;
            nbits -= 8;
        }
    }
    void flush() {
        if (nbits > 0)
            put((1u << (8 - nbits)) - 1, 8 - nbits);
    }
};

void put16(vector<uchar> &o, int v) {
    o.push_back((uchar)(v >> 8));
    o.push_back((uchar)v);
}

// Forward DCT of one 8x8 block + quantisation: libjpeg's JDCT_ISLOW transform and its quantisation rule
// (include/dfx_jpeg_tables.h), i.e. the arithmetic behind cv::imencode(".jpg").  coef[v*8+u] in natural order.  Integer
// arithmetic, so the two forms below — scalar, and eight lines at a time on AVX2 registers where the CPU has them —
// agree by construction.  Ragged right / bottom edge: the last column / row is replicated, as libjpeg's
// edge expansion does (jcprepct.c).
struct QuantTable {
    unsigned div[64], magic[64]; // 8 q and its reciprocal (dfx_jpeg_divide_magic), natural order
};

void fdct_quant_scalar(const uchar *src, size_t pitch, int valid_w, int valid_h, const QuantTable &Q, int *coef) {
    int d[8][8];
    for (int y = 0; y < 8; ++y) {
        const uchar *row = src + (size_t)std::min(y, valid_h - 1) * pitch;
        for (int x = 0; x < 8; ++x)
            d[y][x] = (int)row[std::min(x, valid_w - 1)] - 128;
    }
    for (int y = 0; y < 8; ++y)
        dfx_jpeg_fdct_islow_1d<true>(d[y][0], d[y][1], d[y][2], d[y][3], d[y][4], d[y][5], d[y][6], d[y][7]);
    for (int x = 0; x < 8; ++x)
        dfx_jpeg_fdct_islow_1d<false>(d[0][x], d[1][x], d[2][x], d[3][x], d[4][x], d[5][x], d[6][x], d[7][x]);
    for (int i = 0; i < 64; ++i)
        coef[i] = dfx_jpeg_quantise(d[i >> 3][i & 7], Q.div[i], Q.magic[i]);
}

#if defined(__x86_64__)
typedef int v8i __attribute__((vector_size(32)));

__attribute__((target("avx2"))) inline void transpose8(v8i v[8]) {
    __m256 r[8];
    for (int i = 0; i < 8; ++i)
        r[i] = _mm256_castsi256_ps((__m256i)v[i]);
    __m256 t0 = _mm256_unpacklo_ps(r[0], r[1]), t1 = _mm256_unpackhi_ps(r[0], r[1]);
    __m256 t2 = _mm256_unpacklo_ps(r[2], r[3]), t3 = _mm256_unpackhi_ps(r[2], r[3]);
    __m256 t4 = _mm256_unpacklo_ps(r[4], r[5]), t5 = _mm256_unpackhi_ps(r[4], r[5]);
    __m256 t6 = _mm256_unpacklo_ps(r[6], r[7]), t7 = _mm256_unpackhi_ps(r[6], r[7]);
    __m256 s0 = _mm256_shuffle_ps(t0, t2, 0x44), s1 = _mm256_shuffle_ps(t0, t2, 0xEE);
    __m256 s2 = _mm256_shuffle_ps(t1, t3, 0x44), s3 = _mm256_shuffle_ps(t1, t3, 0xEE);
    __m256 s4 = _mm256_shuffle_ps(t4, t6, 0x44), s5 = _mm256_shuffle_ps(t4, t6, 0xEE);
    __m256 s6 = _mm256_shuffle_ps(t5, t7, 0x44), s7 = _mm256_shuffle_ps(t5, t7, 0xEE);
    r[0] = _mm256_permute2f128_ps(s0, s4, 0x20), r[1] = _mm256_permute2f128_ps(s1, s5, 0x20);
    r[2] = _mm256_permute2f128_ps(s2, s6, 0x20), r[3] = _mm256_permute2f128_ps(s3, s7, 0x20);
    r[4] = _mm256_permute2f128_ps(s0, s4, 0x31), r[5] = _mm256_permute2f128_ps(s1, s5, 0x31);
    r[6] = _mm256_permute2f128_ps(s2, s6, 0x31), r[7] = _mm256_permute2f128_ps(s3, s7, 0x31);
    for (int i = 0; i < 8; ++i)
        v[i] = (v8i)_mm256_castps_si256(r[i]);
}

// eight lines at a time: the same integer operations as fdct_quant_scalar on AVX2 registers
__attribute__((target("avx2"))) void fdct_quant_avx2(const uchar *src, size_t pitch, const QuantTable &Q, int *coef) {
    v8i r[8]; // r[y] = row y of the block (lane x)
    const __m256i bias = _mm256_set1_epi32(128);
    for (int y = 0; y < 8; ++y)
        r[y] = (v8i)_mm256_sub_epi32(
            _mm256_cvtepu8_epi32(_mm_loadl_epi64(reinterpret_cast<const __m128i *>(src + (size_t)y * pitch))), bias);
    transpose8(r); // r[x] = column x (lane y): pass 1 transforms all eight rows at once
    dfx_jpeg_fdct_islow_1d<true>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    transpose8(r); // r[y] = row y (lane u): pass 2 transforms all eight columns at once
    dfx_jpeg_fdct_islow_1d<false>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    for (int v = 0; v < 8; ++v) { // dfx_jpeg_quantise on eight coefficients: |c| + d/2, high half of the product with magic, sign
        const __m256i c = (__m256i)r[v];
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(Q.div + 8 * v));
        const __m256i m = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(Q.magic + 8 * v));
        const __m256i n = _mm256_add_epi32(_mm256_abs_epi32(c), _mm256_srli_epi32(d, 1));
        const __m256i even = _mm256_srli_epi64(_mm256_mul_epu32(n, m), 32);                     // lanes 0 2 4 6
        const __m256i odd = _mm256_mul_epu32(_mm256_srli_epi64(n, 32), _mm256_srli_epi64(m, 32)); // high halves: lanes 1 3 5 7
        const __m256i q = _mm256_blend_epi32(even, odd, 0xAA);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(coef + 8 * v), _mm256_sign_epi32(q, c));
    }
}
const bool g_has_avx2 = __builtin_cpu_supports("avx2");
#endif
bool g_force_portable = false;

inline void fdct_quant(const uchar *src, size_t pitch, int valid_w, int valid_h, const QuantTable &Q, int *coef) {
#if defined(__x86_64__)
    if (g_has_avx2 && !g_force_portable && valid_w >= 8 && valid_h >= 8) {
        fdct_quant_avx2(src, pitch, Q, coef);
        return;
    }
#endif
    fdct_quant_scalar(src, pitch, valid_w, valid_h, Q, coef);
}

inline int bit_length(int a) { return a ? 32 - __builtin_clz((unsigned)a) : 0; }

} // namespace

void imencodeJpegForcePortable(bool on) { g_force_portable = on; }

bool imencodeJpeg(const Mat &gray, vector<uchar> &out, int quality) {
    if (gray.empty() || gray.type() != CV_8UC1)
        return false;
    uchar q[64];
    QuantTable Q;
    dfx_jpeg_quantiser(quality, q);
    for (int i = 0; i < 64; ++i) {
        Q.div[i] = 8u * q[i];
        Q.magic[i] = dfx_jpeg_divide_magic(Q.div[i]);
    }
    struct Tables {
        HuffTable dc, ac;
        uchar nat2zig[64];  // position of natural-order coefficient i in the zig-zag scan
        Tables() {
            dc.build(kDcBits, kDcVal);
            ac.build(kAcBits, kAcVal);
            for (int k = 0; k < 64; ++k)
                nat2zig[kZigzag[k]] = (uchar)k;
        }
    };
    static const Tables T; // thread-safe one-time initialisation: encoders run in parallel
    const HuffTable &dc = T.dc, &ac = T.ac;
    const int W = gray.cols, H = gray.rows;
    out.clear();
    const uchar soi_app0[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    out.insert(out.end(), soi_app0, soi_app0 + sizeof soi_app0);
    out.push_back(0xFF), out.push_back(0xDB), put16(out, 67), out.push_back(0);
    for (int i = 0; i < 64; ++i)
        out.push_back(q[kZigzag[i]]);
    out.push_back(0xFF), out.push_back(0xC0), put16(out, 11), out.push_back(8), put16(out, H), put16(out, W);
    out.push_back(1), out.push_back(1), out.push_back(0x11), out.push_back(0);
    out.push_back(0xFF), out.push_back(0xC4), put16(out, 2 + 1 + 16 + 12), out.push_back(0x00);
    out.insert(out.end(), kDcBits + 1, kDcBits + 17), out.insert(out.end(), kDcVal, kDcVal + 12);
    out.push_back(0xFF), out.push_back(0xC4), put16(out, 2 + 1 + 16 + 162), out.push_back(0x10);
    out.insert(out.end(), kAcBits + 1, kAcBits + 17), out.insert(out.end(), kAcVal, kAcVal + 162);
    const uchar sos[] = {0xFF, 0xDA, 0, 8, 1, 1, 0x00, 0, 63, 0};
    out.insert(out.end(), sos, sos + sizeof sos);

    // worst case per block: 64 coefficients x (16-bit code + 11 value bits) = 216 bytes, doubled by stuffing
    const size_t blocks = (size_t)((W + 7) / 8) * ((H + 7) / 8);
    // per-thread scratch for the entropy-coded segment (kept across calls: no page faults, no zero fill)
    static thread_local vector<uchar> scratch;
    if (scratch.size() < blocks * 432 + 16)
        scratch.resize(blocks * 432 + 16);
    BitWriter bw(scratch.data());
    int prev_dc = 0;
    for (int by = 0; by < H; by += 8) {
        for (int bx = 0; bx < W; bx += 8) {
            alignas(32) int coef[64];
            fdct_quant(gray.ptr<uchar>(by) + bx, gray.step, W - bx, H - by, Q, coef);
            // DC difference
            const int diff = coef[0] - prev_dc;
            prev_dc = coef[0];
            const int nb = bit_length(diff < 0 ? -diff : diff);
            bw.put(dc.code[nb], dc.len[nb]);
            if (nb)
                bw.put((unsigned)(diff < 0 ? diff - 1 : diff), nb);
            // AC run lengths over the zig-zag scan: visit only the non-zero coefficients
            unsigned long long nz = 0; // bit k: zig-zag position k holds a non-zero coefficient
            for (int i = 1; i < 64; ++i)
                nz |= (unsigned long long)(coef[i] != 0) << T.nat2zig[i];
            int last = 0;
            while (nz) {
                const int k = __builtin_ctzll(nz);
                nz &= nz - 1;
                int run = k - last - 1;
                last = k;
                while (run > 15) {
                    bw.put(ac.code[0xF0], ac.len[0xF0]);
                    run -= 16;
                }
                const int v = coef[kZigzag[k]];
                const int n = bit_length(v < 0 ? -v : v);
                const int sym = (run << 4) | n;
                bw.put(((unsigned)ac.code[sym] << n) | ((unsigned)(v < 0 ? v - 1 : v) & ((1u << n) - 1)), ac.len[sym] + n);
            }
            if (last != 63)
                bw.put(ac.code[0x00], ac.len[0x00]);
        }
    }
    bw.flush();
    out.insert(out.end(), scratch.data(), bw.p);
    out.push_back(0xFF), out.push_back(0xD9);
    return true;
}

// ------------------------------------------------------------------------------------------------
// PNG writer: 8-bit gray or BGR (stored as RGB) — the file cv::imencode(".png") writes (reference src/common.cpp:70).
// OpenCV's PngEncoder (modules/imgcodecs/src/grfmt_png.cpp, 4.5.2 as the reference's Dockerfile pins it) drives libpng
// with: every scanline filtered with SUB, zlib level Z_BEST_SPEED, strategy Z_RLE (IMWRITE_PNG_STRATEGY_RLE, its
// default), no interlace, png_set_bgr.  libpng 1.6 (pngwutil.c) adds: windowBits shrunk for images of <= 16384 filtered
// bytes (png_deflate_claim), memLevel 8, the zlib header's window field minimised afterwards (optimize_cmf), and the
// compressed stream cut into IDAT chunks of its 8192-byte buffer.  libpng is a third-party dependency that is not in
// /root/reference; this restates those published rules on top of the same zlib, and tests/test_png_libpng_pin.py holds
// it to the bytes the system's libpng16 writes under exactly those calls.

namespace {
void put32(vector<uchar> &o, unsigned v) {
    o.push_back((uchar)(v >> 24)), o.push_back((uchar)(v >> 16)), o.push_back((uchar)(v >> 8)), o.push_back((uchar)v);
}
void chunk(vector<uchar> &o, const char *type, const uchar *data, size_t n) {
    put32(o, (unsigned)n);
    const size_t start = o.size();
    o.insert(o.end(), type, type + 4);
    o.insert(o.end(), data, data + n);
    // zlib's CRC-32 (the PNG chunk CRC); a lazily built local table here was a data race between parallel encoders
    put32(o, (unsigned)crc32(0L, o.data() + start, (uInt)(o.size() - start)));
}
} // namespace

bool imencodePng(const Mat &img, vector<uchar> &out) {
    if (img.empty() || (img.type() != CV_8UC1 && img.type() != CV_8UC3))
        return false;
    const int ch = img.channels(), W = img.cols, H = img.rows;
    const size_t rb = (size_t)W * ch;
    // scanlines in PNG order (RGB), each SUB-filtered: byte i minus the byte one pixel to the left.  libpng drops SUB for
    // images one pixel wide (png_write_start_row): the same bytes, labelled NONE
    vector<uchar> raw((size_t)H * (rb + 1));
    for (int y = 0; y < H; ++y) {
        const uchar *r = img.ptr<uchar>(y);
        uchar *dst = raw.data() + (size_t)y * (rb + 1);
        *dst++ = W == 1 ? 0 : 1;
        if (ch == 1) {
            dst[0] = r[0];
            for (size_t i = 1; i < rb; ++i)
                dst[i] = (uchar)(r[i] - r[i - 1]);
        } else {
            for (int x = 0; x < W; ++x)
                for (int k = 0; k < 3; ++k) // output channel k = input channel 2 - k (png_set_bgr)
                    dst[3 * x + k] = (uchar)(r[3 * x + 2 - k] - (x ? r[3 * x - 1 - k] : 0));
        }
    }
    // libpng's png_deflate_claim: the window only as large as the data needs (images of <= 16384 filtered bytes)
    int window_bits = 15;
    if (raw.size() <= 16384) {
        unsigned half_window = 1u << (window_bits - 1);
        while (raw.size() + 262 <= half_window) {
            half_window >>= 1;
            --window_bits;
        }
    }
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, Z_BEST_SPEED, Z_DEFLATED, window_bits, 8, Z_RLE) != Z_OK)
        return false;
    vector<uchar> z(deflateBound(&zs, (uLong)raw.size()) + 64);
    zs.next_in = raw.data(), zs.avail_in = (uInt)raw.size();
    zs.next_out = z.data(), zs.avail_out = (uInt)z.size();
    const int rc = deflate(&zs, Z_FINISH);
    const size_t zlen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END)
        return false;
    // libpng's optimize_cmf: the smallest window field the data size allows, header check bits redone
    if (raw.size() <= 16384 && (z[0] & 0x0f) == 8 && (z[0] & 0xf0) <= 0x70) {
        unsigned cinfo = z[0] >> 4, half_window = 1u << (cinfo + 7);
        if (raw.size() <= half_window) {
            do {
                half_window >>= 1;
                --cinfo;
            } while (cinfo > 0 && raw.size() <= half_window);
            z[0] = (uchar)((z[0] & 0x0f) | (cinfo << 4));
            unsigned t = z[1] & 0xe0;
            t += 0x1f - ((z[0] << 8) + t) % 0x1f;
            z[1] = (uchar)t;
        }
    }
    out.clear();
    const uchar sig[] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.insert(out.end(), sig, sig + 8);
    vector<uchar> ihdr;
    put32(ihdr, (unsigned)W), put32(ihdr, (unsigned)H);
    ihdr.push_back(8), ihdr.push_back(ch == 1 ? 0 : 2), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    for (size_t off = 0; off < zlen; off += 8192) // libpng's zbuffer: one IDAT per 8192 bytes of compressed stream
        chunk(out, "IDAT", z.data() + off, std::min<size_t>(8192, zlen - off));
    chunk(out, "IEND", nullptr, 0);
    return true;
}
