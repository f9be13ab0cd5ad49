/*
 * include/dfx_jpeg_tables.h — the constants of the baseline JPEG encoder (ITU-T T.81, 8-bit gray), shared by the host
 * encoder of the shell (src/image_io.cpp: imencodeJpeg) and the device encoder of libdfx
 * (denseflow_amd/csrc/jpeg_kernels.hip: dfx_calc_batch_jpeg).  Both must produce the same bytes, so both use the SAME
 * tables and the SAME transform, which is integer arithmetic throughout.
 *
 * Replaces (together with the two encoders) the reference's `imencode(".jpg", ...)` of every bounded flow plane
 * (/root/reference/src/common.cpp:56-57).  OpenCV's imencode drives libjpeg(-turbo) with its defaults: baseline
 * sequential DCT, one component, the Annex K luminance quantiser scaled for quality 95, the Annex K luminance Huffman
 * tables, and libjpeg's default forward transform JDCT_ISLOW with its quantisation rule.  libjpeg is a third-party
 * dependency of the reference that is not in /root/reference; its published algorithm (IJG jfdctint.c: the
 * Loeffler-Ligtenberg-Moschytz 8-point DCT in 13-bit fixed point, two passes; jcdctmgr.c: divide by 8 q, round half away
 * from zero) is restated below.  PINNED: a libjpeg-turbo is importable in this container through Pillow, and with this
 * transform both encoders write files that are byte-identical to libjpeg-turbo's for the same plane and quality
 * (tests/test_jpeg_libjpeg_pin.py, live against Pillow where it is installed, and against tests/golden/
 * jpeg_libjpeg_golden.npz everywhere).
 */
#ifndef DFX_JPEG_TABLES_H
#define DFX_JPEG_TABLES_H

#if defined(__HIPCC__)
#define DFX_JPEG_HD __host__ __device__
#else
#define DFX_JPEG_HD
#endif

/* zig-zag scan (position -> natural index), Annex K.1 luminance quantiser, Annex K.3 luminance Huffman tables */
static const unsigned char kDfxJpegZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                           41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                           30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static const unsigned char kDfxJpegLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                          14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                          18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                          49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const unsigned char kDfxJpegDcBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const unsigned char kDfxJpegDcVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const unsigned char kDfxJpegAcBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const unsigned char kDfxJpegAcVal[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

/* quantiser for a quality setting, libjpeg's scaling (what cv::imencode's IMWRITE_JPEG_QUALITY means) */
static inline void dfx_jpeg_quantiser(int quality, unsigned char q[64]) {
    int i, scale;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (i = 0; i < 64; ++i) {
        const int v = (kDfxJpegLumaQ[i] * scale + 50) / 100;
        q[i] = (unsigned char)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

#ifdef __cplusplus
/* One pass of libjpeg's JDCT_ISLOW forward DCT (IJG jfdctint.c) over one line of eight values, for any integer type I
 * with + - * << >> (int on the device and in the portable host form, a vector of eight ints in the host's SIMD form —
 * integer arithmetic: every form gives the same coefficients).  FIRST = true: pass 1 (libjpeg: rows), results scaled
 * up by 2^PASS1; FIRST = false: pass 2 (columns), the scaling removed again.  After both passes the block holds 8x the
 * DCT coefficients; the factor is part of the divisor (dfx_jpeg_quantise).  CONST = 13 bits: products stay below 2^31
 * for 8-bit samples (jfdctint.c's own range analysis). */
template <bool FIRST, class I>
DFX_JPEG_HD inline void dfx_jpeg_fdct_islow_1d(I &d0, I &d1, I &d2, I &d3, I &d4, I &d5, I &d6, I &d7) {
    const int CONST = 13, PASS1 = 2;
    const int S = FIRST ? CONST - PASS1 : CONST + PASS1; /* descale shift of the odd part and of outputs 2 and 6 */
    const int R = 1 << (S - 1);
    I t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6, t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    const I t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    if (FIRST) { /* x 2^PASS1 (libjpeg's LEFT_SHIFT; a multiplication because << of a negative int is undefined before C++20) */
        d0 = (t10 + t11) * (1 << PASS1);
        d4 = (t10 - t11) * (1 << PASS1);
    } else {
        d0 = (t10 + t11 + (1 << (PASS1 - 1))) >> PASS1;
        d4 = (t10 - t11 + (1 << (PASS1 - 1))) >> PASS1;
    }
    I z1 = (t12 + t13) * 4433;               /* FIX(0.541196100) */
    d2 = (z1 + t13 * 6270 + R) >> S;         /* FIX(0.765366865) */
    d6 = (z1 - t12 * 15137 + R) >> S;        /* FIX(1.847759065) */
    z1 = t4 + t7;
    I z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const I z5 = (z3 + z4) * 9633;           /* FIX(1.175875602) */
    t4 = t4 * 2446;                          /* FIX(0.298631336) */
    t5 = t5 * 16819;                         /* FIX(2.053119869) */
    t6 = t6 * 25172;                         /* FIX(3.072711026) */
    t7 = t7 * 12299;                         /* FIX(1.501321110) */
    z1 = z1 * -7373;                         /* FIX(0.899976223) */
    z2 = z2 * -20995;                        /* FIX(2.562915447) */
    z3 = z3 * -16069 + z5;                   /* FIX(1.961570560) */
    z4 = z4 * -3196 + z5;                    /* FIX(0.390180644) */
    d7 = (t4 + z1 + z3 + R) >> S;
    d5 = (t5 + z2 + z4 + R) >> S;
    d3 = (t6 + z2 + z3 + R) >> S;
    d1 = (t7 + z1 + z4 + R) >> S;
}

/* libjpeg's quantisation of one coefficient (jcdctmgr.c, forward_DCT): the divisor is 8 q (the transform's scale), the
 * quotient is rounded half away from zero.  `magic` = dfx_jpeg_divide_magic(8 q): the quotient as the high half of a
 * 32 x 32 product, exact for every numerator this encoder can produce (numerator * divisor < 2^32; checked
 * exhaustively in tests/test_jpeg_libjpeg_pin.py). */
DFX_JPEG_HD inline unsigned dfx_jpeg_divide_magic(unsigned divisor) { return (unsigned)(0x100000000ull / divisor) + 1u; }
DFX_JPEG_HD inline int dfx_jpeg_quantise(int coef8, unsigned divisor, unsigned magic) {
    const unsigned n = (unsigned)(coef8 < 0 ? -coef8 : coef8) + (divisor >> 1);
    const int q = (int)(((unsigned long long)n * magic) >> 32);
    return coef8 < 0 ? -q : q;
}
#endif

#endif /* DFX_JPEG_TABLES_H */
