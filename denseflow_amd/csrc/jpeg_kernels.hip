// jpeg_kernels.hip — baseline JPEG (T.81, 8-bit gray, Huffman) of the bounded flow planes, on the device.
//
// Replaces the two `imencode(".jpg", ...)` per flow of the reference's encodeFlowMap (/root/reference/src/common.cpp:
// 56-57), which its single save thread runs for every pair (src/denseflow_gpu.cpp:396-454).  With the flow bounding
// already on the device (quantize_kernels.hip) the planes never have to leave it uncompressed: the entropy-coded
// segments do — ~0.1 of a plane's bytes for flow images — and the host only adds the file header and the 0xFF byte
// stuffing.  Output is byte-identical to libjpeg(-turbo)'s — the library behind cv::imencode — and to the shell's host
// encoder (src/image_io.cpp: imencodeJpeg): libjpeg's JDCT_ISLOW integer transform and quantisation rule, shared with
// the host encoder through include/dfx_jpeg_tables.h, the Annex K tables, the same code construction
// (tests/test_jpeg_libjpeg_pin.py, tests/test_jpeg_gpu.py).
//
// Structure (everything on the batch's compute stream):
//   1. k_jpeg_blocks<false>  one thread per 8x8 block: DCT + quantisation in registers, the block's AC bit count and its
//                            quantised DC to memory;
//   2. k_jpeg_scan           one workgroup per plane: DC differences (the only coupling between blocks), exclusive
//                            prefix sum of the blocks' bit counts -> bit offset of every block, bits of the plane;
//   3. k_jpeg_layout         one thread: byte offset of every plane's stream in the shared buffer, totals to the host
//                            (mapped page-locked memory);
//   4. zero-fill of the used part of the shared buffer, then
//      k_jpeg_blocks<true>   the same DCT again (a few hundred integer operations per block are cheaper than keeping
//                            128 bytes of coefficients per block in HBM), codes appended at the block's bit offset with
//                            32-bit atomic ORs
//                            (big-endian words: JPEG bit order is MSB first).
// One thread per block is deliberate: a plane has 32 400 blocks (1080p) and a batch 258 planes — millions of
// independent blocks, so there is no need to split a block over lanes, and the per-block code is the host encoder's
// loop line for line.  Cost: ~1 ms per 129-pair batch at 1080p, against ~300 ms of TVL1.
#include <hip/hip_runtime.h>

#include "../../include/dfx_jpeg_tables.h"
#include "jpeg_kernels.h"

namespace {

__device__ __forceinline__ int bit_length(int a) { return a ? 32 - __builtin_clz((unsigned)a) : 0; }

struct Emitter { // MSB-first bit string appended at an arbitrary bit position of a zeroed big-endian word stream
    unsigned *words;
    unsigned long long acc; // pending bits, left-aligned
    int nacc;               // number of pending bits (< 32 between puts)
    __device__ __forceinline__ void begin(unsigned *stream, unsigned long long bitpos) {
        words = stream + (bitpos >> 5);
        nacc = (int)(bitpos & 31); // the first word is shared with the previous block: its leading bits stay zero here
        acc = 0;
    }
    __device__ __forceinline__ void put(unsigned code, int len) { // len <= 27
        acc |= (unsigned long long)(code & ((1u << len) - 1u)) << (64 - nacc - len);
        nacc += len;
        if (nacc >= 32) {
            atomicOr(words, __builtin_bswap32((unsigned)(acc >> 32)));
            ++words;
            acc <<= 32;
            nacc -= 32;
        }
    }
    __device__ __forceinline__ void end() {
        if (nacc > 0)
            atomicOr(words, __builtin_bswap32((unsigned)(acc >> 32)));
    }
};

template <bool EMIT>
__global__ __launch_bounds__(64) void k_jpeg_blocks(JpegCtx c) {
    __shared__ short zz[64][64]; // [zig-zag position][lane]: the entropy loop indexes coefficients dynamically
    const int lane = threadIdx.x;
    const int plane = blockIdx.y;
    const int nblk = c.bw * c.bh;
    const int blk = blockIdx.x * 64 + lane;
    if (EMIT && c.hdr[1] != 0) // the streams do not fit the shared buffer (k_jpeg_layout): nothing is written
        return;
    if (blk >= nblk)
        return;
    const JpegTables &T = *c.tab;
    const int by = blk / c.bw, bx = blk - by * c.bw;
    const unsigned char *P =
        c.planes + (long long)(plane < c.n_x ? plane : c.y_first + plane - c.n_x) * c.plane_stride;
    // ---- load (ragged right / bottom edge: replicate the last column / row, like libjpeg's edge expansion and
    //      fdct_quant_scalar of the host encoder)
    int a[8][8];
    const int x0 = bx * 8, y0 = by * 8;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const unsigned char *row = P + (long long)min(y0 + y, c.h - 1) * c.pitch;
#pragma unroll
        for (int x = 0; x < 8; ++x)
            a[y][x] = (int)row[min(x0 + x, c.w - 1)] - 128;
    }
    // ---- forward DCT: libjpeg's JDCT_ISLOW, rows then columns (include/dfx_jpeg_tables.h) — integer arithmetic, the
    //      host encoder's and libjpeg's own coefficients
#pragma unroll
    for (int y = 0; y < 8; ++y)
        dfx_jpeg_fdct_islow_1d<true>(a[y][0], a[y][1], a[y][2], a[y][3], a[y][4], a[y][5], a[y][6], a[y][7]);
#pragma unroll
    for (int x = 0; x < 8; ++x)
        dfx_jpeg_fdct_islow_1d<false>(a[0][x], a[1][x], a[2][x], a[3][x], a[4][x], a[5][x], a[6][x], a[7][x]);
    int dc = 0;
    unsigned long long nz = 0; // bit k: zig-zag position k holds a non-zero coefficient
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = dfx_jpeg_quantise(a[v][u], T.div[v * 8 + u], T.magic[v * 8 + u]); // divide by 8 q, half away from 0
            const int k = T.nat2zig[v * 8 + u];
            if (v == 0 && u == 0) {
                dc = q;
            } else {
                zz[k][lane] = (short)q;
                nz |= (unsigned long long)(q != 0) << k;
            }
        }
    const long long bi = (long long)plane * nblk + blk;
    if (!EMIT) {
        // ---- count: AC codes of this block (the DC difference needs the neighbour's DC: k_jpeg_scan adds it)
        unsigned bits = 0;
        int last = 0;
        while (nz) {
            const int k = __builtin_ctzll(nz);
            nz &= nz - 1;
            const int run = k - last - 1;
            last = k;
            const int vq = zz[k][lane];
            const int n = bit_length(vq < 0 ? -vq : vq);
            bits += (unsigned)(run >> 4) * T.ac_len[0xF0] + T.ac_len[((run & 15) << 4) | n] + n;
        }
        if (last != 63)
            bits += T.ac_len[0x00];
        c.dc[bi] = (short)dc;
        c.bits[bi] = bits;
        return;
    }
    // ---- emit: the host encoder's block loop (imencodeJpeg), appended at this block's bit offset
    Emitter E;
    E.begin(c.stream, c.plane_base[plane] * 8ull + c.bits[bi]);
    const int prev_dc = blk > 0 ? (int)c.dc[bi - 1] : 0;
    const int diff = dc - prev_dc;
    const int nb = bit_length(diff < 0 ? -diff : diff);
    E.put(T.dc_code[nb], T.dc_len[nb]);
    if (nb)
        E.put((unsigned)(diff < 0 ? diff - 1 : diff), nb);
    int last = 0;
    while (nz) {
        const int k = __builtin_ctzll(nz);
        nz &= nz - 1;
        int run = k - last - 1;
        last = k;
        while (run > 15) {
            E.put(T.ac_code[0xF0], T.ac_len[0xF0]);
            run -= 16;
        }
        const int vq = zz[k][lane];
        const int n = bit_length(vq < 0 ? -vq : vq);
        const int sym = (run << 4) | n;
        E.put(((unsigned)T.ac_code[sym] << n) | ((unsigned)(vq < 0 ? vq - 1 : vq) & ((1u << n) - 1u)), T.ac_len[sym] + n);
    }
    if (last != 63)
        E.put(T.ac_code[0x00], T.ac_len[0x00]);
    E.end();
}

// One workgroup per plane: bits of every block (DC difference code + AC codes) and their exclusive prefix sum.
__global__ __launch_bounds__(1024) void k_jpeg_scan(JpegCtx c) {
    __shared__ unsigned wave_sum[16];
    __shared__ unsigned long long running;
    const int plane = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = c.bw * c.bh;
    const JpegTables &T = *c.tab;
    const short *dc = c.dc + (long long)plane * nblk;
    unsigned *bits = c.bits + (long long)plane * nblk;
    if (tid == 0)
        running = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + tid;
        unsigned v = 0;
        if (b < nblk) {
            const int diff = (int)dc[b] - (b > 0 ? (int)dc[b - 1] : 0);
            const int nb = bit_length(diff < 0 ? -diff : diff);
            v = (unsigned)T.dc_len[nb] + (unsigned)nb + bits[b];
        }
        unsigned incl = v; // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += o;
        }
        if (lane == 63)
            wave_sum[wave] = incl;
        __syncthreads();
        unsigned before = 0;
        for (int i = 0; i < wave; ++i)
            before += wave_sum[i];
        const unsigned long long base = running;
        if (b < nblk) {
            const unsigned long long off = base + before + incl - v;
            bits[b] = (unsigned)off; // < 2^32: a plane's segment is at most blocks x 1728 bits
        }
        __syncthreads();
        if (tid == 1023)
            running = base + before + incl;
        __syncthreads();
    }
    if (tid == 0)
        c.plane_bits[plane] = running;
}

// One thread: where every plane's stream starts (4-byte aligned: the emit pass ORs whole words), totals for the host.
__global__ void k_jpeg_layout(JpegCtx c) {
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    const int n = c.n_planes;
    unsigned long long at = 0;
    for (int p = 0; p < n; ++p) {
        const unsigned long long bits = c.plane_bits[p];
        c.plane_base[p] = at;
        c.info[2 + 2 * p] = bits;
        c.info[2 + 2 * p + 1] = at;
        at += ((bits + 31) >> 5) << 2;
    }
    c.hdr[0] = c.info[0] = at;
    c.hdr[1] = c.info[1] = at > c.capacity_bytes ? 1ull : 0ull;
    __threadfence_system();
}

// Zero the part of the shared buffer the emit pass will OR into.
__global__ __launch_bounds__(256) void k_jpeg_zero(JpegCtx c) {
    if (c.hdr[1] != 0)
        return;
    const unsigned long long words = c.hdr[0] >> 2;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * 256)
        c.stream[i] = 0u;
}

} // namespace

void jpeg_launch_encode(hipStream_t s, const JpegCtx &c) {
    const int nblk = c.bw * c.bh, n_planes = c.n_planes;
    const dim3 grid((nblk + 63) / 64, n_planes);
    hipLaunchKernelGGL(k_jpeg_blocks<false>, grid, dim3(64), 0, s, c);
    hipLaunchKernelGGL(k_jpeg_scan, dim3(n_planes), dim3(1024), 0, s, c);
    hipLaunchKernelGGL(k_jpeg_layout, dim3(1), dim3(1), 0, s, c);
    hipLaunchKernelGGL(k_jpeg_zero, dim3(1024), dim3(256), 0, s, c);
    hipLaunchKernelGGL(k_jpeg_blocks<true>, grid, dim3(64), 0, s, c);
}
