// jpeg_kernels.h — baseline JPEG encoding of bounded flow planes on the device (jpeg_kernels.hip) and the host-side
// assembly of the files (jpeg_host.cpp).  SURVEY.md §8f-1: replaces the two imencode(".jpg") per flow of encodeFlowMap
// (/root/reference/src/common.cpp:56-57).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

struct JpegTables { // one per (handle, quality), in device memory
    unsigned div[64];   // 8 q: the divisor of libjpeg's quantisation (include/dfx_jpeg_tables.h), natural order
    unsigned magic[64]; // dfx_jpeg_divide_magic(div)
    unsigned short dc_code[12];
    unsigned char dc_len[12];
    unsigned short ac_code[256];
    unsigned char ac_len[256];
    unsigned char nat2zig[64]; // zig-zag position of natural-order coefficient i
};

// What the device reports to the host after a batch (page-locked, mapped: written by k_jpeg_layout with system-scope
// stores, read by the host after the stream is synchronised).
struct JpegInfo {
    unsigned long long total_bytes; // bytes of the batch's streams in the shared buffer (each plane 4-byte aligned)
    unsigned long long overflow;    // != 0: the streams did not fit the shared buffer; nothing was written
    // followed by n_planes x { unsigned long long bits, base_byte }
};

struct JpegCtx {
    const unsigned char *planes; // plane p: planes + (p < n_x ? p : y_first + p - n_x) * plane_stride
    long long plane_stride;
    int pitch, w, h, bw, bh;
    int n_planes, n_x, y_first;   // planes 0..n_x-1 are consecutive, the other n_planes - n_x start at index y_first
    const JpegTables *tab;
    short *dc;                    // [n_planes][bw * bh] quantised DC of every block
    unsigned *bits;               // [n_planes][bw * bh] AC bits of every block, then its exclusive bit offset
    unsigned long long *plane_bits; // [n_planes] bits of each plane's entropy-coded segment (before stuffing)
    unsigned long long *plane_base; // [n_planes] first byte of each plane's stream in `stream`
    unsigned *stream;             // shared output buffer, zeroed before the emit pass
    unsigned long long capacity_bytes;
    unsigned long long *info;     // JpegInfo + per-plane entries (device view of the page-locked block); written once
    unsigned long long *hdr;      // device copy of {total_bytes, overflow} for the zero-fill and emit passes
};

// Enqueue the whole encode of a batch on stream s: count pass, scans, layout, zero-fill, emit pass.
void jpeg_launch_encode(hipStream_t s, const JpegCtx &c);

// ---- host side (jpeg_host.cpp) ---------------------------------------------------------------------------------------
void jpeg_build_tables(int quality, JpegTables &t, unsigned char q_out[64]);
// The file header up to and including SOS for a w x h gray image with quantiser q (what imencodeJpeg writes).
std::vector<unsigned char> jpeg_file_header(int w, int h, const unsigned char q[64]);
// header + byte-stuffed entropy-coded segment (`bits` bits at src, last byte padded with ones) + EOI into dst.
// Returns the file size, or 0 when it would exceed `capacity`.
size_t jpeg_assemble(const std::vector<unsigned char> &header, const unsigned char *src, unsigned long long bits,
                     unsigned char *dst, size_t capacity);
