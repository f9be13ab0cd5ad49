// prepare_kernels.hip — frame preparation on the device.
//
// Replaces the two per-frame host steps of DenseFlow::load_frames_batch
// (/root/reference/src/denseflow_gpu.cpp:146-177): cvtColor(frame, gray, COLOR_BGR2GRAY) (:163) and
// cv::resize(gray, resized, size) with the default INTER_LINEAR (:169), so the loader thread only reads
// bytes and source-size frames cross PCIe once.  The arithmetic is OpenCV's 8-bit integer arithmetic
// (imgproc 4.5.2, not in the reference repository — parity unpinned, see oracle/prepare_oracle.h):
//   * BGR2GRAY: (B*3735 + G*19235 + R*9798 + 2^14) >> 15;
//   * resize INTER_LINEAR: source coordinate (float)((d + 0.5)*scale - 0.5), weights rounded to 11-bit
//     fixed point (x2048), horizontal pass in int, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16),
//     +2 >> 2; an exact 2x2 decimation is what cv::resize turns into INTER_AREA: (a+b+c+d+2) >> 2.
// One thread per destination pixel; the four taps of a pixel are byte gathers served by L2 (a frame is
// read once per destination pixel, 1-3 B/px of source + 1 B/px written: HBM-bound, tiny next to the flow).
#include "prepare_kernels.h"

namespace {

__device__ __forceinline__ int gray_at(const unsigned char *row, int x, int channels) {
    if (channels == 1)
        return row[x];
    const unsigned char *p = row + 3 * x; // B, G, R
    return (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
}

// cv::resize's coefficient tables (resize.cpp, INTER_LINEAR, 8-bit): the source coordinate is evaluated in
// double and rounded to float, the weights are rounded to 11-bit fixed point (saturate_cast<short> = cvRound).
// The two axes treat the borders differently, as upstream does: along x an out-of-range index is clamped AND
// its fraction zeroed when the table is built; along y the table keeps the fraction and the row loop clamps
// the two row indices.
__device__ __forceinline__ void linear_coeff_x(int d, double scale, int ssize, int &s, int &w0, int &w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) {
        f = 0.f;
        s = 0;
    }
    if (s >= ssize - 1) {
        f = 0.f;
        s = ssize - 1;
    }
    w0 = (int)rintf((1.f - f) * 2048.f); // exact products: 11-bit scale
    w1 = (int)rintf(f * 2048.f);
}
__device__ __forceinline__ void linear_coeff_y(int d, double scale, int ssize, int &s0, int &s1, int &w0, int &w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    s0 = min(max(s, 0), ssize - 1);
    s1 = min(max(s + 1, 0), ssize - 1);
    w0 = (int)rintf((1.f - f) * 2048.f);
    w1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void k_prepare_frames(const unsigned char *src, long long src_pitch,
                                                         long long src_frame_stride, int sw, int sh, int channels,
                                                         unsigned char *dst, long long dst_pitch,
                                                         long long dst_frame_stride, int dw, int dh, double scale_x,
                                                         double scale_y, int mode) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh)
        return;
    const unsigned char *S = src + (long long)blockIdx.z * src_frame_stride;
    unsigned char *D = dst + (long long)blockIdx.z * dst_frame_stride;
    int out;
    if (mode == 0) { // same size: colour conversion only
        out = gray_at(S + (long long)dy * src_pitch, dx, channels);
    } else if (mode == 1) { // exact 2x decimation: INTER_AREA fast path
        const unsigned char *r0 = S + (long long)(2 * dy) * src_pitch, *r1 = r0 + src_pitch;
        out = (gray_at(r0, 2 * dx, channels) + gray_at(r0, 2 * dx + 1, channels) + gray_at(r1, 2 * dx, channels) +
               gray_at(r1, 2 * dx + 1, channels) + 2) >> 2;
    } else {
        int sx, a0, a1, sy0, sy1, b0, b1;
        linear_coeff_x(dx, scale_x, sw, sx, a0, a1);
        linear_coeff_y(dy, scale_y, sh, sy0, sy1, b0, b1);
        const int sx1 = min(sx + 1, sw - 1); // weight 0 whenever this clamps
        const unsigned char *r0 = S + (long long)sy0 * src_pitch, *r1 = S + (long long)sy1 * src_pitch;
        const int h0 = gray_at(r0, sx, channels) * a0 + gray_at(r0, sx1, channels) * a1; // HResizeLinear, int
        const int h1 = gray_at(r1, sx, channels) * a0 + gray_at(r1, sx1, channels) * a1;
        out = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2; // VResizeLinear<uchar, int, short>
    }
    D[(long long)dy * dst_pitch + dx] = (unsigned char)out;
}

} // namespace

int prepare_mode(int sw, int sh, int dw, int dh) {
    if (sw == dw && sh == dh)
        return 0;
    if (sw == 2 * dw && sh == 2 * dh)
        return 1;
    return 2;
}

void prepare_launch(hipStream_t s, const unsigned char *d_src, long long src_pitch, long long src_frame_stride, int sw,
                    int sh, int channels, int n, unsigned char *d_dst, long long dst_pitch, long long dst_frame_stride,
                    int dw, int dh) {
    if (n <= 0)
        return;
    // cv::resize: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
    const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4, n);
    hipLaunchKernelGGL(k_prepare_frames, grid, dim3(256), 0, s, d_src, src_pitch, src_frame_stride, sw, sh, channels,
                       d_dst, dst_pitch, dst_frame_stride, dw, dh, scale_x, scale_y, prepare_mode(sw, sh, dw, dh));
}
