// prepare_kernels.h — launcher of the frame-preparation kernel (prepare_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

// 0: same size (colour conversion only), 1: exact 2x decimation (INTER_AREA fast path), 2: INTER_LINEAR
int prepare_mode(int sw, int sh, int dw, int dh);

// n source frames (sw x sh, `channels` = 1 gray or 3 BGR interleaved) -> n gray frames of dw x dh.
void prepare_launch(hipStream_t s, const unsigned char *d_src, long long src_pitch, long long src_frame_stride, int sw,
                    int sh, int channels, int n, unsigned char *d_dst, long long dst_pitch, long long dst_frame_stride,
                    int dw, int dh);
