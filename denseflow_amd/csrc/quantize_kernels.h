// quantize_kernels.h — launcher of the flow bounding kernel (quantize_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

// n dense W*H*2 float flows (flow i at d_flows + i*flow_stride floats) -> n x-planes and n y-planes of
// 8-bit values (plane i at d_img_* + i*img_stride bytes, img_pitch bytes per row).
void quant_launch_flow_to_u8(hipStream_t s, const float *d_flows, long long flow_stride, int n, int w, int h,
                             double lo, double hi, unsigned char *d_img_x, unsigned char *d_img_y,
                             long long img_pitch, long long img_stride);

// The -st=png scheme (convertFlowToPngImage, /root/reference/src/common.cpp:18-46) for n dense flows: per-flow extrema,
// the reference's adaptive bounds (d_bounds[2 * i] = {bound_x, bound_y}, doubles) and the two convertTo(CV_8U) planes.
// d_scratch: quant_png_scratch_bytes(n) bytes of device memory (extrema keys + scale factors).  Four launches.
size_t quant_png_scratch_bytes(int n);
void quant_launch_flow_to_png_planes(hipStream_t s, const float *d_flows, long long flow_stride, int n, int w, int h,
                                     void *d_scratch, double *d_bounds, unsigned char *d_img_x, unsigned char *d_img_y,
                                     long long img_pitch, long long img_stride);
