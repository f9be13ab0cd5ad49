// brox_kernels.h — device layout and launchers of the -a=brox kernels (brox_kernels.hip).
//
//   frame slot f : six pyramids (planes I, Dx I, Dy I, Dxx I, Dxy I, Dyy I), all levels; plane p of
//                  level l at element offset p*pyr_elems + lvl_off[l], pitch = round_up(w_l, 64).
//                  Built once per frame: as first frame of a pair only I, Dx, Dy are read.
//   pair slot b  : BROX_PL_COUNT planes sized for level 0: u,v (two sets: the level being solved and
//                  the level it was prolongated from), du, dv, gx, gy, inv_den_u, inv_den_v, num_dudv, num_u, num_v
#pragma once

#include <hip/hip_runtime.h>

#include "dfx_device.h"

enum : int { BROX_PL_U0 = 0, BROX_PL_V0, BROX_PL_U1, BROX_PL_V1, BROX_PL_DU, BROX_PL_DV, BROX_PL_GX, BROX_PL_GY,
             BROX_PL_IDU, BROX_PL_IDV, BROX_PL_NDUDV, BROX_PL_NU, BROX_PL_NV, BROX_PL_DU1, BROX_PL_DV1, BROX_PL_COUNT };
// du/dv exist twice (DU/DV = set 0, DU1/DV1 = set 1): the fused SOR kernel reads one set (with halo) and
// writes the other, because neighbouring workgroups of the same launch read each other's pixels.
enum : int { BROX_FP_I = 0, BROX_FP_DX, BROX_FP_DY, BROX_FP_DXX, BROX_FP_DXY, BROX_FP_DYY, BROX_FP_COUNT };

struct BroxLevelCtx {
    int w, h, pitch;
    long long lvl_off;        // element offset of this level inside one pyramid
    const float *frames;      // base of frame slot 0
    long long frame_stride;   // elements between frame slots
    long long pyr_elems;      // elements of one pyramid (all levels of one plane)
    float *planes;            // base of pair slot 0
    long long plane_stride, slot_stride;
    const PairDesc *pairs;
    int n_pairs;
    float alpha, gamma, omega;
    int sor_stream;           // fused SOR: persistent workgroups (this many: the device's CUs) that prefetch the next tile's
                              // coefficient planes by LDS-DMA while they sweep (0: one workgroup per tile, DFX_VAR_BROX_SOR_PER_TILE)
    float *sor_sink;          // streaming SOR: 16 floats per workgroup that take the stores of lanes owning nothing (see there)
    int sor_progress;         // fused SOR: band-wise progress counters (DFX_VAR_BROX_SOR_PROGRESS) instead of a barrier per half sweep
};

void brox_launch_u8_to_f32(hipStream_t s, const unsigned char *src, long long src_frame_stride, long long src_pitch,
                           const int *frame_slots, int n_frames, float *frames, long long frame_stride, int w, int h,
                           int pitch, float scale);
void brox_launch_downsample(hipStream_t s, float *frames, long long frame_stride, const int *frame_slots,
                            int n_frames, long long src_off, int sw, int sh, int spitch, long long dst_off, int dw,
                            int dh, int dpitch, float factor);
// dst plane = D_axis(src plane) for one level (axis 0 = x, 1 = y)
void brox_launch_deriv(hipStream_t s, float *frames, long long frame_stride, const int *frame_slots, int n_frames,
                       long long src_off, long long dst_off, int w, int h, int pitch, int axis);
void brox_launch_level_init(hipStream_t s, const BroxLevelCtx &c, int uv_set, int zero_uv); // zeroes du/dv set 0
void brox_launch_stage1(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_set);
void brox_launch_stage2(hipStream_t s, const BroxLevelCtx &c);
void brox_launch_sor(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_set, int color); // in place
// n_sweeps (<= brox_fused_sweeps()) full red+black sweeps in one launch (64 x 64 LDS tile, recomputed halo); writes
// set d_src ^ 1
void brox_launch_sor_fused(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_src, int n_sweeps);
int brox_fused_sweeps();
void brox_launch_add_increment(hipStream_t s, const BroxLevelCtx &c, int uv_set, int d_set);
// (u,v)[uv_set ^ 1] at the finer geometry = bicubic(u,v[uv_set]) * mul
void brox_launch_prolongate(hipStream_t s, const BroxLevelCtx &c_coarse, int uv_set, int dw, int dh, int dpitch,
                            float factor, float mul);
void brox_launch_merge(hipStream_t s, const BroxLevelCtx &c0, int uv_set, float *out, long long out_stride);
