// dfx_device.h — data layout in HBM shared by the host control code and the HIP kernels.
//
// Layout (all planes are float32, row pitch padded to 64 floats = 256 B so every row starts on a
// cache-line boundary and 16-B vector accesses stay aligned):
//
//   frame slot f  : three pyramids  I, Ix, Iy     (Ix/Iy = centred gradient)
//                   level s lives at element offset lvl_off[s] of the slot, pitch lvl_pitch[s]
//   pair slot b   : NPLANES work planes sized for level 0, reused by every level with the
//                   level's own pitch:  u[2 sets][2], p[2 sets][4], I1wx, I1wy, grad, rho_c
//                   (ping-pong sets: a fused U+dual step reads one set and writes the other)
//
// A frame's pyramid is built once and serves as I1 of pair i and as I0 of pair i+step.
#pragma once

#include "tvl1_ctrl.h"

#define DFX_LVL_MAX 16

enum : int {
    PL_U1_0 = 0, PL_U2_0, PL_U1_1, PL_U2_1,                // u sets 0/1
    PL_P11_0, PL_P12_0, PL_P21_0, PL_P22_0,                // p set 0
    PL_P11_1, PL_P12_1, PL_P21_1, PL_P22_1,                // p set 1
    PL_I1WX, PL_I1WY, PL_GRAD, PL_RHOC,
    PL_COUNT
};

struct PairDesc {
    int frame_a; // frame slot of I0
    int frame_b; // frame slot of I1
};

struct Tvl1Consts {
    float l_t;   // (float)(lambda*theta)
    float taut;  // (float)(tau/theta)
    float theta; // (float)theta
    int hyp;     // hypot reading of the exact arithmetic (tvl1_math.h: TVL1_HYP_*) for the scalar kernel forms
};

// Everything a TVL1 kernel needs for one level; passed by value as the kernel argument.
struct Tvl1LevelCtx {
    // geometry of this level
    int w, h, pitch;
    // frame pyramids
    const float *frame_I;   // base of frame slot 0, pyramid I
    const float *frame_Ix;
    const float *frame_Iy;
    long long frame_stride; // elements between frame slots
    long long lvl_off;      // element offset of this level inside a frame slot
    // pair work planes
    float *planes;          // base of pair slot 0
    long long plane_stride; // elements between planes of one slot
    long long slot_stride;  // elements between pair slots
    // control
    Tvl1State *state;       // [n_pairs]
    const PairDesc *pairs;  // [n_pairs]
    double *partials;       // [n_pairs][partials_stride]
    int partials_stride;
    int n_pairs;
    Tvl1LoopCfg loop;
    Tvl1Consts k;
    double thr;             // scaledEpsilon of this level = eps^2 * (w*h)   (A.3)
    int level;              // level index (0 = full resolution)
    int *iters_out;         // [n_pairs][DFX_LVL_MAX][TVL1_MAX_WARPS] executed inner iterations
    int *checks_out;        // [n_pairs][DFX_LVL_MAX][2] convergence sums evaluated, steps that did work
    long long *work_out;    // [n_pairs][DFX_LVL_MAX][2] half-row updates per tile column: step kernel, warp-and-head kernel
    // completion signalling
    unsigned int *level_done_count; // device counter of pairs that finished the level
    volatile int *host_done_flag;   // pinned host word: set to done_token when every pair finished
    int done_token;
    int split_warp; // the backward warp runs as its own kernel in front of every step: the step kernel skips phase WARP
    int warp_lds;   // that kernel gathers through an LDS tile (the default; 0 with DFX_VAR_TVL1_WARP_GATHER)
    int head;       // the warp kernel also runs the head of the loop it starts (k_tvl1_warp_head).  The dual planes of a level's
                    // first warp are zero by definition (A.3): that kernel does not read them and k_tvl1_zero_planes does
                    // not write them
    int geom;       // tile geometry of the default step kernel: bit 0 = tile columns start at x = 0, bit 1 = halo as wide
                    // as the step is long (k_tvl1_step_fused; 0 = classic)
};

static inline int dfx_round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// XCD-aware workgroup -> tile mapping (device code only).  MI355X has 8 XCDs with a private 4 MiB L2 each and the
// dispatcher is observed to place workgroup b of a launch on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch"):
// with the plain blockIdx -> tile mapping the neighbours of a tile — whose halo, box-filter or gather footprint
// overlaps its own — run on seven OTHER XCDs and every L2 fetches the shared rows for itself.  dfx_block_xy() hands
// each XCD one contiguous run of the (x, y) tiles of a grid instead (row-major, bijective for any tile count; grid.z
// = pair / frame is left alone).  A speed choice only: nothing depends on where a workgroup runs.
// Tile index of the workgroup with dispatch index `lin` among `nt`: workgroups lin = k, k + 8, k + 16 ... (XCD k) get the
// contiguous tiles [k*q + min(k, r), ...) with q = nt / 8, r = nt % 8 — a bijection of [0, nt) for every nt (checked on
// the CPU: tests/test_ctrl_logic.py).  Plain C++ so that the test compiles the very function the kernels use.
DFX_HD int dfx_xcd_tile_index(int lin, int nt) {
    const int q = nt >> 3, rem = nt & 7, k = lin & 7;
    return k * q + (k < rem ? k : rem) + (lin >> 3);
}
#if defined(__HIPCC__)
#ifndef DFX_XCD_REMAP
#define DFX_XCD_REMAP 1 // 0: plain blockIdx (A/B builds, scripts/build_variant.sh)
#endif
struct DfxBlockXY {
    int x, y;
};
__device__ __forceinline__ DfxBlockXY dfx_block_xy() {
    DfxBlockXY r;
    r.x = (int)blockIdx.x;
    r.y = (int)blockIdx.y;
#if DFX_XCD_REMAP
    const int gx = (int)gridDim.x;
    const int t = dfx_xcd_tile_index(r.y * gx + r.x, gx * (int)gridDim.y);
    r.y = t / gx;
    r.x = t - r.y * gx;
#endif
    return r;
}
// the same for a grid whose x dimension already is a linear tile index
__device__ __forceinline__ int dfx_block_linear() {
#if DFX_XCD_REMAP
    return dfx_xcd_tile_index((int)blockIdx.x, (int)gridDim.x);
#else
    return (int)blockIdx.x;
#endif
}
#endif

