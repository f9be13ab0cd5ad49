// egress_kernels.hip — flows / bounded planes leave the device through a kernel of a FEW persistent workgroups.
//
// Replaces, on the PCIe-inclusive path, the blocking `flow_gpu.download(flows[i])` of the reference
// (/root/reference/src/denseflow_gpu.cpp:339).  Why not hipMemcpyAsync: on this stack a device-to-host copy into
// page-locked memory is executed by a shader blit (`__amd_rocclr_copyBuffer`, seen in the kernel trace, not by the SDMA
// engine like the uploads), sized to fill the machine.  Its waves wait on PCIe (16.6 MB per 1080p flow at 55 GB/s =
// 300 us each) and meanwhile hold the wave slots the flow kernels of the NEXT batch need: in the Farneback float-out
// timeline every batch that computed beside a download of 129 flows took 109 ms instead of 80.5 ms
// (profiles/round3/farn_pcie_timeline.md) — 1410 instead of 1670 pairs/s, which round 2 mistook for the link limit
// (the link does 55 GB/s: profiles/round3/pcie_probe.txt).  Posted PCIe writes need no occupancy to reach that rate:
// a few dozen workgroups that walk the batch's rows saturate the link and leave the rest of the machine to compute.
//
// Work unit = up to 16 KB of one row (1024 x 16 bytes: four 16-byte stores per thread); units are dealt round-robin
// to the workgroups, so concurrent stores walk consecutive addresses of the same flow.
#include "egress_kernels.h"

#include <algorithm>
#include <cstdint>

namespace {

constexpr unsigned UNIT_VEC = 1024; // uint4 per unit

__device__ __forceinline__ unsigned units_per_row(unsigned long long row_bytes) {
    return (unsigned)((row_bytes / 16 + UNIT_VEC - 1) / UNIT_VEC);
}

__global__ __launch_bounds__(256) void k_egress(const EgressItem *items, int n_items, unsigned total_units) {
    __shared__ EgressItem it;
    __shared__ unsigned cur_first, cur_end;
    if (threadIdx.x == 0) {
        cur_first = 1;
        cur_end = 0; // empty range: the first unit loads its item
    }
    __syncthreads();
    for (unsigned u = blockIdx.x; u < total_units; u += gridDim.x) {
        if (u < cur_first || u >= cur_end) { // uniform: another item (units of an item are consecutive)
            __syncthreads();
            if (threadIdx.x == 0) {
                int lo = 0, hi = n_items - 1;
                while (lo < hi) { // last item whose first_unit <= u
                    const int mid = (lo + hi + 1) >> 1;
                    if (items[mid].first_unit <= u)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                it = items[lo];
                cur_first = it.first_unit;
                cur_end = it.first_unit + it.rows * units_per_row(it.row_bytes);
            }
            __syncthreads();
        }
        const unsigned upr = units_per_row(it.row_bytes);
        const unsigned local = u - it.first_unit;
        const unsigned row = local / upr, seg = local - row * upr;
        const unsigned long long nvec = it.row_bytes / 16;
        const uint4 *src = reinterpret_cast<const uint4 *>((const char *)it.src + (unsigned long long)row * it.spitch);
        uint4 *dst = reinterpret_cast<uint4 *>((char *)it.dst + (unsigned long long)row * it.dpitch);
        const unsigned long long v0 = (unsigned long long)seg * UNIT_VEC;
        uint4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long v = v0 + threadIdx.x + 256ull * k;
            if (v < nvec)
                t[k] = src[v];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long v = v0 + threadIdx.x + 256ull * k;
            if (v < nvec)
                dst[v] = t[k];
        }
    }
}

} // namespace

bool egress_item_ok(const EgressItem &it) {
    const unsigned long long m = (unsigned long long)(uintptr_t)it.dst | (unsigned long long)(uintptr_t)it.src | it.dpitch |
                                 it.spitch | it.row_bytes;
    return (m & 15ull) == 0 && it.rows > 0 && it.row_bytes > 0;
}

bool egress_launch(hipStream_t s, EgressItem *items, int n, int workgroups) {
    unsigned total = 0;
    for (int i = 0; i < n; ++i) {
        EgressItem &it = items[i];
        if (it.dpitch == it.row_bytes && it.spitch == it.row_bytes) { // dense on both sides: one long row
            it.row_bytes *= it.rows;
            it.dpitch = it.spitch = it.row_bytes;
            it.rows = 1;
        }
        it.first_unit = total;
        total += it.rows * (unsigned)((it.row_bytes / 16 + UNIT_VEC - 1) / UNIT_VEC);
    }
    if (total == 0)
        return false;
    const unsigned g = (unsigned)std::max<long long>(1, std::min<long long>(workgroups, total));
    hipLaunchKernelGGL(k_egress, dim3(g), dim3(256), 0, s, (const EgressItem *)items, n, total);
    return true;
}
