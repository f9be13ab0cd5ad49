// egress_kernels.h — the device-to-host leg of the FlowBuffer driver as a kernel of this library (egress_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

// One 2-D copy: `rows` rows of `row_bytes` bytes from device memory (src, spitch) into device-accessible host memory
// (dst, dpitch).  Everything a multiple of 16 bytes (egress_item_ok).
struct EgressItem {
    void *dst;
    const void *src;
    unsigned long long dpitch, spitch, row_bytes;
    unsigned int rows;
    unsigned int first_unit; // filled by egress_launch: index of the item's first work unit
};

bool egress_item_ok(const EgressItem &it);
// items: n items in memory the DEVICE can read (page-locked host memory is fine); enqueues one launch of `workgroups`
// persistent workgroups on stream s.  Returns false when there is nothing to copy.
bool egress_launch(hipStream_t s, EgressItem *items, int n, int workgroups);
