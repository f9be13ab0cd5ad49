// dfx_plan.h — the host-side plan of one FlowBuffer: which frame pairs it holds, how they are cut into device batches and
// which frames every batch has to bring in.  Pure C++ (no HIP): compiled into dfx_api.cpp and, for the CPU suite, into
// tests/plan_harness.cpp (tests/test_plan_logic.py checks it against the reference's pair rule and its invariants).
//
// Reference: DenseFlow::calc_optflows_imp, /root/reference/src/denseflow_gpu.cpp:307-316 — for a FlowBuffer of N frames
// and step s, M = max(N - |s|, 0) flows; flow i is (frame i, frame i + s) for s > 0 and (frame i - s, frame i) otherwise.
// Several clips joined into one FlowBuffer (dfx_next_segments) apply that rule inside every clip.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <vector>

struct DfxPairs {
    std::vector<int> lo, hi; // the two frames of pair i, frame ids counted over the whole FlowBuffer, lo + |step| == hi
    int size() const { return (int)lo.size(); }
};

// seg: frames per clip, in order.  Both signs of the step select the same frame pairs (the sign only says which of the
// two is the "previous" image: dfx_pair_a / dfx_pair_b).
inline DfxPairs dfx_build_pairs(const std::vector<int> &seg, int step) {
    DfxPairs p;
    const int astep = std::abs(step);
    int off = 0;
    for (int n : seg) {
        for (int i = 0; i + astep < n; ++i) {
            p.lo.push_back(off + i);
            p.hi.push_back(off + i + astep);
        }
        off += n;
    }
    return p;
}
inline int dfx_pair_a(const DfxPairs &p, int i, int step) { return step > 0 ? p.lo[i] : p.hi[i]; }
inline int dfx_pair_b(const DfxPairs &p, int i, int step) { return step > 0 ? p.hi[i] : p.lo[i]; }

// Frames one batch of at most `batch` consecutive pairs can need: batch + |step| inside one clip, |step| more for every
// clip boundary it spans.  The engines keep frame id f in slot f % F, so F >= this keeps a batch's frames apart.
inline int dfx_frames_needed(const DfxPairs &p, int batch) {
    int need = 0;
    const int M = p.size();
    for (int i0 = 0; i0 < M; i0 += batch)
        need = std::max(need, p.hi[std::min(i0 + batch, M) - 1] - p.lo[i0] + 1);
    return need;
}

struct DfxBatchPlan {
    int i0, nb;          // pairs [i0, i0 + nb)
    long long first_new; // first frame id that has to be prepared for this batch
    int n_new;           // number of such frames (consecutive ids)
};

// Frames [lo of its first pair, hi of its last pair] must be resident for a batch; earlier batches already prepared the
// ids below their own end, so every frame is prepared exactly once, in increasing order.
inline std::vector<DfxBatchPlan> dfx_plan_batches(const DfxPairs &p, int batch) {
    std::vector<DfxBatchPlan> plan;
    const int M = p.size();
    long long built = 0;
    for (int i0 = 0; i0 < M; i0 += batch) {
        DfxBatchPlan b;
        b.i0 = i0;
        b.nb = std::min(batch, M - i0);
        const long long need_end = (long long)p.hi[i0 + b.nb - 1] + 1;
        b.first_new = std::max<long long>(built, p.lo[i0]);
        b.n_new = (int)std::max<long long>(need_end - b.first_new, 0);
        built = std::max(built, need_end);
        plan.push_back(b);
    }
    return plan;
}
