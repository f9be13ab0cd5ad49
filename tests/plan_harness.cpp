// tests/plan_harness.cpp — TEST INFRASTRUCTURE: C wrappers around denseflow_amd/csrc/dfx_plan.h (the pure host logic of
// a FlowBuffer's batching, the header dfx_api.cpp compiles) so that tests/test_plan_logic.py can drive it on the CPU.
#include "../denseflow_amd/csrc/dfx_plan.h"
#include "../denseflow_amd/csrc/farneback_plan.h"

extern "C" {

// pairs of a FlowBuffer: returns M; a[i], b[i] (the reference's "previous" and "next" frame of flow i) up to cap entries
int ph_pairs(const int *seg, int n_seg, int step, int *a, int *b, int cap) {
    const DfxPairs p = dfx_build_pairs(std::vector<int>(seg, seg + n_seg), step);
    for (int i = 0; i < p.size() && i < cap; ++i) {
        a[i] = dfx_pair_a(p, i, step);
        b[i] = dfx_pair_b(p, i, step);
    }
    return p.size();
}

int ph_frames_needed(const int *seg, int n_seg, int step, int batch) {
    return dfx_frames_needed(dfx_build_pairs(std::vector<int>(seg, seg + n_seg), step), batch);
}

// batches: returns their number; row k of out = {i0, nb, first_new, n_new}
int ph_plan(const int *seg, int n_seg, int step, int batch, long long *out, int cap) {
    const std::vector<DfxBatchPlan> plan = dfx_plan_batches(dfx_build_pairs(std::vector<int>(seg, seg + n_seg), step), batch);
    for (size_t k = 0; k < plan.size() && (int)k < cap; ++k) {
        out[4 * k] = plan[k].i0, out[4 * k + 1] = plan[k].nb, out[4 * k + 2] = plan[k].first_new, out[4 * k + 3] = plan[k].n_new;
    }
    return (int)plan.size();
}

// rows per segment of the Farneback row-stream kernel's column strips
int ph_farn_seg_rows(int w, int h, int n_pairs) { return farn_stream_seg_rows(w, h, n_pairs); }
}
