// tests/helper_tsan.cpp — TEST INFRASTRUCTURE: denseflow_amd/csrc/dfx_helper.h under ThreadSanitizer on the CPU.
// The pattern of calc_batch_body's host-pointer path with bounce buffers and device JPEG over many batches: the owning
// thread "drives batch k" (writes its own data) while the helper hands batch k-1 over (reads what the owner published
// before start(), writes the caller's buffers) and gathers batch k+1; finish() before the owner reads the results or
// reuses a buffer; a failing job's status is the one finish() reports; destruction with a job in flight.
#include <cstdio>
#include <numeric>
#include <vector>

#include "../denseflow_amd/csrc/dfx_helper.h"

int main() {
    long long total = 0;
    int bad = 0;
    for (int round = 0; round < 50; ++round) {
        const int batches = 3 + round % 9;
        std::vector<std::vector<int>> bounce(2, std::vector<int>(4096)), caller(batches, std::vector<int>(4096));
        std::vector<int> staged(4096);
        DfxHelper helper; // declared after what its jobs touch: its destructor (finish + join) runs first
        for (int k = 0; k < batches; ++k) {
            if (k >= 1) { // hand batch k-1 over and gather batch k+1 beside this thread
                helper.start([&, k] {
                    caller[k - 1] = bounce[(k - 1) & 1];                       // scatter(k - 1)
                    std::iota(staged.begin(), staged.end(), (k + 1) * 1000);   // upload(k + 1): gather
                    return (round == 7 && k == 2) ? 42 : 0;                    // one failing job
                });
            }
            std::vector<int> mine(4096);
            std::iota(mine.begin(), mine.end(), k); // "the device computes batch k": the owner's own work
            const int rc = helper.finish();         // batch k-1 is in the caller's buffers
            if (rc != ((round == 7 && k == 2) ? 42 : 0))
                ++bad;
            bounce[k & 1] = mine;                   // "download(k)" lands in the bounce buffer of this parity
            if (helper.finish() != 0)               // idempotent, and the status was consumed
                ++bad;
        }
        helper.start([&] {
            caller[batches - 1] = bounce[(batches - 1) & 1];
            return 0;
        });
        if (round & 1) { // every other round: let the destructor finish the last job
            if (helper.finish() != 0)
                ++bad;
            for (int k = 0; k < batches; ++k)
                total += caller[k][0] + caller[k][4095];
        }
    }
    std::printf("bad %d total %lld\n", bad, total);
    return bad ? 1 : 0;
}
