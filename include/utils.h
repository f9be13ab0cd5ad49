// include/utils.h — small helpers with the reference's names (/root/reference/include/utils.h).
#ifndef DENSEFLOW_UTILS_H
#define DENSEFLOW_UTILS_H

#include <sys/stat.h>

#include <atomic>

#include "common.h"

double CurrentSeconds();          // wall clock, millisecond resolution
void createFile(const path &ph);  // touch

inline bool fileExists(const string &name) {
    struct stat info;
    return stat(name.c_str(), &info) == 0;
}
inline bool dirExists(const string &p) {
    struct stat info;
    return stat(p.c_str(), &info) == 0 && (info.st_mode & S_IFDIR);
}

// Run fn(0) .. fn(n-1) on up to `threads` workers (work is handed out one index at a time).  The first
// exception thrown by any fn is rethrown in the caller after all workers have stopped.
template <class F> void parallelFor(int n, int threads, F fn) {
    if (n <= 0)
        return;
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) {
        for (int i = 0; i < n; ++i)
            fn(i);
        return;
    }
    std::atomic<int> next(0);
    std::exception_ptr err;
    mutex err_mtx;
    auto work = [&] {
        try {
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1))
                fn(i);
        } catch (...) {
            unique_lock<mutex> lock(err_mtx);
            if (!err)
                err = std::current_exception();
            next.store(n); // let the other workers run dry
        }
    };
    vector<thread> pool;
    for (int t = 1; t < threads; ++t)
        pool.emplace_back(work);
    work();
    for (auto &t : pool)
        t.join();
    if (err)
        std::rethrow_exception(err);
}
#endif // DENSEFLOW_UTILS_H
