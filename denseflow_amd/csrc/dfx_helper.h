// dfx_helper.h — the ONE helper thread of a handle (no HIP in this header: tests/helper_tsan.cpp runs it under
// ThreadSanitizer on the CPU).
//
// The host-pointer path of calc_batch_body hands the host-side work of batch k-1 / k+1 (rows from the bounce buffer to the
// caller's buffers, JPEG files assembled, the next batch's frames gathered and sent up) to another thread while the
// calling thread drives batch k.  Rounds 3-4 created a std::thread per batch for that (ADVICE r3 / VERDICT r4 #8: ~20 us
// of clone + join per batch, and a thread id per batch in every trace); this is one persistent thread per handle, created
// on first use, that runs one job at a time:
//     start(fn)  waits for the previous job, then hands fn over and returns;
//     finish()   waits until the thread is idle and returns the last job's status (then DFX_OK until the next job).
// One producer only — the thread that owns the handle (the C ABI's threading rule); finish() may be called any number
// of times.  The destructor finishes the running job and joins.
#pragma once

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

class DfxHelper {
  public:
    DfxHelper() = default;
    DfxHelper(const DfxHelper &) = delete;
    DfxHelper &operator=(const DfxHelper &) = delete;
    ~DfxHelper() {
        {
            std::unique_lock<std::mutex> lock(m_);
            cv_.wait(lock, [&] { return !busy_; });
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable())
            th_.join();
    }

    void start(std::function<int()> fn) {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return !busy_; }); // one job at a time: the previous one first (its status stays for finish())
        if (!th_.joinable())
            th_ = std::thread([this] { run(); });
        job_ = std::move(fn);
        busy_ = true;
        lock.unlock();
        cv_.notify_all();
    }

    int finish() {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return !busy_; });
        const int r = rc_;
        rc_ = 0;
        return r;
    }

  private:
    void run() {
        std::unique_lock<std::mutex> lock(m_);
        for (;;) {
            cv_.wait(lock, [&] { return stop_ || (busy_ && job_); });
            if (stop_)
                return;
            std::function<int()> fn = std::move(job_);
            job_ = nullptr;
            lock.unlock();
            const int r = fn(); // the job may touch anything the producer handed over: published by the mutex above
            lock.lock();
            if (rc_ == 0) // the first failure since the last finish() is the one reported
                rc_ = r;
            busy_ = false;
            cv_.notify_all();
        }
    }

    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool busy_ = false, stop_ = false;
    int rc_ = 0;
};
