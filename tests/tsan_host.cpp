// tests/tsan_host.cpp — ThreadSanitizer driver for the host shell pieces that run on several threads at once
// (bounded queue, parallelFor, concurrent clip opens, parallel JPEG/PNG encoders, the pinned-buffer pool).
// Built and run by tests/test_host_tsan.py; needs no GPU (nothing here calls into the device library).
#include "../include/dense_flow.h"
#include "../include/utils.h"
#include <atomic>
#include <cstdio>
int main(int argc, char **argv) {
    // queue
    FlowBufferQueue q(3);
    thread prod([&] { for (int i = 0; i < 500; ++i) q.push(FlowBuffer({}, path(), i, false), i == 499); });
    long sum = 0; while (true) { bool fin = false; FlowBuffer b = q.pop(&fin); sum += b.base_start; if (fin) break; }
    prod.join();
    // the joining consumer's pattern: a byte budget lets short buffers accumulate, try_pop takes along what is queued
    {
        FlowBufferQueue jq(3);
        jq.set_byte_budget(1 << 20, 48);
        thread jp([&] {
            for (int i = 0; i < 400; ++i) {
                vector<Mat> fr(1);
                fr[0].create(Size(64, 1 + i % 5), CV_8UC1);
                jq.push(FlowBuffer(std::move(fr), path(), i, false), i == 399);
            }
        });
        int expect = 0;
        bool done = false;
        while (!done) {
            bool fin = false;
            FlowBuffer b = jq.pop(&fin);
            if (b.base_start != expect++) sum = -1000000;
            FlowBuffer nx({}, path(), 0, false);
            while (!fin && jq.try_pop(nx, &fin))
                if (nx.base_start != expect++) sum = -1000000;
            done = fin;
        }
        jp.join();
        if (expect != 400) sum = -1000000;
    }
    // parallel opens + reads + jpeg encodes with pooled Mats
    std::atomic<int> bad(0);
    parallelFor(8, 8, [&](int) {
        for (int r = 0; r < 50; ++r) {
            VideoCapture cap; Mat f;
            if (!cap.open(argv[1]) || !cap.read(f)) { bad++; continue; }
            vector<uchar> out; imencodeJpeg(f, out);
            Mat planes[2]; Mat flow(Size(f.cols, f.rows), CV_32FC2);
            for (int y = 0; y < f.rows; ++y) for (int x = 0; x < 2 * f.cols; ++x) flow.ptr<float>(y)[x] = (float)(x - y) * 0.01f;
            split(flow, planes);
            vector<uchar> ex, ey, png; encodeFlowMap(planes[0], planes[1], ex, ey, 20); encodeFlowMapPng(planes[0], planes[1], png);
        }
    });
    printf("sum %ld bad %d\n", sum, bad.load());
    return bad.load();
}
