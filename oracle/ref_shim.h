// oracle/ref_shim.h — TEST INFRASTRUCTURE ONLY.
// The few OpenCV names that /root/reference/src/common.cpp:4-16 (convertFlowToImage) uses, so that those
// reference lines can be compiled as they are (oracle/Makefile, target `ref`).  cv::Mat here is a plain
// dense view; cvRound is OpenCV's x86-64 definition (cvtsd2si: round half to even, INT_MIN on NaN).
#pragma once
#include <emmintrin.h>
#include <stddef.h>

typedef unsigned char uchar;

struct Mat {
    int rows, cols;
    unsigned char *data;
    size_t step; // bytes per row
    template <class T> T &at(int i, int j) const { return reinterpret_cast<T *>(data + (size_t)i * step)[j]; }
};

static inline int cvRound(double value) { return _mm_cvtsd_si32(_mm_set_sd(value)); }
