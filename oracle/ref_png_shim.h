// oracle/ref_png_shim.h — TEST INFRASTRUCTURE ONLY.
// The OpenCV names that /root/reference/src/common.cpp:18-46 (convertFlowToPngImage) uses, so that those reference
// lines can be compiled as they stand (oracle/Makefile, target `ref`): cv::Mat as a plain dense owner / view,
// minMaxLoc, Mat::convertTo(CV_8UC1, alpha, beta), rectangle(FILLED), mixChannels and Point.  What each stand-in does
// is OpenCV 4.5.2's documented behaviour for exactly the argument types those lines pass [UPSTREAM-MEM]:
//   * minMaxLoc(CV_32F): minimum and maximum element as double (NaNs never win a comparison; 0 / 0 for a plane of NaNs);
//   * convertTo(CV_8U, alpha, beta) from CV_32F: saturate_cast<uchar>(src * (float)alpha + (float)beta), the product
//     and the sum rounded separately in float, saturate_cast = cvRound (round half to even) clamped to [0, 255].
//     AN ASSUMPTION, NOT A PIN: this is the scalar tail of cv::cvt_32f8u.  The vector body of OpenCV 4.x's
//     cvt32f8u is written with v_fma / v_muladd, which the AVX2 / FMA3 dispatch executes as ONE fused operation: a real
//     cv::imencode input may then differ from this reading by 1 LSB at pixels where src * alpha + 128 lands within half a
//     float ulp of a .5 tie (rare; never on an all-zero flow).  The device (quantize_kernels.hip: png_cast) and
//     oracle/quant_oracle.c follow this unfused reading; tests/test_opencv_pin.py's kit is where a real OpenCV build
//     settles it, and a fused variant is a one-line change in all three places;
//   * rectangle(img, Point a, Point b, Scalar v, FILLED): every pixel of the inclusive box [a, b] clipped to the image
//     set to saturate_cast<uchar>(v); Point(double, double) converts by cvRound? NO — Point_<int>(w - 1, half_h) is the
//     int constructor: the doubles the reference passes are converted by the C++ implicit conversion = truncation;
//   * mixChannels with from_to {0,0, 1,1, 2,2}: interleave the three single-channel planes.
#pragma once
#include <emmintrin.h>
#include <stddef.h>

#include <algorithm>
#include <cmath>
#include <vector>

using namespace std; // the reference's common.h does the same; its lines use unqualified min / max / abs / ceil

typedef unsigned char uchar;
enum { CV_8UC1 = 0, CV_8UC3 = 16, CV_32FC1 = 5, FILLED = -1 };

static inline int cvRound(double value) { return _mm_cvtsd_si32(_mm_set_sd(value)); }
static inline uchar saturate_u8(double v) { return (uchar)std::min(255, std::max(0, cvRound(v))); }

struct Size {
    int width, height;
};
struct Point {
    int x, y;
    Point(int x_, int y_) : x(x_), y(y_) {}
};

struct Mat {
    int rows = 0, cols = 0, type_ = CV_8UC1;
    unsigned char *data = nullptr;
    size_t step = 0; // bytes per row
    std::vector<unsigned char> own;
    Mat() {}
    Mat(int r, int c, int type, void *d, size_t s) : rows(r), cols(c), type_(type), data((unsigned char *)d), step(s) {}
    Mat(Size sz, int type) : rows(sz.height), cols(sz.width), type_(type) {
        const size_t es = type == CV_8UC3 ? 3 : type == CV_32FC1 ? 4 : 1;
        step = es * cols;
        own.assign(step * rows, 0);
        data = own.data();
    }
    Size size() const { return Size{cols, rows}; }
    template <class T> T &at(int i, int j) const { return reinterpret_cast<T *>(data + (size_t)i * step)[j]; }
    void convertTo(Mat &dst, int, double alpha, double beta) const { // CV_32F -> CV_8U
        const float a = (float)alpha, b = (float)beta;
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < cols; ++j) {
                const float p = at<float>(i, j) * a;
                const float v = p + b;
                dst.at<uchar>(i, j) = saturate_u8((double)v);
            }
    }
};

static inline void minMaxLoc(const Mat &m, double *mn, double *mx) {
    // upstream minMaxIdx searches from +-infinity sentinels (a NaN never wins, wherever it stands) and reports 0 / 0
    // when it located nothing — a plane of NaNs only [UPSTREAM-MEM]
    double lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < m.rows; ++i)
        for (int j = 0; j < m.cols; ++j) {
            const double v = m.at<float>(i, j);
            if (v < lo)
                lo = v;
            if (v > hi)
                hi = v;
        }
    if (lo > hi)
        lo = hi = 0.0;
    *mn = lo;
    *mx = hi;
}

static inline void rectangle(Mat &img, Point a, Point b, double value, int) {
    const uchar v = saturate_u8(value);
    for (int y = std::max(std::min(a.y, b.y), 0); y <= std::min(std::max(a.y, b.y), img.rows - 1); ++y)
        for (int x = std::max(std::min(a.x, b.x), 0); x <= std::min(std::max(a.x, b.x), img.cols - 1); ++x)
            img.at<uchar>(y, x) = v;
}

static inline void mixChannels(const Mat *src, size_t nsrc, Mat *dst, size_t, const int *from_to, size_t npairs) {
    for (size_t k = 0; k < npairs; ++k) {
        const Mat &s = src[from_to[2 * k]];
        const int ch = from_to[2 * k + 1];
        for (int i = 0; i < s.rows; ++i)
            for (int j = 0; j < s.cols; ++j)
                dst->data[(size_t)i * dst->step + 3 * j + ch] = s.at<uchar>(i, j);
    }
    (void)nsrc;
}
