// include/common.h — host-shell types.  The reference's common.h pulls in OpenCV (cv::Mat, cv::Size,
// cv::cuda::Stream) and boost::filesystem (/root/reference/include/common.h:4-31); neither exists in
// the MI355X build, so the shell carries the few value types the pipeline needs.  Names are kept so the
// code above the C ABI reads like the reference's.
#ifndef DENSEFLOW_COMMON_H_H
#define DENSEFLOW_COMMON_H_H

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dfx.h"

using std::condition_variable;
using std::cout;
using std::endl;
using std::mutex;
using std::queue;
using std::string;
using std::thread;
using std::unique_lock;
using std::vector;
using std::filesystem::create_directories;
using std::filesystem::directory_iterator;
using std::filesystem::exists;
using std::filesystem::is_directory;
using std::filesystem::is_regular_file;
using std::filesystem::path;
typedef unsigned char uchar;

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
};

// element types used by the pipeline (same meaning as OpenCV's)
enum { CV_8UC1 = 0, CV_8UC3 = 1, CV_32FC1 = 2, CV_32FC2 = 3 };

// Minimal reference-counted image.  Rows are dense (step == cols * elemSize()).  With
// Mat::setPageLocked(true) (reference: Mat::setDefaultAllocator(PAGE_LOCKED), tools/denseflow.cpp:49)
// buffers come from dfx_host_alloc so host<->device copies are asynchronous.
class Mat {
  public:
    int rows = 0, cols = 0;
    size_t step = 0;
    Mat() {}
    Mat(Size s, int type) { create(s, type); }
    void create(Size s, int type);
    bool empty() const { return !buf_; }
    Size size() const { return Size(cols, rows); }
    int type() const { return type_; }
    int channels() const { return type_ == CV_8UC3 ? 3 : type_ == CV_32FC2 ? 2 : 1; }
    size_t elemSize() const { return (type_ == CV_8UC1 ? 1 : type_ == CV_8UC3 ? 3 : type_ == CV_32FC1 ? 4 : 8); }
    size_t total() const { return (size_t)rows * cols; }
    Mat clone() const;
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(buf_.get() + (size_t)r * step); }
    template <class T> const T *ptr(int r = 0) const {
        return reinterpret_cast<const T *>(buf_.get() + (size_t)r * step);
    }
    uchar *data() { return buf_.get(); }
    const uchar *data() const { return buf_.get(); }
    static void setPageLocked(bool on);

  private:
    int type_ = CV_8UC1;
    std::shared_ptr<uchar> buf_;
};

// Stand-in for cv::cuda::Stream in the reference's signatures (include/dense_flow.h:33, :58-59).  The HIP streams
// live inside the engine handle (include/dfx.h); this type only keeps the reference's API shape.
class Stream {
  public:
    static Stream &Null() {
        static Stream s;
        return s;
    }
};

void split(const Mat &flow, Mat planes[2]); // CV_32FC2 -> two CV_32FC1 (reference: cv::split, :418)

void convertFlowToImage(const Mat &flow_x, const Mat &flow_y, Mat &img_x, Mat &img_y, double lowerBound,
                        double higherBound);
void encodeFlowMap(const Mat &flow_map_x, const Mat &flow_map_y, vector<uchar> &encoded_x, vector<uchar> &encoded_y,
                   int bound, bool to_jpg = true);
void encodeFlowMapPng(const Mat &flow_map_x, const Mat &flow_map_y, vector<uchar> &encoded);
// the same from the two convertTo planes and the adaptive bounds the device already computed (dfx_submit_batch_png):
// interleave them with the bound channel (src/common.cpp:41-45) and encode
void encodeFlowMapPngPlanes(const Mat &plane_x, const Mat &plane_y, double bound_x, double bound_y, vector<uchar> &encoded);
void writeImages(vector<vector<uchar>> images, string name_prefix, const int start = 0);
void writeFlowImages(vector<vector<uchar>> images, string name_prefix, const int step = 1, const int start = 0);
void writeFlowImagesPng(vector<vector<uchar>> images, string name_prefix, const int step, const int start);
void writeFlowImageBytes(const uchar *bytes, size_t size, const string &name_prefix, int step, int index);
// -st=h5: reference src/common.cpp:121-149 (writeHDF5) and src/denseflow_gpu.cpp:223-243 (file creation); written
// by src/h5mini.cpp without libhdf5
string h5FileName(const string &name_prefix, int step);
void createHDF5(const string &name_prefix, int step);
void writeHDF5(const vector<Mat> &images, string name_prefix, string phase, const int step = 1, const int start = 0);

#endif // DENSEFLOW_COMMON_H_H
