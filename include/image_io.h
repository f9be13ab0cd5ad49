// include/image_io.h — decoder-free media I/O of the host shell (no OpenCV / ffmpeg in this build):
// readers for YUV4MPEG2 (.y4m) clips and PGM/PPM frames, a baseline JPEG encoder and a PNG writer.
// Out of the hot path; kept small and dependency-free.
#ifndef DENSEFLOW_IMAGE_IO_H
#define DENSEFLOW_IMAGE_IO_H

#include "common.h"

// Stand-in for cv::VideoCapture: sequential gray frames from a .y4m file (the Y plane is the gray
// image; the reference converts decoded BGR with cvtColor, src/denseflow_gpu.cpp:163).
class VideoCapture {
  public:
    bool open(const string &file);
    bool isOpened() const { return (bool)f_; }
    void release() { f_.reset(); }
    int width() const { return w_; }
    int height() const { return h_; }
    int frameCount() const { return frames_; } // -1 when unknown
    bool read(Mat &gray);                      // false at end of stream
    bool seekFrame(int index);                 // the next read() returns frame `index` (frames without parameters)

  private:
    std::shared_ptr<FILE> f_;
    int w_ = 0, h_ = 0, frames_ = -1;
    size_t chroma_bytes_ = 0;
    long data_start_ = 0;
};

bool imreadGray(const string &file, Mat &gray);          // .pgm (P5) / .ppm (P6, BGR2GRAY fixed-point weights)
void resizeLinear(const Mat &src, Mat &dst, Size size);  // bilinear, half-pixel centres (cv::resize default)

bool imencodeJpeg(const Mat &gray, vector<uchar> &out, int quality = 95); // baseline, 8-bit gray
void imencodeJpegForcePortable(bool on); // testing aid: bypass the AVX2 transform
bool imencodePng(const Mat &img, vector<uchar> &out);                     // 8-bit gray or BGR, stored deflate

#endif
