// src/common.cpp — Mat, flow quantisation and the on-disk writers of the host shell.
// Behaviour follows /root/reference/src/common.cpp: CAST quantisation :4-16, PNG scheme :18-46,
// encodeFlowMap :48-64, file naming :73-118.  Host glue, not on the GPU path.
#include "common.h"
#include "h5mini.h"

#include <algorithm>
#include <cmath>
#include <unordered_map>

#include "image_io.h"

// ------------------------------------------------------------------------------------------------ Mat

static bool g_page_locked = false;

void Mat::setPageLocked(bool on) { g_page_locked = on; }

namespace {
// Page-locking a buffer costs a driver call that maps and pins its pages (around a millisecond for a
// 1080p plane); the pipeline allocates three such buffers per flow.  Released buffers are therefore kept,
// keyed by size, and handed out again: frames and flows of one video all have the same few sizes.
class PinnedPool {
  public:
    void *get(size_t bytes) {
        {
            unique_lock<mutex> lock(mtx_);
            auto it = free_.find(bytes);
            if (it != free_.end()) {
                void *p = it->second;
                free_.erase(it);
                pooled_ -= bytes;
                return p;
            }
        }
        void *p = nullptr;
        return dfx_host_alloc(&p, bytes) == DFX_OK ? p : nullptr;
    }
    void put(void *p, size_t bytes) {
        {
            unique_lock<mutex> lock(mtx_);
            if (pooled_ + bytes <= kMaxPooled) {
                free_.emplace(bytes, p);
                pooled_ += bytes;
                return;
            }
        }
        dfx_host_free(p);
    }

  private:
    static constexpr size_t kMaxPooled = 6ull << 30;
    mutex mtx_;
    std::unordered_multimap<size_t, void *> free_;
    size_t pooled_ = 0;
};
PinnedPool &pinned_pool() {
    static PinnedPool *pool = new PinnedPool(); // never destroyed: buffers may outlive static destruction order
    return *pool;
}
} // namespace

void Mat::create(Size s, int type) {
    if (buf_ && rows == s.height && cols == s.width && type_ == type)
        return;
    rows = s.height;
    cols = s.width;
    type_ = type;
    step = (size_t)cols * elemSize();
    const size_t bytes = std::max<size_t>(step * rows, 1);
    void *p = g_page_locked ? pinned_pool().get(bytes) : nullptr;
    if (p) {
        buf_.reset((uchar *)p, [bytes](uchar *q) { pinned_pool().put(q, bytes); });
    } else {
        buf_.reset((uchar *)std::malloc(bytes), std::free);
    }
}

Mat Mat::clone() const {
    Mat m;
    if (!empty()) {
        m.create(size(), type_);
        std::memcpy(m.data(), data(), step * rows);
    }
    return m;
}

void split(const Mat &flow, Mat planes[2]) {
    planes[0].create(flow.size(), CV_32FC1);
    planes[1].create(flow.size(), CV_32FC1);
    for (int y = 0; y < flow.rows; ++y) {
        const float *s = flow.ptr<float>(y);
        float *a = planes[0].ptr<float>(y), *b = planes[1].ptr<float>(y);
        for (int x = 0; x < flow.cols; ++x) {
            a[x] = s[2 * x];
            b[x] = s[2 * x + 1];
        }
    }
}

// ------------------------------------------------------------------------------------------------ quantisation

static inline int cv_round(double v) { return (int)std::lrint(v); } // round-half-even

static inline uchar cast_bound(double v, double lo, double hi) { // the reference's CAST macro
    return v > hi ? 255 : v < lo ? 0 : (uchar)cv_round(255 * (v - lo) / (hi - lo));
}

void convertFlowToImage(const Mat &flow_x, const Mat &flow_y, Mat &img_x, Mat &img_y, double lowerBound,
                        double higherBound) {
    for (int i = 0; i < flow_x.rows; ++i) {
        const float *fx = flow_x.ptr<float>(i), *fy = flow_y.ptr<float>(i);
        uchar *ox = img_x.ptr<uchar>(i), *oy = img_y.ptr<uchar>(i);
        for (int j = 0; j < flow_y.cols; ++j) {
            ox[j] = cast_bound(fx[j], lowerBound, higherBound);
            oy[j] = cast_bound(fy[j], lowerBound, higherBound);
        }
    }
}

static void min_max(const Mat &m, double &mn, double &mx) {
    mn = 1e300, mx = -1e300;
    for (int y = 0; y < m.rows; ++y) {
        const float *r = m.ptr<float>(y);
        for (int x = 0; x < m.cols; ++x) {
            mn = std::min<double>(mn, r[x]);
            mx = std::max<double>(mx, r[x]);
        }
    }
}

static inline uchar saturate_u8(double v) { return (uchar)std::min(255, std::max(0, cv_round(v))); }

// 3-channel PNG: x and y scaled by an adaptive bound, third channel carries bound/4 (top half: x, bottom: y)
static void convertFlowToPngImage(const Mat &flow_x, const Mat &flow_y, Mat &img_bgr) {
    const double base = 1. / 128.;
    const double h = flow_x.rows, w = flow_x.cols;
    double mn, mx;
    min_max(flow_x, mn, mx);
    double bound_x = std::min(255. * 4, std::ceil((std::min(w, std::max(std::fabs(mn), std::fabs(mx))) * 128. / 127.) / 4) * 4);
    min_max(flow_y, mn, mx);
    double bound_y = std::min(255. * 4, std::ceil((std::min(h, std::max(std::fabs(mn), std::fabs(mx))) * 128. / 127.) / 4) * 4);
    if (int(bound_x) % 8 == 0)
        bound_x += 4;
    if (int(bound_y) % 8 == 0)
        bound_y += 4;
    const float eps_x_inv = (float)(1. / (base * bound_x));
    const float eps_y_inv = (float)(1. / (base * bound_y));
    const double half_h = h / 2;
    for (int y = 0; y < flow_x.rows; ++y) {
        const float *fx = flow_x.ptr<float>(y), *fy = flow_y.ptr<float>(y);
        uchar *o = img_bgr.ptr<uchar>(y);
        // rectangle(b, Point(0, 0), Point(w - 1, half_h), ...) (reference src/common.cpp:41): the double half_h
        // becomes Point's int by TRUNCATION and the rectangle is inclusive; the second one starts at int(half_h + 1)
        const uchar b = (y <= (int)half_h) ? saturate_u8(bound_x / 4) : saturate_u8(bound_y / 4);
        for (int x = 0; x < flow_x.cols; ++x) {
            // Mat::convertTo(CV_8U, alpha, beta) on a float source works in float: saturate_cast<uchar>(v*a + b)
            const float vx = fx[x] * eps_x_inv + 128.f, vy = fy[x] * eps_y_inv + 128.f;
            o[3 * x] = saturate_u8((double)vx);
            o[3 * x + 1] = saturate_u8((double)vy);
            o[3 * x + 2] = b;
        }
    }
}

void encodeFlowMap(const Mat &flow_map_x, const Mat &flow_map_y, vector<uchar> &encoded_x, vector<uchar> &encoded_y,
                   int bound, bool to_jpg) {
    Mat flow_img_x(flow_map_x.size(), CV_8UC1);
    Mat flow_img_y(flow_map_y.size(), CV_8UC1);
    convertFlowToImage(flow_map_x, flow_map_y, flow_img_x, flow_img_y, -bound, bound);
    if (to_jpg) {
        imencodeJpeg(flow_img_x, encoded_x);
        imencodeJpeg(flow_img_y, encoded_y);
    } else {
        encoded_x.assign(flow_img_x.data(), flow_img_x.data() + flow_img_x.total());
        encoded_y.assign(flow_img_y.data(), flow_img_y.data() + flow_img_y.total());
    }
}

void encodeFlowMapPng(const Mat &flow_map_x, const Mat &flow_map_y, vector<uchar> &encoded) {
    Mat flow_img_bgr(flow_map_x.size(), CV_8UC3);
    convertFlowToPngImage(flow_map_x, flow_map_y, flow_img_bgr);
    imencodePng(flow_img_bgr, encoded);
}

void encodeFlowMapPngPlanes(const Mat &plane_x, const Mat &plane_y, double bound_x, double bound_y, vector<uchar> &encoded) {
    Mat flow_img_bgr(plane_x.size(), CV_8UC3);
    const double half_h = (double)plane_x.rows / 2;
    for (int y = 0; y < plane_x.rows; ++y) {
        const uchar *px = plane_x.ptr<uchar>(y), *py = plane_y.ptr<uchar>(y);
        uchar *o = flow_img_bgr.ptr<uchar>(y);
        const uchar b = (y <= (int)half_h) ? saturate_u8(bound_x / 4) : saturate_u8(bound_y / 4); // as in convertFlowToPngImage
        for (int x = 0; x < plane_x.cols; ++x) {
            o[3 * x] = px[x];
            o[3 * x + 1] = py[x];
            o[3 * x + 2] = b;
        }
    }
    imencodePng(flow_img_bgr, encoded);
}

// ------------------------------------------------------------------------------------------------ writers

static void write_blob(const string &file, const vector<uchar> &blob) {
    FILE *fp = fopen(file.c_str(), "wb");
    if (!fp)
        throw std::runtime_error("cannot write " + file);
    fwrite(blob.data(), 1, blob.size(), fp);
    fclose(fp);
}

// flow file suffix: _%05d, _p<step>_%05d for step > 1, _m<|step|>_%05d for step < 0 (index offset |step|)
static string flow_suffix(int step, int index, const char *ext) {
    char tmp[64];
    const int base = step > 0 ? 0 : -step;
    if (step > 1)
        snprintf(tmp, sizeof tmp, "_p%d_%05d%s", step, index + base, ext);
    else if (step < 0)
        snprintf(tmp, sizeof tmp, "_m%d_%05d%s", -step, index + base, ext);
    else
        snprintf(tmp, sizeof tmp, "_%05d%s", index + base, ext);
    return tmp;
}

void writeImages(vector<vector<uchar>> images, string name_prefix, const int start) {
    for (size_t i = 0; i < images.size(); ++i) {
        char tmp[64];
        snprintf(tmp, sizeof tmp, "_%05d.jpg", start + (int)i);
        write_blob(name_prefix + tmp, images[i]);
    }
}

void writeFlowImages(vector<vector<uchar>> images, string name_prefix, const int step, const int start) {
    for (size_t i = 0; i < images.size(); ++i)
        write_blob(name_prefix + flow_suffix(step, start + (int)i, ".jpg"), images[i]);
}

// One flow image whose bytes already exist (the device JPEG encoder's files): same naming as writeFlowImages, no copy.
void writeFlowImageBytes(const uchar *bytes, size_t size, const string &name_prefix, int step, int index) {
    const string file = name_prefix + flow_suffix(step, index, ".jpg");
    FILE *fp = fopen(file.c_str(), "wb");
    if (!fp)
        throw std::runtime_error("cannot write " + file);
    const size_t n = fwrite(bytes, 1, size, fp);
    if (fclose(fp) != 0 || n != size)
        throw std::runtime_error("cannot write " + file);
}

void writeFlowImagesPng(vector<vector<uchar>> images, string name_prefix, const int step, const int start) {
    for (size_t i = 0; i < images.size(); ++i)
        write_blob(name_prefix + flow_suffix(step, start + (int)i, ".png"), images[i]);
}

// -st=h5 (reference: writeHDF5, /root/reference/src/common.cpp:121-149, built there only with USE_HDF5=ON): the file
// `<name_prefix>.h5` (`_p%d.h5` / `_m%d.h5` for other steps) was created when the video was opened; every
// FlowBuffer adds one float dataset per flow, `/<phase>_%05d` with the step infix of the image files.
string h5FileName(const string &name_prefix, int step) {
    char ext[32];
    if (step > 1)
        snprintf(ext, sizeof ext, "_p%d.h5", step);
    else if (step < 0)
        snprintf(ext, sizeof ext, "_m%d.h5", -step);
    else
        snprintf(ext, sizeof ext, ".h5");
    return name_prefix + ext;
}

void createHDF5(const string &name_prefix, int step) { h5mini::create(h5FileName(name_prefix, step)); }

void writeHDF5(const vector<Mat> &images, string name_prefix, string phase, const int step, const int start) {
    vector<h5mini::FloatDataset> ds;
    ds.reserve(images.size());
    for (size_t i = 0; i < images.size(); ++i)
        ds.push_back({phase + flow_suffix(step, start + (int)i, ""), (size_t)images[i].rows, (size_t)images[i].cols,
                      images[i].ptr<float>(), images[i].step});
    h5mini::append(h5FileName(name_prefix, step), ds);
}

