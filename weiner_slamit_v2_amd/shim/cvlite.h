// cvlite.h — the handful of OpenCV types the three hot-path class surfaces mention, for builds
// where OpenCV is not present (this image has none).  When the real OpenCV headers are available
// define SLAMIT_USE_OPENCV before including the shim headers and this file is skipped: the shim
// then uses cv::Mat / cv::KeyPoint directly, which is what a drop-in into the reference's tree
// (jni/ORB_SLAM2) does.  Layouts match OpenCV where the C-ABI relies on them: cv::KeyPoint is
// 28 bytes {Point2f pt; float size, angle, response; int octave, class_id}
// (openCVLibrary341/src/sdk/native/jni/include/opencv2/core/types.hpp:699-766).
#ifndef SLAMIT_CVLITE_H
#define SLAMIT_CVLITE_H
#ifndef SLAMIT_USE_OPENCV

#include <stdint.h>
#include <string.h>

#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5

namespace cv {

typedef unsigned char uchar;

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    Point_& operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
        : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// A minimal reference-counted dense 2-D matrix (uchar / float), enough for images, descriptor
// tables and 4x4 poses.
class Mat {
public:
    int rows, cols;
    size_t step;
    uchar* data;

    Mat() : rows(0), cols(0), step(0), data(nullptr), type_(0) {}
    Mat(int r, int c, int type) : rows(0), cols(0), step(0), data(nullptr), type_(0) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t stp = 0) : rows(r), cols(c), step(stp ? stp : (size_t)c * esz(type)), data((uchar*)ext), type_(type) {}

    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data && buf_) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        buf_.reset(new std::vector<uchar>((size_t)r * step));
        data = buf_->empty() ? nullptr : buf_->data();
    }
    void release() { buf_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    bool isContinuous() const { return step == (size_t)cols * esz(type_); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * esz(type_));
        return m;
    }
    Mat row(int r) const { Mat m(*this); m.rows = 1; m.data = data + (size_t)r * step; return m; }
    Mat rowRange(int a, int b) const { Mat m(*this); m.rows = b - a; m.data = data + (size_t)a * step; return m; }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T& at(int r, int c) { return ((T*)(data + (size_t)r * step))[c]; }
    template <typename T> const T& at(int r, int c) const { return ((const T*)(data + (size_t)r * step))[c]; }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); if (m.data) memset(m.data, 0, (size_t)r * m.step); return m; }

private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_;
    std::shared_ptr<std::vector<uchar> > buf_;
};

// cv::InputArray / cv::OutputArray as the reference's signatures use them
class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
private:
    const Mat* m_;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv

#endif  // SLAMIT_USE_OPENCV
#endif  // SLAMIT_CVLITE_H
