// The few cv:: types the hot path's public surface mentions (cv::Point2f, cv::Mat), for builds
// where OpenCV is not installed.  When <opencv2/core.hpp> exists it is used instead and this file
// defines nothing, so vslam.cpp / PointMap.cpp keep compiling against the real types.
#pragma once

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define VSLAM_HAVE_OPENCV 1
#else
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    // OpenCV converts through saturate_cast: float -> int rounds to nearest-even (cvRound), it does not truncate
    template <typename U>
    Point_(const Point_<U> &o) : x(convert_(o.x)), y(convert_(o.y)) {}
    T dot(const Point_ &o) const { return x * o.x + y * o.y; }
    bool operator==(const Point_ &o) const { return x == o.x && y == o.y; }
    bool operator!=(const Point_ &o) const { return !(*this == o); }

   private:
    template <typename U>
    static T convert_(U v) {
        if (std::is_integral<T>::value && std::is_floating_point<U>::value) return static_cast<T>(std::lrint(v));
        return static_cast<T>(v);
    }
};
template <typename T>
inline Point_<T> operator-(const Point_<T> &a, const Point_<T> &b) { return Point_<T>(a.x - b.x, a.y - b.y); }
using Point2f = Point_<float>;
using Point = Point_<int>;

// Row-major, reference-counted matrix: just enough of cv::Mat for Frame / RansacFilter
class Mat {
   public:
    int rows = 0, cols = 0;
    unsigned char *data = nullptr;
    size_t step = 0;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *external, size_t step_bytes = 0)
        : rows(r), cols(c), data(static_cast<unsigned char *>(external)), type_(type) {
        step = step_bytes ? step_bytes : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elemSize();
        store_ = std::shared_ptr<unsigned char>(new unsigned char[step * (size_t)(r > 0 ? r : 0) + 1],
                                                std::default_delete<unsigned char[]>());
        data = store_.get();
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_32F ? 4 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <typename T>
    T *ptr(int r = 0) { return reinterpret_cast<T *>(data + step * (size_t)r); }
    template <typename T>
    const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + step * (size_t)r); }
    template <typename T>
    T &at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T>
    const T &at(int r, int c) const { return ptr<T>(r)[c]; }
    template <typename T>
    T &at(int i) { return ptr<T>(0)[i]; }
    void copyTo(Mat &dst) const {
        dst.create(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(dst.data + dst.step * r, data + step * r, (size_t)cols * elemSize());
    }

   private:
    int type_ = 0;
    std::shared_ptr<unsigned char> store_;
};

}  // namespace cv
#endif
