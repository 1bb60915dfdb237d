// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the handful of OpenCV names [un-vendored dependency of /root/reference, absent from this image] the reference bodies
// compiled into oracle/_ref touch: cv::Mat as a typed 2-D array (8-bit 3-channel, float 1- and 3-channel), cv::Vec3b / cv::Vec3f,
// the tick counter of nv::Timer.  Images live in shared buffers (clone() copies, assignment aliases — as cv::Mat does).
// No image processing (pyrDown, cvtColor, imread) is provided: pyramids are handed in from outside.  Nothing of this is reference code.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC3 16
#define CV_32FC1 5
#define CV_32FC3 21

namespace cv {

template <class T, int N> struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
    Vec(T a, T b, T c) { static_assert(N == 3, "size"); val[0] = a; val[1] = b; val[2] = c; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
};
typedef Vec<unsigned char, 3> Vec3b;
typedef Vec<float, 3> Vec3f;

struct Size { int width, height; Size(int w = 0, int h = 0) : width(w), height(h) {} };

struct Mat {
    int rows = 0, cols = 0, type_ = CV_32FC1;
    unsigned char* data = nullptr;
    const float* p = nullptr;                        // (legacy alias used by the first generation of wrappers: float view of data)
    std::shared_ptr<std::vector<unsigned char>> own;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    static size_t elem(int type) { return type == CV_8UC3 ? 3 : type == CV_32FC3 ? 12 : 4; }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        own = std::make_shared<std::vector<unsigned char>>((size_t)r * c * elem(type), (unsigned char)0);
        data = own->data(); p = reinterpret_cast<const float*>(data);
    }
    // borrow caller memory (the C ABI wrappers do this; the buffer outlives the call)
    static Mat wrap(int r, int c, int type, const void* ptr) { Mat m; m.rows = r; m.cols = c; m.type_ = type; m.data = (unsigned char*)ptr; m.p = (const float*)ptr; return m; }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat zeros(Size s, int type) { return Mat(s.height, s.width, type); }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return (type_ == CV_8UC3 || type_ == CV_32FC3) ? 3 : 1; }
    Mat clone() const { Mat m; if (empty()) return m; m.create(rows, cols, type_); std::memcpy(m.data, data, (size_t)rows * cols * elem(type_)); return m; }
    template <class T> T& at(int y, int x) { return reinterpret_cast<T*>(data)[(size_t)y * cols + x]; }
    template <class T> const T& at(int y, int x) const { return reinterpret_cast<const T*>(data)[(size_t)y * cols + x]; }
};

inline int64_t getTickCount() { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline double getTickFrequency() { return 1e9; }

}  // namespace cv
