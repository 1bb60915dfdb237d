// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the handful of OpenCV names [un-vendored dependency of /root/reference, absent from this image] the reference bodies
// compiled into oracle/_ref touch: cv::Mat as a typed 2-D array (8-bit 3-channel, float 1- and 3-channel), cv::Vec3b / cv::Vec3f,
// the tick counter of nv::Timer.  Images live in shared buffers (clone() copies, assignment aliases — as cv::Mat does).
// No pyramid / file image processing (pyrDown, imread) is provided: pyramids are handed in from outside.  For KeyframeSelection::estimateBlur the few
// OpenCV calls it makes are restated at the bottom (cvtColor BGR2GRAY in 8-bit fixed point, convertTo with a scale, a float filter2D with the default
// BORDER_REFLECT_101 border, transpose, sum) — OpenCV's published behaviour as we read it, NOT OpenCV: the product and this agree by construction there;
// what the reference contributes is the class around them.  Nothing of this is reference code.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8UC1 0
#define CV_16UC1 2
#define CV_8UC3 16
#define CV_32FC1 5
#define CV_32F 5
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
    static size_t elem(int type) { return type == CV_8UC1 ? 1 : type == CV_16UC1 ? 2 : type == CV_8UC3 ? 3 : type == CV_32FC3 ? 12 : 4; }
    explicit Mat(const std::vector<unsigned char>& buf) { create(1, (int)buf.size(), CV_8UC1); if (!buf.empty()) std::memcpy(data, buf.data(), buf.size()); }      // an encoded file handed to imdecode
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        own = std::make_shared<std::vector<unsigned char>>((size_t)r * c * elem(type), (unsigned char)0);
        data = own->data(); p = reinterpret_cast<const float*>(data);
    }
    // borrow caller memory (the C ABI wrappers do this; the buffer outlives the call)
    static Mat wrap(int r, int c, int type, const void* ptr) { Mat m; m.rows = r; m.cols = c; m.type_ = type; m.data = (unsigned char*)ptr; m.p = (const float*)ptr; return m; }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat zeros(Size s, int type) { return Mat(s.height, s.width, type); }
    static Mat ones(int r, int c, int type) { Mat m(r, c, type); for (size_t i = 0; i < (size_t)r * c; ++i) reinterpret_cast<float*>(m.data)[i] = 1.0f; return m; }      // (CV_32F only)
    // dst = saturate_cast<float>(src * alpha): 8- / 16-bit 1-channel -> float with the scale applied in float (cv::Mat::convertTo, the only conversions the reference bodies use)
    void convertTo(Mat& dst, int type, double alpha = 1.0) const {
        if (empty()) { dst = Mat(); return; }
        if (type_ == CV_8UC3 && type == CV_32FC3) {                  // colour image to float, channel by channel
            Mat o3(rows, cols, CV_32FC3); const float a3 = (float)alpha;
            for (size_t i = 0; i < (size_t)rows * cols * 3; ++i) reinterpret_cast<float*>(o3.data)[i] = (float)data[i] * a3;
            dst = o3; return;
        }
        Mat o(rows, cols, type); const float a = (float)alpha;
        for (size_t i = 0; i < (size_t)rows * cols; ++i) reinterpret_cast<float*>(o.data)[i] = (type_ == CV_16UC1 ? (float)reinterpret_cast<const uint16_t*>(data)[i] : (float)data[i]) * a;
        dst = o;
    }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return (type_ == CV_8UC3 || type_ == CV_32FC3) ? 3 : 1; }
    Mat clone() const { Mat m; if (empty()) return m; m.create(rows, cols, type_); std::memcpy(m.data, data, (size_t)rows * cols * elem(type_)); return m; }
    template <class T> T& at(int y, int x) { return reinterpret_cast<T*>(data)[(size_t)y * cols + x]; }
    template <class T> const T& at(int y, int x) const { return reinterpret_cast<const T*>(data)[(size_t)y * cols + x]; }
};

class FileNode; class FileStorage;                  // only named in private declarations of nv::Settings that are never defined here
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {} };
inline Mat operator*(const Mat& m, double f) { Mat o = m.clone(); const float ff = (float)f; for (size_t i = 0; i < (size_t)o.rows * o.cols; ++i) reinterpret_cast<float*>(o.data)[i] *= ff; return o; }
inline void transpose(const Mat& a, Mat& b) { Mat o(a.cols, a.rows, a.type()); for (int y = 0; y < a.rows; ++y) for (int x = 0; x < a.cols; ++x) o.at<float>(x, y) = a.at<float>(y, x); b = o; }
enum { COLOR_BGR2GRAY = 6 };
// 8-bit BGR -> grey in OpenCV's 14-bit fixed point: (B 1868 + G 9617 + R 4899 + 2^13) >> 14
inline void cvtColor(const Mat& src, Mat& dst, int /*code*/) {
    if (src.type() == CV_32FC3) {                                    // float images: gray = 0.114 b + 0.587 g + 0.299 r in float
        Mat of(src.rows, src.cols, CV_32FC1); const float* p = reinterpret_cast<const float*>(src.data);
        for (size_t i = 0; i < (size_t)src.rows * src.cols; ++i) reinterpret_cast<float*>(of.data)[i] = (p[3 * i] * 0.114f + p[3 * i + 1] * 0.587f) + p[3 * i + 2] * 0.299f;
        dst = of; return;
    }
    Mat o(src.rows, src.cols, CV_8UC1);
    for (size_t i = 0; i < (size_t)src.rows * src.cols; ++i) o.data[i] = (unsigned char)((src.data[3 * i] * 1868 + src.data[3 * i + 1] * 9617 + src.data[3 * i + 2] * 4899 + (1 << 13)) >> 14);
    dst = o;
}
// float correlation with the kernel anchored at its centre, BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), products accumulated in float in kernel order
inline void filter2D(const Mat& src, Mat& dst, int /*ddepth*/, const Mat& k) {
    auto refl = [](int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p; return p; };
    Mat o(src.rows, src.cols, CV_32FC1); const int ay = k.rows / 2, ax = k.cols / 2;
    for (int y = 0; y < src.rows; ++y) for (int x = 0; x < src.cols; ++x) {
        float s = 0.0f;
        for (int j = 0; j < k.rows; ++j) for (int i = 0; i < k.cols; ++i) s += k.at<float>(j, i) * src.at<float>(refl(y + j - ay, src.rows), refl(x + i - ax, src.cols));
        o.at<float>(y, x) = s;
    }
    dst = o;
}
// cv::pyrDown to (cols/2, rows/2): [1 4 6 4 1] x [1 4 6 4 1] / 256, BORDER_REFLECT_101, horizontal pass first.  Float images in float; 8-bit images in
// integers with (sum + 128) >> 8.
inline void pyrDown(const Mat& src, Mat& dst, Size sz) {
    auto refl = [](int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i; return i; };
    const int ow = sz.width, oh = sz.height, w = src.cols, h = src.rows;
    if (src.type() == CV_32FC1) {
        Mat o(oh, ow, CV_32FC1); const float* s = reinterpret_cast<const float*>(src.data);
        for (int y = 0; y < oh; ++y) for (int x = 0; x < ow; ++x) {
            float row[5];
            for (int j = 0; j < 5; ++j) {
                const float* line = s + (size_t)refl(2 * y - 2 + j, h) * w;
                const float m2 = line[refl(2 * x - 2, w)], m1 = line[refl(2 * x - 1, w)], c0 = line[refl(2 * x, w)], p1 = line[refl(2 * x + 1, w)], p2 = line[refl(2 * x + 2, w)];
                row[j] = ((c0 * 6.0f + (m1 + p1) * 4.0f) + m2) + p2;
            }
            o.at<float>(y, x) = (((row[2] * 6.0f + (row[1] + row[3]) * 4.0f) + row[0]) + row[4]) * (1.0f / 256.0f);
        }
        dst = o; return;
    }
    const int ch = src.channels(); Mat o(oh, ow, src.type());
    for (int y = 0; y < oh; ++y) for (int x = 0; x < ow; ++x) for (int c = 0; c < ch; ++c) {
        int acc = 0; const int wt[5] = {1, 4, 6, 4, 1};
        for (int j = 0; j < 5; ++j) { const unsigned char* line = src.data + (size_t)refl(2 * y - 2 + j, h) * w * ch; int r = 0; for (int i = 0; i < 5; ++i) r += wt[i] * line[refl(2 * x - 2 + i, w) * ch + c]; acc += wt[j] * r; }
        o.data[((size_t)y * ow + x) * ch + c] = (unsigned char)((acc + 128) >> 8);
    }
    dst = o;
}
// cv::threshold on float images: THRESH_TOZERO keeps src where src > thresh, THRESH_TOZERO_INV where src <= thresh (the comparison in float)
enum { THRESH_TOZERO = 3, THRESH_TOZERO_INV = 4 };
inline double threshold(const Mat& src, Mat& dst, double thresh, double /*maxval*/, int type) {
    Mat o = src.clone(); const float t = (float)thresh; float* p = reinterpret_cast<float*>(o.data);
    for (size_t i = 0; i < (size_t)o.rows * o.cols; ++i) { const bool above = p[i] > t; if (type == THRESH_TOZERO ? !above : above) p[i] = 0.0f; }
    dst = o; return thresh;
}
// cv::imdecode: decoding is NOT restated here — the harness installs a decoder (tests: Pillow through a ctypes callback); without one images are empty
enum { IMREAD_UNCHANGED = -1 };
typedef int (*imdecode_hook_t)(const unsigned char* buf, size_t size, int* rows, int* cols, int* type, unsigned char* out /* NULL: sizes only */);
inline imdecode_hook_t& imdecode_hook() { static imdecode_hook_t h = nullptr; return h; }
inline Mat imdecode(const Mat& buf, int /*flags*/) {
    Mat m; if (!imdecode_hook() || buf.empty()) return m;
    int r = 0, c = 0, t = 0;
    if (!imdecode_hook()(buf.data, (size_t)buf.cols, &r, &c, &t, nullptr) || r <= 0 || c <= 0) return m;
    m.create(r, c, t); imdecode_hook()(buf.data, (size_t)buf.cols, &r, &c, &t, m.data); return m;
}
// interactive display / drawing of AppKeyframes' `show_keyframes` branch and KeyframeSelection::drawScore: never run by the harness, present so that the
// reference bodies compile unchanged
struct Point { int x, y; Point(int a = 0, int b = 0) : x(a), y(b) {} };
enum { FONT_HERSHEY_COMPLEX = 3 };
inline void putText(Mat&, const std::string&, Point, int, double, Scalar, int = 1) {}
inline void imshow(const std::string&, const Mat&) {}
inline int waitKey(int = 0) { return -1; }
inline void destroyWindow(const std::string&) {}
inline Scalar sum(const Mat& m) { double s = 0.0; for (size_t i = 0; i < (size_t)m.rows * m.cols; ++i) s += (double)reinterpret_cast<const float*>(m.data)[i]; return Scalar(s); }

inline int64_t getTickCount() { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline double getTickFrequency() { return 1e9; }

}  // namespace cv
