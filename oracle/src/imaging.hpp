// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Image containers + the float camera of the observation pass + the Catmull-Rom
// bicubic sampler of the residual.
//   camera.cpp:124-154       Camera::project (float; distortion skipped when all |k|<=1e-5)
//   rgbd/processing.cpp:238-301  bilinear interpolate<T>, interpolateRGB
//   cost.h:108-127           interpolate() -> ceres::Grid2D<float,1,true,true> + BiCubicInterpolator,
//                            called as Evaluate(row = y, col = x)
//   [Ceres 2.1.0 cubic_interpolation.h, not in reference] CubicHermiteSpline / clamped Grid2D
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include "jet.hpp"

namespace orc {

struct Image {            // one pyramid level of one keyframe
    int w = 0, h = 0;
    const float* lum = nullptr;     // h*w, luminance in [0,1]
    const float* depth = nullptr;   // h*w, metres, 0 = invalid
    const uint8_t* bgr = nullptr;   // h*w*3 (OpenCV order), may be null when only optimize() is exercised
};

struct Frames {
    int K = 0, levels = 0;
    std::vector<Image> img;         // [K*levels], index f*levels + lvl
    const Image& at(int f, int lvl) const { return img[(size_t)f * levels + lvl]; }
};

struct CameraF {          // camera.h:50-100 (float model used by SDFColorization)
    float fx, fy, cx, cy; int w, h; float k[5];
    bool dist_is_zero() const {           // Eigen isZero(): all |k| <= 1e-5 [Eigen, not in reference]
        for (int i = 0; i < 5; ++i) if (std::fabs(k[i]) > 1e-5f) return false;
        return true; }
    bool project(const float p[3], float p2f[2], int p2i[2]) const {
        float x = p[0] / p[2], y = p[1] / p[2];
        if (!dist_is_zero()) {
            const float r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
            const float dc = 1.0f + k[0] * r2 + k[1] * r4 + k[2] * r6;
            x = x * dc + 2.0f * k[3] * x * y + k[4] * (r2 + 2.0f * x * x);
            y = y * dc + 2.0f * k[4] * x * y + k[3] * (r2 + 2.0f * y * y);   // uses the distorted x (hazard 3)
        }
        p2f[0] = fx * x + cx; p2f[1] = fy * y + cy;
        p2i[0] = (int)(p2f[0] + 0.5f); p2i[1] = (int)(p2f[1] + 0.5f);
        return !(p2i[0] < 0 || p2i[0] >= w || p2i[1] < 0 || p2i[1] >= h);
    }
};

// processing.cpp:238-287 for an 8-bit channel
inline uint8_t bilinear_u8(const uint8_t* img, int w, int h, int nc, float x, float y, int ch) {
    int x0 = (int)std::floor(x), y0 = (int)std::floor(y); const int x1 = x0 + 1, y1 = y0 + 1;
    float x1w = x - (float)x0, y1w = y - (float)y0, x0w = 1.0f - x1w, y0w = 1.0f - y1w;
    if (x0 < 0 || x0 >= w) x0w = 0.0f;
    if (x1 < 0 || x1 >= w) x1w = 0.0f;
    if (y0 < 0 || y0 >= h) y0w = 0.0f;
    if (y1 < 0 || y1 >= h) y1w = 0.0f;
    const float w00 = x0w * y0w, w10 = x1w * y0w, w01 = x0w * y1w, w11 = x1w * y1w;
    const float sw = w00 + w10 + w01 + w11;
    float sum = 0.0f;
    if (w00 > 0.0f) sum += (float)img[((size_t)y0 * w + x0) * nc + ch] * w00;
    if (w01 > 0.0f) sum += (float)img[((size_t)y1 * w + x0) * nc + ch] * w01;
    if (w10 > 0.0f) sum += (float)img[((size_t)y0 * w + x1) * nc + ch] * w10;
    if (w11 > 0.0f) sum += (float)img[((size_t)y1 * w + x1) * nc + ch] * w11;
    uint8_t out = 0;
    if (sw > 0.0f) out = (uint8_t)(sum / sw);
    return out;
}

// [Ceres 2.1.0] CubicHermiteSpline<1>
inline void cubic_hermite(double p0, double p1, double p2, double p3, double x, double* f, double* dfdx) {
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double c = 0.5 * (-p0 + p2);
    const double d = p1;
    if (f) *f = d + x * (c + x * (b + x * a));
    if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}

// [Ceres 2.1.0] BiCubicInterpolator<Grid2D<float,1,row-major>>::Evaluate(r, c, f, dfdr, dfdc); borders clamp
inline void bicubic(const float* img, int w, int h, double r, double c, double* f, double* dfdr, double* dfdc) {
    const int row = (int)std::floor(r), col = (int)std::floor(c);
    double fr[4], dfc[4];
    for (int i = 0; i < 4; ++i) {
        const int rr = std::min(std::max(0, row - 1 + i), h - 1);
        double p[4];
        for (int j = 0; j < 4; ++j) {
            const int cc = std::min(std::max(0, col - 1 + j), w - 1);
            p[j] = (double)img[(size_t)rr * w + cc];
        }
        cubic_hermite(p[0], p[1], p[2], p[3], c - col, &fr[i], &dfc[i]);
    }
    cubic_hermite(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
    cubic_hermite(dfc[0], dfc[1], dfc[2], dfc[3], r - row, dfdc, nullptr);
}

inline void bicubic_T(const float* img, int w, int h, const double& r, const double& c, double* out) {
    double dr, dc; bicubic(img, w, h, r, c, out, &dr, &dc);
}
template <int N>
inline void bicubic_T(const float* img, int w, int h, const Jet<N>& r, const Jet<N>& c, Jet<N>* out) {
    double f, dr, dc; bicubic(img, w, h, r.a, c.a, &f, &dr, &dc);
    out->a = f; for (int i = 0; i < N; ++i) out->v[i] = dr * r.v[i] + dc * c.v[i];
}

}  // namespace orc
