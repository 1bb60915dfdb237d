// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the part of Ceres Solver 2.1.0 [un-vendored dependency of /root/reference, pinned only by its README.md:75; absent
// from this image] that the reference's residual functors call: Jet forward-mode duals (jet.h), AngleAxisRotatePoint
// (rotation.h), Grid2D + CubicHermiteSpline + BiCubicInterpolator (cubic_interpolation.h), IsNaN / IsInfinite.  Written from the
// published algorithms of that release, independently of oracle/src/jet.hpp, so that the reference bodies compiled into
// oracle/_ref run on a second implementation of the same semantics.  Nothing of this is reference code.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>

namespace ceres {

template <typename T, int N>
struct Jet {
    T a;
    T v[N];
    Jet() : a() { for (int i = 0; i < N; ++i) v[i] = T(); }
    Jet(const T& value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); }   // NOLINT (implicit, as in ceres)
    Jet(int value) : a(static_cast<T>(value)) { for (int i = 0; i < N; ++i) v[i] = T(); }   // NOLINT
    Jet(const T& value, int k) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); v[k] = T(1.0); }
    Jet& operator+=(const Jet& y) { a += y.a; for (int i = 0; i < N; ++i) v[i] += y.v[i]; return *this; }
    Jet& operator-=(const Jet& y) { a -= y.a; for (int i = 0; i < N; ++i) v[i] -= y.v[i]; return *this; }
};

#define MC_JET template <typename T, int N> inline Jet<T, N>
MC_JET operator+(const Jet<T, N>& f) { return f; }
MC_JET operator-(const Jet<T, N>& f) { Jet<T, N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
MC_JET operator+(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
MC_JET operator+(const Jet<T, N>& f, T s) { Jet<T, N> h = f; h.a = f.a + s; return h; }
MC_JET operator+(T s, const Jet<T, N>& f) { Jet<T, N> h = f; h.a = f.a + s; return h; }
MC_JET operator-(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
MC_JET operator-(const Jet<T, N>& f, T s) { Jet<T, N> h = f; h.a = f.a - s; return h; }
MC_JET operator-(T s, const Jet<T, N>& f) { Jet<T, N> h; h.a = s - f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
MC_JET operator*(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
MC_JET operator*(const Jet<T, N>& f, T s) { Jet<T, N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
MC_JET operator*(T s, const Jet<T, N>& f) { Jet<T, N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
// jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
MC_JET operator/(const Jet<T, N>& f, const Jet<T, N>& g) {
    Jet<T, N> h; const T gi = T(1.0) / g.a; const T fg = f.a * gi; h.a = fg;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h; }
MC_JET operator/(T s, const Jet<T, N>& g) { Jet<T, N> h; h.a = s / g.a; const T m = -s / (g.a * g.a); for (int i = 0; i < N; ++i) h.v[i] = g.v[i] * m; return h; }
MC_JET operator/(const Jet<T, N>& f, T s) { Jet<T, N> h; const T si = T(1.0) / s; h.a = f.a * si; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * si; return h; }
MC_JET sqrt(const Jet<T, N>& f) { Jet<T, N> h; const T t = std::sqrt(f.a); const T two_a_inverse = T(1.0) / (T(2.0) * t); h.a = t; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * two_a_inverse; return h; }
MC_JET cos(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::cos(f.a); const T m = -std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = m * f.v[i]; return h; }
MC_JET sin(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::sin(f.a); const T m = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = m * f.v[i]; return h; }
#undef MC_JET

// comparisons look at the scalar part only
#define MC_CMP(op) \
    template <typename T, int N> inline bool operator op(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a op g.a; } \
    template <typename T, int N> inline bool operator op(const Jet<T, N>& f, const T& s) { return f.a op s; }            \
    template <typename T, int N> inline bool operator op(const T& s, const Jet<T, N>& g) { return s op g.a; }
MC_CMP(<) MC_CMP(<=) MC_CMP(>) MC_CMP(>=) MC_CMP(==) MC_CMP(!=)
#undef MC_CMP

inline bool IsNaN(double x) { return std::isnan(x); }
inline bool IsInfinite(double x) { return std::isinf(x); }
template <typename T, int N> inline bool IsNaN(const Jet<T, N>& f) { if (std::isnan(f.a)) return true; for (int i = 0; i < N; ++i) if (std::isnan(f.v[i])) return true; return false; }
template <typename T, int N> inline bool IsFinite(const Jet<T, N>& f) { if (!std::isfinite(f.a)) return false; for (int i = 0; i < N; ++i) if (!std::isfinite(f.v[i])) return false; return true; }
template <typename T, int N> inline bool IsInfinite(const Jet<T, N>& f) { return !IsFinite(f); }

// rotation.h
template <typename T>
inline void AngleAxisRotatePoint(const T angle_axis[3], const T pt[3], T result[3]) {
    using std::sqrt; using std::cos; using std::sin;
    const T theta2 = angle_axis[0] * angle_axis[0] + angle_axis[1] * angle_axis[1] + angle_axis[2] * angle_axis[2];
    if (theta2 > T(std::numeric_limits<double>::epsilon())) {
        const T theta = sqrt(theta2);
        const T costheta = cos(theta);
        const T sintheta = sin(theta);
        const T theta_inverse = T(1.0) / theta;
        const T w[3] = {angle_axis[0] * theta_inverse, angle_axis[1] * theta_inverse, angle_axis[2] * theta_inverse};
        const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
        const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
        result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
        result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
        result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
    } else {
        const T w_cross_pt[3] = {angle_axis[1] * pt[2] - angle_axis[2] * pt[1], angle_axis[2] * pt[0] - angle_axis[0] * pt[2],
                                 angle_axis[0] * pt[1] - angle_axis[1] * pt[0]};
        result[0] = pt[0] + w_cross_pt[0];
        result[1] = pt[1] + w_cross_pt[1];
        result[2] = pt[2] + w_cross_pt[2];
    }
}

// cubic_interpolation.h (kDataDimension = 1 is all the reference uses)
template <typename T, int kDataDimension = 1, bool kRowMajor = true, bool kInterleaved = true>
struct Grid2D {
    enum { DATA_DIMENSION = kDataDimension };
    Grid2D(const T* data, int row_begin, int row_end, int col_begin, int col_end)
        : data_(data), row_begin_(row_begin), row_end_(row_end), col_begin_(col_begin), col_end_(col_end),
          num_rows_(row_end - row_begin), num_cols_(col_end - col_begin) {}
    void GetValue(int r, int c, double* f) const {
        const int row_idx = (std::min)((std::max)(row_begin_, r), row_end_ - 1) - row_begin_;
        const int col_idx = (std::min)((std::max)(col_begin_, c), col_end_ - 1) - col_begin_;
        const int n = kRowMajor ? num_cols_ * row_idx + col_idx : num_rows_ * col_idx + row_idx;
        f[0] = static_cast<double>(data_[n]);
    }
    const T* data_; int row_begin_, row_end_, col_begin_, col_end_, num_rows_, num_cols_;
};

inline void CubicHermiteSpline1(double p0, double p1, double p2, double p3, double x, double* f, double* dfdx) {
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double c = 0.5 * (-p0 + p2);
    const double d = p1;
    if (f) *f = d + x * (c + x * (b + x * a));                 // Horner
    if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}

template <typename Grid>
struct BiCubicInterpolator {
    explicit BiCubicInterpolator(const Grid& grid) : grid_(grid) {}
    void Evaluate(double r, double c, double* f, double* dfdr, double* dfdc) const {
        const int row = (int)std::floor(r);
        const int col = (int)std::floor(c);
        double fr[4], dfr[4];
        for (int k = 0; k < 4; ++k) {
            double p0, p1, p2, p3;
            grid_.GetValue(row - 1 + k, col - 1, &p0); grid_.GetValue(row - 1 + k, col, &p1);
            grid_.GetValue(row - 1 + k, col + 1, &p2); grid_.GetValue(row - 1 + k, col + 2, &p3);
            CubicHermiteSpline1(p0, p1, p2, p3, c - col, &fr[k], &dfr[k]);
        }
        CubicHermiteSpline1(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
        if (dfdc) CubicHermiteSpline1(dfr[0], dfr[1], dfr[2], dfr[3], r - row, dfdc, nullptr);
    }
    void Evaluate(const double& r, const double& c, double* f) const { Evaluate(r, c, f, nullptr, nullptr); }
    template <typename JetT>
    void Evaluate(const JetT& r, const JetT& c, JetT* f) const {
        double frc, dfdr, dfdc;
        Evaluate(r.a, c.a, &frc, &dfdr, &dfdc);
        f[0].a = frc;
        for (size_t i = 0; i < sizeof(f[0].v) / sizeof(f[0].v[0]); ++i) f[0].v[i] = dfdr * r.v[i] + dfdc * c.v[i];
    }
    const Grid& grid_;
};

}  // namespace ceres

#include "mini_ceres_solver.hpp"
