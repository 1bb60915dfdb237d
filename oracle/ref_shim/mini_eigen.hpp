// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the small part of Eigen 3 [un-vendored dependency of /root/reference, absent from this image] that the reference
// bodies compiled into oracle/_ref use: fixed-size column vectors / matrices with element access, +, -, scalar *, /, cast<>,
// Constant, Zero, dot, cross, norm, normalized, isZero, and a dynamic VectorXd.  Written from Eigen's documented semantics:
//   * sum reductions of fixed-size expressions are completely unrolled by halving (Eigen/src/Core/Redux.h, redux_novec_unroller):
//     a 3-element sum is  a0 + (a1 + a2);
//   * normalized(): v / sqrt(squaredNorm) when squaredNorm > 0, else v;
//   * isZero(prec = NumTraits::dummy_precision()): every |coeff| <= prec (1e-5 for float, 1e-12 for double).
// Nothing of this is reference code.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace Eigen {

template <class T> struct DummyPrec { static T value() { return T(1e-12); } };
template <> struct DummyPrec<float> { static float value() { return 1e-5f; } };
template <> struct DummyPrec<int> { static int value() { return 0; } };
template <> struct DummyPrec<unsigned char> { static unsigned char value() { return 0; } };

template <class T, int R, int C>
struct Matrix {
    T d[R * C];                                     // column-major
    typedef T Scalar;
    Matrix() {}                                     // uninitialised, like Eigen
    Matrix(T x, T y) { static_assert(R * C == 2, "size"); d[0] = x; d[1] = y; }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, "size"); d[0] = x; d[1] = y; d[2] = z; }
    Matrix(T x, T y, T z, T w) { static_assert(R * C == 4, "size"); d[0] = x; d[1] = y; d[2] = z; d[3] = w; }
    static Matrix Constant(T v) { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = v; return m; }
    static Matrix Zero() { return Constant(T(0)); }
    T& operator[](size_t i) { return d[i]; }
    const T& operator[](size_t i) const { return d[i]; }
    T& operator()(int r, int c) { return d[c * R + r]; }
    const T& operator()(int r, int c) const { return d[c * R + r]; }
    T* data() { return d; }
    const T* data() const { return d; }
    template <class U> Matrix<U, R, C> cast() const { Matrix<U, R, C> m; for (int i = 0; i < R * C; ++i) m.d[i] = static_cast<U>(d[i]); return m; }
    Matrix operator+(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] + o.d[i]; return m; }
    Matrix operator-(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] - o.d[i]; return m; }
    Matrix operator*(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] * s; return m; }
    Matrix operator/(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] / s; return m; }
    friend Matrix operator*(T s, const Matrix& a) { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = s * a.d[i]; return m; }
    bool operator==(const Matrix& o) const { for (int i = 0; i < R * C; ++i) if (!(d[i] == o.d[i])) return false; return true; }
    // halving reduction of n terms starting at s (Redux.h)
    template <class F> static T redux(const F& term, int s, int n) { if (n == 1) return term(s); const int h = n / 2; return redux(term, s, h) + redux(term, s + h, n - h); }
    T dot(const Matrix& o) const { return redux([&](int i) { return d[i] * o.d[i]; }, 0, R * C); }
    T squaredNorm() const { return redux([&](int i) { return d[i] * d[i]; }, 0, R * C); }
    T norm() const { return std::sqrt(squaredNorm()); }
    Matrix normalized() const { const T z = squaredNorm(); if (z > T(0)) return *this / std::sqrt(z); return *this; }
    bool isZero() const { const T p = DummyPrec<T>::value(); for (int i = 0; i < R * C; ++i) if (std::abs(d[i]) > p) return false; return true; }
    Matrix cross(const Matrix& o) const {
        static_assert(R * C == 3, "cross");
        return Matrix(d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]);
    }
};

typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 2, 1> Vector2f;  typedef Matrix<float, 3, 1> Vector3f;  typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<int, 2, 1> Vector2i;    typedef Matrix<int, 3, 1> Vector3i;    typedef Matrix<int, 4, 1> Vector4i;
typedef Matrix<double, 2, 2> Matrix2d; typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 2, 2> Matrix2f;  typedef Matrix<float, 3, 3> Matrix3f;  typedef Matrix<float, 4, 4> Matrix4f;

struct VectorXd {
    std::vector<double> v;
    VectorXd() {}
    explicit VectorXd(int n) : v((size_t)n, 0.0) {}
    double& operator[](size_t i) { return v[i]; }
    const double& operator[](size_t i) const { return v[i]; }
    size_t size() const { return v.size(); }
};

}  // namespace Eigen
