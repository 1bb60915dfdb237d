// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the small part of Eigen 3 [un-vendored dependency of /root/reference, absent from this image] that the reference
// bodies compiled into oracle/_ref use: fixed-size column vectors / matrices with element access, +, -, scalar *, /, cast<>,
// Constant, Zero, Ones, Identity, dot, cross, norm, normalized, normalize, isZero, products, corner / row blocks, a 4x4 inverse,
// AngleAxisd::matrix(), the comma initialiser, and dynamic VectorXd / VectorXf.  Written from Eigen's documented semantics:
//   * sum reductions of fixed-size expressions are completely unrolled by halving (Eigen/src/Core/Redux.h, redux_novec_unroller):
//     a 3-element sum is  a0 + (a1 + a2);
//   * a small fixed-size product is coefficient-based: c(i,j) = (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum(), the same
//     halving reduction (what Eigen's packet paths and its GEMV for run-time sized blocks do instead is compiler-flag dependent and
//     stays unpinned — DESIGN.md §6);
//   * normalized(): v / sqrt(squaredNorm) when squaredNorm > 0, else v;  normalize(): the same in place;
//   * isZero(prec = NumTraits::dummy_precision()): every |coeff| <= prec (1e-5 for float, 1e-12 for double);
//   * AngleAxis::toRotationMatrix (Geometry/AngleAxis.h): sin_axis = sin(a)*axis, cos1_axis = (1-cos(a))*axis, off-diagonals
//     tmp -/+ sin_axis, diagonal cos1_axis[i]*axis[i] + cos(a);
//   * scalar * VectorXd with a float scalar promotes the scalar to double first (promote_scalar_arg).
// Nothing of this is reference code.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <ostream>
#include <vector>

namespace Eigen {

template <class T> struct DummyPrec { static T value() { return T(1e-12); } };
template <> struct DummyPrec<float> { static float value() { return 1e-5f; } };
template <> struct DummyPrec<int> { static int value() { return 0; } };
template <> struct DummyPrec<unsigned char> { static unsigned char value() { return 0; } };

template <class T, int R, int C> struct Matrix;

// writable view of a sub-block (pose.topLeftCorner<3,3>() = rot; vec.topRows<3>() = ...)
template <class T, int R, int C, int BR, int BC>
struct BlockRef {
    Matrix<T, R, C>& m; int r0, c0;
    BlockRef& operator=(const Matrix<T, BR, BC>& v) { for (int c = 0; c < BC; ++c) for (int r = 0; r < BR; ++r) m(r0 + r, c0 + c) = v(r, c); return *this; }
    operator Matrix<T, BR, BC>() const { Matrix<T, BR, BC> o; for (int c = 0; c < BC; ++c) for (int r = 0; r < BR; ++r) o(r, c) = m(r0 + r, c0 + c); return o; }
    T operator()(int r, int c) const { return m(r0 + r, c0 + c); }
};

template <class T, int R, int C>
struct CommaInit {                                   // bbox << a, b, c, ...;  (row-major fill order, as Eigen's)
    Matrix<T, R, C>& m; int k;
    CommaInit& operator,(T v) { m(k / C, k % C) = v; ++k; return *this; }
};

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
    static Matrix Zero(int) { return Constant(T(0)); }
    static Matrix Ones() { return Constant(T(1)); }
    static Matrix Identity() { Matrix m = Zero(); for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
    void setZero() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
    void setZero(int) { setZero(); }                    // (fixed-size: Eigen asserts the size matches)
    T& operator[](size_t i) { return d[i]; }
    const T& operator[](size_t i) const { return d[i]; }
    T& operator()(int r, int c) { return d[c * R + r]; }
    const T& operator()(int r, int c) const { return d[c * R + r]; }
    T* data() { return d; }
    const T* data() const { return d; }
    int rows() const { return R; }
    int cols() const { return C; }
    int size() const { return R * C; }
    template <class U> Matrix<U, R, C> cast() const { Matrix<U, R, C> m; for (int i = 0; i < R * C; ++i) m.d[i] = static_cast<U>(d[i]); return m; }
    Matrix operator+(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] + o.d[i]; return m; }
    Matrix operator-(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] - o.d[i]; return m; }
    Matrix operator-() const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = -d[i]; return m; }
    Matrix operator*(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] * s; return m; }
    Matrix operator/(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] / s; return m; }
    friend Matrix operator*(T s, const Matrix& a) { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = s * a.d[i]; return m; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] = d[i] + o.d[i]; return *this; }
    Matrix& operator-=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] = d[i] - o.d[i]; return *this; }
    Matrix& operator*=(T s) { for (int i = 0; i < R * C; ++i) d[i] = d[i] * s; return *this; }
    Matrix& operator/=(T s) { for (int i = 0; i < R * C; ++i) d[i] = d[i] / s; return *this; }
    bool operator==(const Matrix& o) const { for (int i = 0; i < R * C; ++i) if (!(d[i] == o.d[i])) return false; return true; }
    bool operator!=(const Matrix& o) const { return !(*this == o); }
    // halving reduction of n terms starting at s (Redux.h)
    template <class F> static T redux(const F& term, int s, int n) { if (n == 1) return term(s); const int h = n / 2; return redux(term, s, h) + redux(term, s + h, n - h); }
    T dot(const Matrix& o) const { return redux([&](int i) { return d[i] * o.d[i]; }, 0, R * C); }
    T squaredNorm() const { return redux([&](int i) { return d[i] * d[i]; }, 0, R * C); }
    T norm() const { return std::sqrt(squaredNorm()); }
    Matrix normalized() const { const T z = squaredNorm(); if (z > T(0)) return *this / std::sqrt(z); return *this; }
    void normalize() { const T z = squaredNorm(); if (z > T(0)) *this /= std::sqrt(z); }
    bool isZero() const { const T p = DummyPrec<T>::value(); for (int i = 0; i < R * C; ++i) if (std::abs(d[i]) > p) return false; return true; }
    Matrix cross(const Matrix& o) const {
        static_assert(R * C == 3, "cross");
        return Matrix(d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]);
    }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> m; for (int c = 0; c < C; ++c) for (int r = 0; r < R; ++r) m(c, r) = (*this)(r, c); return m; }
    template <int C2> Matrix<T, R, C2> operator*(const Matrix<T, C, C2>& o) const {
        Matrix<T, R, C2> m;
        for (int c = 0; c < C2; ++c) for (int r = 0; r < R; ++r) { const Matrix& a = *this; m(r, c) = redux([&](int k) { return a(r, k) * o(k, c); }, 0, C); }
        return m;
    }
    // blocks (copies for reads, BlockRef for writes)
    template <int BR, int BC> Matrix<T, BR, BC> topLeftCorner() const { Matrix<T, BR, BC> o; for (int c = 0; c < BC; ++c) for (int r = 0; r < BR; ++r) o(r, c) = (*this)(r, c); return o; }
    template <int BR, int BC> Matrix<T, BR, BC> topRightCorner() const { Matrix<T, BR, BC> o; for (int c = 0; c < BC; ++c) for (int r = 0; r < BR; ++r) o(r, c) = (*this)(r, C - BC + c); return o; }
    template <int BR, int BC> BlockRef<T, R, C, BR, BC> topLeftCorner() { return BlockRef<T, R, C, BR, BC>{*this, 0, 0}; }
    template <int BR, int BC> BlockRef<T, R, C, BR, BC> topRightCorner() { return BlockRef<T, R, C, BR, BC>{*this, 0, C - BC}; }
    template <int BR> Matrix<T, BR, C> topRows() const { Matrix<T, BR, C> o; for (int c = 0; c < C; ++c) for (int r = 0; r < BR; ++r) o(r, c) = (*this)(r, c); return o; }
    template <int BR> BlockRef<T, R, C, BR, C> topRows() { return BlockRef<T, R, C, BR, C>{*this, 0, 0}; }
    template <int BR> BlockRef<T, R, C, BR, C> bottomRows() { return BlockRef<T, R, C, BR, C>{*this, R - BR, 0}; }
    template <int BR> Matrix<T, BR, C> bottomRows() const { Matrix<T, BR, C> o; for (int c = 0; c < C; ++c) for (int r = 0; r < BR; ++r) o(r, c) = (*this)(R - BR + r, c); return o; }
    // run-time sized corners: the reference only asks for (3,3) and (3,1) of a 4x4
    // (a run-time sized block times a vector is Eigen's GEMV: every row accumulates its columns left to right)
    struct Corner33 { Matrix<T, 3, 3> m; Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1>& v) const { Matrix<T, 3, 1> o; for (int r = 0; r < 3; ++r) o[r] = (m(r, 0) * v[0] + m(r, 1) * v[1]) + m(r, 2) * v[2]; return o; } };
    Corner33 topLeftCorner(int, int) const { return Corner33{topLeftCorner<3, 3>()}; }
    Matrix<T, 3, 1> topRightCorner(int, int) const { return topRightCorner<3, 1>(); }
    CommaInit<T, R, C> operator<<(T v) { (*this)(0, 0) = v; return CommaInit<T, R, C>{*this, 1}; }
    Matrix inverse() const {                        // general inverse by cofactors in T (the order of Eigen's SSE 4x4 kernel is not reproduced)
        static_assert(R == 4 && C == 4, "inverse: 4x4 only");
        const Matrix& a = *this; Matrix inv; T det = T(0);
        auto minor3 = [&](int r, int c) {
            int rr[3], cc[3]; for (int i = 0, k = 0; i < 4; ++i) if (i != r) rr[k++] = i; for (int i = 0, k = 0; i < 4; ++i) if (i != c) cc[k++] = i;
            return a(rr[0], cc[0]) * (a(rr[1], cc[1]) * a(rr[2], cc[2]) - a(rr[1], cc[2]) * a(rr[2], cc[1]))
                 - a(rr[0], cc[1]) * (a(rr[1], cc[0]) * a(rr[2], cc[2]) - a(rr[1], cc[2]) * a(rr[2], cc[0]))
                 + a(rr[0], cc[2]) * (a(rr[1], cc[0]) * a(rr[2], cc[1]) - a(rr[1], cc[1]) * a(rr[2], cc[0]));
        };
        for (int c = 0; c < 4; ++c) det += a(0, c) * (((c & 1) ? T(-1) : T(1)) * minor3(0, c));
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inv(c, r) = ((((r + c) & 1) ? T(-1) : T(1)) * minor3(r, c)) / det;
        return inv;
    }
};

template <class T, int R, int C> std::ostream& operator<<(std::ostream& os, const Matrix<T, R, C>& m) {
    for (int r = 0; r < R; ++r) { for (int c = 0; c < C; ++c) os << (c ? " " : "") << m(r, c); if (r + 1 < R) os << "\n"; } return os; }

typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 2, 1> Vector2f;  typedef Matrix<float, 3, 1> Vector3f;  typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<int, 2, 1> Vector2i;    typedef Matrix<int, 3, 1> Vector3i;    typedef Matrix<int, 4, 1> Vector4i;
typedef Matrix<double, 2, 2> Matrix2d; typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 2, 2> Matrix2f;  typedef Matrix<float, 3, 3> Matrix3f;  typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<int, 2, 2> Matrix2i;    typedef Matrix<int, 3, 3> Matrix3i;    typedef Matrix<int, 4, 4> Matrix4i;

struct AngleAxisd {
    double angle_; Vector3d axis_;
    AngleAxisd(double a, const Vector3d& ax) : angle_(a), axis_(ax) {}
    // AngleAxis(rotation matrix) = AngleAxis(Quaternion(matrix)) (Geometry/AngleAxis.h: fromRotationMatrix, operator=(QuaternionBase))
    template <class M3> explicit AngleAxisd(const M3& m);
    double angle() const { return angle_; }
    const Vector3d& axis() const { return axis_; }
    Matrix3d matrix() const {
        Matrix3d res;
        const Vector3d sin_axis = std::sin(angle_) * axis_;
        const double c = std::cos(angle_);
        const Vector3d cos1_axis = (1.0 - c) * axis_;
        double tmp;
        tmp = cos1_axis[0] * axis_[1]; res(0, 1) = tmp - sin_axis[2]; res(1, 0) = tmp + sin_axis[2];
        tmp = cos1_axis[0] * axis_[2]; res(0, 2) = tmp + sin_axis[1]; res(2, 0) = tmp - sin_axis[1];
        tmp = cos1_axis[1] * axis_[2]; res(1, 2) = tmp - sin_axis[0]; res(2, 1) = tmp + sin_axis[0];
        for (int i = 0; i < 3; ++i) res(i, i) = cos1_axis[i] * axis_[i] + c;
        return res;
    }
};

// Eigen::Quaternionf as Sensor::loadPoses / savePoses use it: from a rotation matrix (Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>: the
// trace branch, else the largest diagonal element), from (w, x, y, z), and toRotationMatrix (the tx = 2x ... products of QuaternionBase::toRotationMatrix)
template <class S>
struct QuaternionT {
    S c[4];                                                        // x y z w (Eigen's coefficient order)
    QuaternionT(S w, S x, S y, S z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
    template <class M3> explicit QuaternionT(const M3& mm) {
        Matrix<S, 3, 3> m; for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) m(r, k) = mm(r, k);
        S t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > S(0)) {
            t = std::sqrt(t + S(1)); c[3] = S(0.5) * t; t = S(0.5) / t;
            c[0] = (m(2, 1) - m(1, 2)) * t; c[1] = (m(0, 2) - m(2, 0)) * t; c[2] = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0; if (m(1, 1) > m(0, 0)) i = 1; if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + S(1)); c[i] = S(0.5) * t; t = S(0.5) / t;
            c[3] = (m(k, j) - m(j, k)) * t; c[j] = (m(j, i) + m(i, j)) * t; c[k] = (m(k, i) + m(i, k)) * t;
        }
    }
    S x() const { return c[0]; } S y() const { return c[1]; } S z() const { return c[2]; } S w() const { return c[3]; }
    Matrix<S, 3, 3> toRotationMatrix() const {
        Matrix<S, 3, 3> res;
        const S tx = S(2) * c[0], ty = S(2) * c[1], tz = S(2) * c[2];
        const S twx = tx * c[3], twy = ty * c[3], twz = tz * c[3], txx = tx * c[0], txy = ty * c[0], txz = tz * c[0], tyy = ty * c[1], tyz = tz * c[1], tzz = tz * c[2];
        res(0, 0) = S(1) - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = S(1) - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = S(1) - (txx + tyy);
        return res;
    }
};
typedef QuaternionT<float> Quaternionf;
typedef QuaternionT<double> Quaterniond;

template <class M3> AngleAxisd::AngleAxisd(const M3& m) {
    const Quaterniond q(m);
    double n = std::sqrt(q.x() * q.x() + (q.y() * q.y() + q.z() * q.z()));          // q.vec().norm(): fixed-size reduction a0 + (a1 + a2)
    // (below epsilon Eigen recomputes with stableNorm(): the same value up to scaling against overflow, which cannot occur for a unit quaternion)
    if (n != 0.0) {
        angle_ = 2.0 * std::atan2(n, std::abs(q.w()));
        if (q.w() < 0.0) n = -n;
        axis_[0] = q.x() / n; axis_[1] = q.y() / n; axis_[2] = q.z() / n;
    } else { angle_ = 0.0; axis_[0] = 1.0; axis_[1] = 0.0; axis_[2] = 0.0; }
}

template <class S>
struct VectorX {                                    // dynamic column vector (VectorXd / VectorXf)
    std::vector<S> v;
    VectorX() {}
    explicit VectorX(int n) : v((size_t)n, S(0)) {}
    static VectorX Zero(int n) { return VectorX(n); }
    void setZero() { for (auto& x : v) x = S(0); }
    void setZero(int n) { v.assign((size_t)n, S(0)); }
    S& operator[](size_t i) { return v[i]; }
    const S& operator[](size_t i) const { return v[i]; }
    S* data() { return v.data(); }
    const S* data() const { return v.data(); }
    size_t size() const { return v.size(); }
    template <class U> VectorX<U> cast() const { VectorX<U> o((int)v.size()); for (size_t i = 0; i < v.size(); ++i) o.v[i] = static_cast<U>(v[i]); return o; }
    bool isZero() const { const S p = DummyPrec<S>::value(); for (auto x : v) if (std::abs(x) > p) return false; return true; }
    S dot(const VectorX& o) const { S s = S(0); for (size_t i = 0; i < v.size(); ++i) s += v[i] * o.v[i]; return s; }   // (only the debug shading path; unpinned order)
    S norm() const { S s = S(0); for (auto x : v) s += x * x; return std::sqrt(s); }
    VectorX operator*(S s) const { VectorX o((int)v.size()); for (size_t i = 0; i < v.size(); ++i) o.v[i] = v[i] * s; return o; }
    friend VectorX operator*(S s, const VectorX& a) { VectorX o((int)a.v.size()); for (size_t i = 0; i < a.v.size(); ++i) o.v[i] = s * a.v[i]; return o; }
    VectorX& operator+=(const VectorX& o) { for (size_t i = 0; i < v.size(); ++i) v[i] = v[i] + o.v[i]; return *this; }
};
typedef VectorX<double> VectorXd;
typedef VectorX<float> VectorXf;

}  // namespace Eigen
