// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Forward-mode dual numbers, restating the semantics of ceres::Jet<double,N>
// [Ceres 2.1.0, not in /root/reference; pinned by README.md:75].  The reference
// differentiates every residual functor through these (shading_cost.cpp:85,
// volumetric_regularizer.cpp:67, lighting_svsh.cpp:243).  Ceres evaluates the
// 29 partials of a ShadingCost row in 8 passes of stride 4; one pass of width
// 29 is mathematically identical.
#pragma once
#include <cmath>

namespace orc {

template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
    Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT implicit like ceres
    static Jet var(double s, int k) { Jet j(s); j.v[k] = 1.0; return j; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
    Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
    // ceres: g_a_inverse = 1/g.a; f_a_by_g_a = f.a*g_a_inverse; v = (f.v - f_a_by_g_a*g.v)*g_a_inverse
    Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h; }
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }

template <int N> inline Jet<N> sqrt(const Jet<N>& f) {
    Jet<N> h; const double t = std::sqrt(f.a); const double two_a_inv = 1.0 / (2.0 * t); h.a = t;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * two_a_inv; return h; }
template <int N> inline Jet<N> sin(const Jet<N>& f) {
    Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a);
    for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline Jet<N> cos(const Jet<N>& f) {
    Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a);
    for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }

// comparisons look at the scalar part only (ceres jet.h)
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }

inline double scalar_of(double x) { return x; }
template <int N> inline double scalar_of(const Jet<N>& x) { return x.a; }

// ceres::IsNaN / IsInfinite on a Jet: true if ANY component is (cost.h:73-77 relies on it)
inline bool all_finite(double x) { return std::isfinite(x); }
template <int N> inline bool all_finite(const Jet<N>& x) {
    if (!std::isfinite(x.a)) return false;
    for (int i = 0; i < N; ++i) if (!std::isfinite(x.v[i])) return false;
    return true; }

using std::sqrt; using std::sin; using std::cos;

}  // namespace orc
