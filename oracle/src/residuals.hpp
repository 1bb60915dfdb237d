// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Restatement of the gradient-based shading residual Eg and its templated helpers.
//   refinement/shading_cost.h:85-198   ShadingCost::operator()  (parameter order, validity rules)
//   refinement/cost.h:73-127           isValid, transform, transformVoxelIso, interpolate
//   sdf/operators.h:49-86              voxelToWorld, voxelCenterToIso, computeNormal
//   camera.h:96-116                    CameraT::project (always distorts; y uses distorted x)
//   shading.h:53-148                   shBasisFunctions, computeShading, computeShadingGradientDifference
//   [Ceres 2.1.0 rotation.h, not in reference]  AngleAxisRotatePoint
#pragma once
#include <limits>
#include "jet.hpp"
#include "imaging.hpp"

namespace orc {

// parameter slots of one Eg row (shading_cost.cpp:90-129), 29 scalars in this order
enum { P_SDF = 0, P_ALB = 10, P_POSE = 14, P_INTR = 20, P_DIST = 24, P_TOTAL = 29 };
// voxel offsets of the 10 sdf slots and the 4 albedo slots (x,y,z)
static const int SDF_OFF[10][3] = {{0,0,0},{0,1,0},{0,2,0},{0,1,1},{0,0,1},{0,0,2},{1,0,0},{1,1,0},{1,0,1},{2,0,0}};
static const int ALB_OFF[4][3]  = {{0,0,0},{1,0,0},{0,1,0},{0,0,1}};

struct ShadingRowConst {     // everything that is NOT a parameter of the row
    int vx, vy, vz;          // voxel coordinates
    double sh[9];            // per-voxel SH coefficients (constant inside optimize)
    double pyr_scale;        // 2^-rgbd_level
    double voxel_size;
    int w, h; const float* lum;
};

template <class T> inline void compute_normal(const T& s, const T& sx, const T& sy, const T& sz, T n[3]) {
    n[0] = sx - s; n[1] = sy - s; n[2] = sz - s;
    T len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (len > T(0.0)) { n[0] = n[0] / len; n[1] = n[1] / len; n[2] = n[2] / len; }
}

template <class T> inline void angle_axis_rotate(const T aa[3], const T pt[3], T out[3]) {
    const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > T(std::numeric_limits<double>::epsilon())) {
        const T theta = sqrt(theta2);
        const T ct = cos(theta), st = sin(theta);
        const T ti = T(1.0) / theta;
        const T w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
        const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
        const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - ct);
        for (int i = 0; i < 3; ++i) out[i] = pt[i] * ct + wxp[i] * st + w[i] * tmp;
    } else {
        const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
        for (int i = 0; i < 3; ++i) out[i] = pt[i] + wxp[i];
    }
}

template <class T> inline void transform_voxel_iso(const T& vs, const T aa[3], const T t[3], const int c[3],
                                                   const T& sdf, const T n[3], T out[3]) {
    T p[3], piso[3];
    for (int i = 0; i < 3; ++i) p[i] = T((double)c[i]) * vs;            // operators.h:49-54
    for (int i = 0; i < 3; ++i) piso[i] = p[i] - n[i] * sdf;            // operators.h:59-66
    angle_axis_rotate(aa, piso, out);
    for (int i = 0; i < 3; ++i) out[i] = out[i] + t[i];
}

template <class T> inline bool project_T(const T& fx, const T& fy, const T& cx, const T& cy, const T* k,
                                         int w, int h, const T p[3], T p2d[2]) {
    T x = p[0] / p[2], y = p[1] / p[2];
    const T r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const T dc = T(1.0) + k[0] * r2 + k[1] * r4 + k[2] * r6;
    x = x * dc + T(2.0) * k[3] * x * y + k[4] * (r2 + T(2.0) * x * x);
    y = y * dc + T(2.0) * k[4] * x * y + k[3] * (r2 + T(2.0) * y * y);
    p2d[0] = fx * x + cx; p2d[1] = fy * y + cy;
    return !(p2d[0] < T(0.0) || p2d[0] > T((double)(w - 1)) || p2d[1] < T(0.0) || p2d[1] > T((double)(h - 1)));
}

template <class T> inline T shade(const double sh[9], const T n[3], const T& albedo) {
    T b[9];
    b[0] = T(1.0); b[1] = n[1]; b[2] = n[2]; b[3] = n[0]; b[4] = n[0] * n[1]; b[5] = n[1] * n[2];
    b[6] = (-(n[0] * n[0])) - (n[1] * n[1]) + T(2.0) * (n[2] * n[2]);
    b[7] = n[0] * n[2]; b[8] = (n[0] * n[0]) - (n[1] * n[1]);
    T s = T(0.0);
    for (int i = 0; i < 9; ++i) s += T(sh[i]) * b[i];
    return albedo * s;
}

// Evaluates one Eg row.  params[29] in slot order.  Returns the residual (0.0 == NV_INVALID_RESIDUAL).
template <class T> inline T shading_residual(const ShadingRowConst& k, const T* prm) {
    const T* s = prm + P_SDF; const T* a = prm + P_ALB;
    const T* aa = prm + P_POSE; const T* tr = prm + P_POSE + 3;
    const T ps = T(k.pyr_scale);
    const T fx = prm[P_INTR + 0] * ps, fy = prm[P_INTR + 1] * ps, cx = prm[P_INTR + 2] * ps, cy = prm[P_INTR + 3] * ps;
    const T* dist = prm + P_DIST;
    // sdf slots: 0:000 1:010 2:020 3:011 4:001 5:002 6:100 7:110 8:101 9:200
    T n0[3], n1[3], n2[3], n3[3];
    compute_normal(s[0], s[6], s[1], s[4], n0);      // voxel 000
    compute_normal(s[6], s[9], s[7], s[8], n1);      // voxel 100
    compute_normal(s[1], s[7], s[2], s[3], n2);      // voxel 010
    compute_normal(s[4], s[8], s[3], s[5], n3);      // voxel 001
    const int c0[3] = {k.vx, k.vy, k.vz}, c1[3] = {k.vx + 1, k.vy, k.vz}, c2[3] = {k.vx, k.vy + 1, k.vz}, c3[3] = {k.vx, k.vy, k.vz + 1};
    const T vs = T(k.voxel_size);
    T q0[3], q1[3], q2[3], q3[3];
    transform_voxel_iso(vs, aa, tr, c0, s[0], n0, q0);
    transform_voxel_iso(vs, aa, tr, c1, s[6], n1, q1);
    transform_voxel_iso(vs, aa, tr, c2, s[1], n2, q2);
    transform_voxel_iso(vs, aa, tr, c3, s[4], n3, q3);
    T u0[2], u1[2], u2[2], u3[2];
    const bool v0 = project_T(fx, fy, cx, cy, dist, k.w, k.h, q0, u0);
    const bool v1 = project_T(fx, fy, cx, cy, dist, k.w, k.h, q1, u1);
    const bool v2 = project_T(fx, fy, cx, cy, dist, k.w, k.h, q2, u2);
    const bool v3 = project_T(fx, fy, cx, cy, dist, k.w, k.h, q3, u3);
    if (!v0 || !v1 || !v2 || !v3) return T(0.0);
    T lum[4];
    bicubic_T(k.lum, k.w, k.h, u0[1], u0[0], &lum[0]);
    bicubic_T(k.lum, k.w, k.h, u1[1], u1[0], &lum[1]);
    bicubic_T(k.lum, k.w, k.h, u2[1], u2[0], &lum[2]);
    bicubic_T(k.lum, k.w, k.h, u3[1], u3[0], &lum[3]);
    for (int i = 0; i < 4; ++i) if (!all_finite(lum[i])) return T(0.0);
    T B[4];
    B[0] = shade(k.sh, n0, a[0]); B[1] = shade(k.sh, n1, a[1]); B[2] = shade(k.sh, n2, a[2]); B[3] = shade(k.sh, n3, a[3]);
    const T dx = (B[1] - B[0]) - (lum[1] - lum[0]);
    const T dy = (B[2] - B[0]) - (lum[2] - lum[0]);
    const T dz = (B[3] - B[0]) - (lum[3] - lum[0]);
    T r = sqrt(dx * dx + dy * dy + dz * dz);
    if (!all_finite(r)) return T(0.0);
    return r;
}

}  // namespace orc
