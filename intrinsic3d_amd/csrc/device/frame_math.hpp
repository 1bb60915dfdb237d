// Per-keyframe constants from a pose vector (angle-axis | translation), ONE definition for the host (the assembled point: assemble(), context.cpp) and
// for the device (the candidate point of every LM attempt: k_cand_frames, lm_kernels.hip — the host no longer sits between the PCG solve and the cost
// evaluation of its candidate).
//   R(omega) exactly as ceres::AngleAxisRotatePoint applies it [Ceres 2.1.0 rotation.h, not in /root/reference; used at cost.h:84], with its three partial
//   derivatives by forward-mode duals over omega -> the right Jacobian Jr of the rotation (pose columns of the Eg rows);
//   math::poseVecAAToMat (math.cpp:151-163) = Eigen::AngleAxisd(|w|, w/|w|).matrix() [Eigen, not in /root/reference] in float for the observation pass.
// Host and device differ in the last bits of sin / cos (libm vs the device library); both are values of the same fp64 formulas.
#pragma once
#include "common.hpp"
#include <cmath>

namespace i3d {
namespace fm {

struct D3 { double a, v[3]; };
__host__ __device__ inline D3 mk(double a) { return {a, {0, 0, 0}}; }
__host__ __device__ inline D3 operator+(D3 x, D3 y) { return {x.a + y.a, {x.v[0] + y.v[0], x.v[1] + y.v[1], x.v[2] + y.v[2]}}; }
__host__ __device__ inline D3 operator-(D3 x, D3 y) { return {x.a - y.a, {x.v[0] - y.v[0], x.v[1] - y.v[1], x.v[2] - y.v[2]}}; }
__host__ __device__ inline D3 operator*(D3 x, D3 y) { return {x.a * y.a, {x.a * y.v[0] + x.v[0] * y.a, x.a * y.v[1] + x.v[1] * y.a, x.a * y.v[2] + x.v[2] * y.a}}; }
__host__ __device__ inline D3 operator/(D3 x, D3 y) { const double gi = 1.0 / y.a, q = x.a * gi; return {q, {(x.v[0] - q * y.v[0]) * gi, (x.v[1] - q * y.v[1]) * gi, (x.v[2] - q * y.v[2]) * gi}}; }
__host__ __device__ inline D3 dsqrt(D3 x) { const double t = sqrt(x.a), h = 1.0 / (2.0 * t); return {t, {x.v[0] * h, x.v[1] * h, x.v[2] * h}}; }
__host__ __device__ inline D3 dsin(D3 x) { const double c = cos(x.a); return {sin(x.a), {c * x.v[0], c * x.v[1], c * x.v[2]}}; }
__host__ __device__ inline D3 dcos(D3 x) { const double s = -sin(x.a); return {cos(x.a), {s * x.v[0], s * x.v[1], s * x.v[2]}}; }

__host__ __device__ inline void rotate_dual(const double aa[3], const double pt[3], D3 out[3]) {
    D3 w[3]; for (int i = 0; i < 3; ++i) { w[i] = mk(aa[i]); w[i].v[i] = 1.0; }
    const D3 p[3] = {mk(pt[0]), mk(pt[1]), mk(pt[2])};
    const D3 th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2.a > 2.220446049250313e-16 /* std::numeric_limits<double>::epsilon() */) {
        const D3 th = dsqrt(th2), ct = dcos(th), stn = dsin(th), ti = mk(1.0) / th;
        const D3 k[3] = {w[0] * ti, w[1] * ti, w[2] * ti};
        const D3 kxp[3] = {k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0]};
        const D3 tmp = (k[0] * p[0] + k[1] * p[1] + k[2] * p[2]) * (mk(1.0) - ct);
        for (int i = 0; i < 3; ++i) out[i] = p[i] * ct + kxp[i] * stn + k[i] * tmp;
    } else {
        const D3 wxp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
        for (int i = 0; i < 3; ++i) out[i] = p[i] + wxp[i];
    }
}
// math::poseVecAAToMat (math.cpp:151-163): Eigen::AngleAxisd(|w|, w/|w|).matrix()
__host__ __device__ inline void pose_to_mat_eigen(const double p[6], double R[9]) {
    const double n2 = p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]);       // Vec3::norm(): halving reduction
    const double angle = sqrt(n2);
    double ax[3] = {p[0], p[1], p[2]};
    if (n2 > 0.0) { ax[0] /= angle; ax[1] /= angle; ax[2] /= angle; }
    const double s = sin(angle), c = cos(angle);
    const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const double ca[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
    double tmp;
    tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}

// the pose-dependent fields of a FrameConst (image pointers and sizes are the caller's)
__host__ __device__ inline void frame_from_pose(const double p[6], FrameConst& fc) {
    double dR[3][9];
    for (int col = 0; col < 3; ++col) {         // columns of R and dR/dw_i = rotation of the basis vectors
        double e[3] = {0, 0, 0}; e[col] = 1.0;
        D3 o[3]; rotate_dual(p, e, o);
        for (int row = 0; row < 3; ++row) { fc.hot.R[3 * row + col] = o[row].a; for (int i = 0; i < 3; ++i) dR[i][3 * row + col] = o[row].v[i]; }
    }
    for (int i = 0; i < 3; ++i) {               // Jr e_i = vee(R^T dR_i) (skew part; exact for a rotation matrix)
        double Sk[9];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double v = 0.0; for (int k = 0; k < 3; ++k) v += fc.hot.R[3 * k + a] * dR[i][3 * k + b]; Sk[3 * a + b] = v; }
        fc.hot.Jr[0 * 3 + i] = (float)(0.5 * (Sk[7] - Sk[5])); fc.hot.Jr[1 * 3 + i] = (float)(0.5 * (Sk[2] - Sk[6])); fc.hot.Jr[2 * 3 + i] = (float)(0.5 * (Sk[3] - Sk[1]));
    }
    for (int i = 0; i < 3; ++i) fc.hot.t[i] = p[3 + i];
    fc.hot.pad = 0;
    double Re[9]; pose_to_mat_eigen(p, Re);
    for (int i = 0; i < 9; ++i) fc.Rf[i] = (float)Re[i];
    for (int i = 0; i < 3; ++i) fc.tf[i] = (float)p[3 + i];
}

}  // namespace fm
}  // namespace i3d
