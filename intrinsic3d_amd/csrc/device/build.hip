// K2/K3/K7 — residual + Jacobian build of the gradient-based shading term Eg, regulariser row flags, and the
// residual-only cost evaluation at a candidate point.
//
// Replaces ShadingCost::create + ShadingCost::operator() + ceres::DynamicAutoDiffCostFunction<ShadingCost,4>
// (refinement/shading_cost.cpp:59-150, shading_cost.h:85-198) and the row bookkeeping of
// Optimizer::addVoxelResiduals (optimizer.cpp:176-282).  The 29 partials of a row are ANALYTIC (chain rule through
// normal -> iso-projection -> rotation -> Brown distortion -> Catmull-Rom bicubic -> SH shading -> ||.||2); the CPU
// oracle differentiates the same functor with dual numbers, and tests/ compares the two.
//
// Work mapping: one lane per ACTIVE voxel.  Everything that does not depend on the keyframe (4 normals, their
// projectors, 4 SH shadings and their partials, 4 iso-points) is computed once and shared by the voxel's <= `slots`
// rows.  Per row the work is split in two phases so that few values are live at once (occupancy):
//   phase 1 (fp64, the VALUE path): Q = R P + t, projection with distortion, bicubic luminance at the 4 stencil points
//            -> residual r = ||grad B - grad I||.  fp64 because r is a difference of differences of O(1) quantities and
//            the reference computes it in fp64 (parity bar 1e-4).
//   phase 2 (fp32, the DERIVATIVE path): with c_j = d r / d E_j fixed, every partial is linear in per-point terms and is
//            accumulated point by point; the rotation columns use  d(RP)/d omega = -R [P]x Jr  (Jr per keyframe), so the
//            per-point work is one cross product instead of three 3x3 products.
// The 16 bicubic taps of a point are 4 unaligned float4 loads (interior) instead of 16 scalar gathers.
#include <cstdlib>
#include <type_traits>
#include "kernels.hpp"
#include "reduce_device.hpp"

namespace i3d {

// The row set (1.46 GB on the bench workload) is written once per outer iteration and read much later: non-temporal stores, so that it does not
// displace the keyframe images (245 MB, re-read by every wave) and the voxel state from the last-level cache.
typedef float v4f_row __attribute__((ext_vector_type(4)));
typedef float v2f_row __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
static __device__ inline void st_row2(float2* p, float a, float b) { v2f_row v; v.x = a; v.y = b; __builtin_nontemporal_store(v, reinterpret_cast<v2f_row*>(p)); }
static __device__ inline void st_row(float4* p, float a, float b, float c, float d) { v4f_row v; v.x = a; v.y = b; v.z = c; v.w = d; __builtin_nontemporal_store(v, reinterpret_cast<v4f_row*>(p)); }

// 4 consecutive taps of one image row as ONE 16-byte load from a 4-byte aligned address.  The pointer is re-typed to the global
// address space: it reaches the kernel through a per-keyframe struct staged in LDS, which makes it a FLAT pointer for the compiler,
// and flat accesses are neither widened nor allowed to be misaligned (64 flat_load_dword per row instead of 16 global_load_dwordx4).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) f4u* gf4u_ptr;
typedef const __attribute__((address_space(1))) float* gf_ptr;

constexpr int Q_LDS_STRIDE = 296;      // 4 x 72 B + 8: 8-byte aligned, 74 words -> at most 2-way bank conflicts across a wave
struct PointShared {   // per stencil point j in {000,100,010,001}; value path fp64, derivative path fp32
    double P[3];       // iso-projected world position
    double B;          // albedo * (l . H(n))
    float n[3];        // unit normal (or raw gradient if its length is 0, operators.h:79-84)
    float inv_len;     // 1/|g| or 0 when |g| == 0 (then dn/dg = I)
    float s;           // sdf at the point
    float Ls;          // l . H(n)
    float dLs[3];      // d(l.H)/dn
    float alb;
};

// 1 / x on the fp64 VALUE path: v_rcp_f64 + two Newton steps (~40 cycles) instead of the correctly rounded division sequence (68 cycles, ten
// instructions; tools/experiments/valu_rate.hip).  Within 1-2 ulp — the value path is held to 1e-4 of the reference, discrete decisions do not
// go through here — and x = 0 / Inf / NaN still end in a NaN or out-of-range coordinate, i.e. in a dropped row, as with the division.
static __device__ inline double rcp64(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

static __device__ inline void shared_point(PointShared& q, double s, double sx, double sy, double sz, double alb, const float sh[9],
                                           int cx, int cy, int cz, double vs) {
    double g0 = sx - s, g1 = sy - s, g2 = sz - s;
    const double len = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
    // (one reciprocal instead of the reference's three divisions: 1e-16 relative on a value path held to 1e-4; an fp64 division is 67 cycles here)
    if (len > 0.0) { const double il = rcp64(len); q.inv_len = (float)il; g0 *= il; g1 *= il; g2 *= il; } else q.inv_len = 0.0f;
    q.s = (float)s; q.alb = (float)alb;
    q.P[0] = (double)cx * vs - g0 * s; q.P[1] = (double)cy * vs - g1 * s; q.P[2] = (double)cz * vs - g2 * s;
    const double nx = g0, ny = g1, nz = g2;
    // shading.h:53-67 basis order: 1, ny, nz, nx, nx*ny, ny*nz, -nx^2-ny^2+2nz^2, nx*nz, nx^2-ny^2
    const double Ls = (double)sh[0] + (double)sh[1] * ny + (double)sh[2] * nz + (double)sh[3] * nx + (double)sh[4] * (nx * ny) + (double)sh[5] * (ny * nz) +
                      (double)sh[6] * ((-(nx * nx)) - (ny * ny) + 2.0 * (nz * nz)) + (double)sh[7] * (nx * nz) + (double)sh[8] * ((nx * nx) - (ny * ny));
    q.B = alb * Ls; q.Ls = (float)Ls;
    const float fx = (float)nx, fy = (float)ny, fz = (float)nz;
    q.n[0] = fx; q.n[1] = fy; q.n[2] = fz;
    q.dLs[0] = sh[3] + sh[4] * fy - 2.0f * sh[6] * fx + sh[7] * fz + 2.0f * sh[8] * fx;
    q.dLs[1] = sh[1] + sh[4] * fx + sh[5] * fz - 2.0f * sh[6] * fy - 2.0f * sh[8] * fy;
    q.dLs[2] = sh[2] + sh[5] * fy + 4.0f * sh[6] * fz + sh[7] * fx;
}

// v -> (I - n n^T) v / |g|   (or v when |g| == 0)
static __device__ inline void apply_normal_jac(const PointShared& q, const float v[3], float out[3]) {
    if (q.inv_len == 0.0f) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    const float d = q.n[0] * v[0] + q.n[1] * v[1] + q.n[2] * v[2];
    out[0] = (v[0] - q.n[0] * d) * q.inv_len; out[1] = (v[1] - q.n[1] * d) * q.inv_len; out[2] = (v[2] - q.n[2] * d) * q.inv_len;
}

// [Ceres 2.1.0 cubic_interpolation.h] CubicHermiteSpline
template <class T> static __device__ inline T hermite_val(T p0, T p1, T p2, T p3, T x) {
    const T a = (T)0.5 * (-p0 + (T)3.0 * p1 - (T)3.0 * p2 + p3);
    const T b = (T)0.5 * ((T)2.0 * p0 - (T)5.0 * p1 + (T)4.0 * p2 - p3);
    const T c = (T)0.5 * (-p0 + p2);
    return p1 + x * (c + x * (b + x * a));
}
template <class T> static __device__ inline T hermite_der(T p0, T p1, T p2, T p3, T x) {
    const T a = (T)0.5 * (-p0 + (T)3.0 * p1 - (T)3.0 * p2 + p3);
    const T b = (T)0.5 * ((T)2.0 * p0 - (T)5.0 * p1 + (T)4.0 * p2 - p3);
    const T c = (T)0.5 * (-p0 + p2);
    return c + x * ((T)2.0 * b + (T)3.0 * a * x);
}
// BiCubicInterpolator::Evaluate(r = v, c = u) on a clamped Grid2D<float,1> (cost.h:108-127): value in fp64, derivatives in fp32.
// Split in two so that the tap loads of ALL four stencil points of a row are in flight together (one memory round trip per row instead of
// four dependent ones: the kernel runs two waves per SIMD and is latency-bound, not issue-bound): bicubic_taps only issues the 4 x 16 B
// loads, bicubic_eval consumes them.
struct Taps { float4 t[4]; double xc, xr; unsigned map; };
// min(max(x, 0), hi) in ONE instruction (the compiler forms v_med3_i32 only when it can prove 0 <= hi)
static __device__ inline int clamp0(int x, int hi) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi)); return r; }
// BRANCH-FREE since round 4: always ONE 16-byte load per image row, from a start column clamped so that the four floats lie inside the row; where the 4 x 4 window
// crosses the left / right border (Grid2D clamps the column, cost.h:108-127), `map` records which loaded float each tap is (byte j = index of tap j) and
// bicubic_eval folds the spline weights accordingly.  The two-path form it replaces (x4 load in the interior, four clamped scalar loads at the border, merged into the
// same registers) made the compiler put `s_waitcnt vmcnt(2)` in front of every load — a load may not be issued into registers an older load is still writing — so the
// taps of several points were never really in flight together.  Images narrower than 4 pixels: the load runs into the next row / the 16 bytes of padding every
// luminance image is allocated with (context.cpp), and `map` only points at columns of this row.
constexpr unsigned TAPS_IDENTITY = 0x03020100u;
static __device__ inline void bicubic_taps(const float* __restrict__ img, int w, int h, double r, double c, Taps& o) {
    const int row = (int)floor(r), col = (int)floor(c);
    o.xc = c - (double)col; o.xr = r - (double)row;
    const int col0 = clamp0(col - 1, max(w - 4, 0));
    unsigned map = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) map |= (unsigned)(clamp0(col - 1 + j, w - 1) - col0) << (8 * j);
    o.map = map;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = clamp0(row - 1 + i, h - 1);
        const f4u v = *(gf4u_ptr)((gf_ptr)(img + (size_t)rr * w) + col0);
        o.t[i] = make_float4(v.x, v.y, v.z, v.w);
    }
}
// spline weights of the four taps -> weights of the four LOADED floats (border columns are clamped onto the same float: their weights add up)
template <class T> static __device__ inline void fold_weights(unsigned map, T w[4]) {
    if (map == TAPS_IDENTITY) return;
    T m[4] = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned idx = (map >> (8 * j)) & 0xffu;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] += (idx == (unsigned)k) ? w[j] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = m[k];
}
// Catmull-Rom in WEIGHT form.  Ceres evaluates p1 + x (c + x (b + x a)) with a, b, c recombined from the taps for every spline (5 splines per
// point, ~15 operations each); the same cubic as a weighted sum of the taps shares ONE weight vector per axis between the four row splines and
// the column spline: f = sum_i wr_i (sum_j wc_j t_ij).  Values differ from the Horner form by fp64 round-off only (1e-16 relative).
//   w0 = x(-1 + x(2 - x))/2   w1 = 1 + x^2(3x - 5)/2   w2 = x(1 + x(4 - 3x))/2   w3 = x^2(x - 1)/2
//   w0' = (-1 + x(4 - 3x))/2  w1' = x(9x - 10)/2       w2' = (1 + x(8 - 9x))/2   w3' = x(3x - 2)/2
template <class T> static __device__ inline void cr_weights(T x, T w[4]) {
    const T h = (T)0.5, x2 = x * x;
    w[0] = h * x * ((T)-1.0 + x * ((T)2.0 - x));
    w[2] = h * x * ((T)1.0 + x * ((T)4.0 - (T)3.0 * x)); w[3] = h * x2 * (x - (T)1.0);
    w[1] = (T)1.0 - (w[0] + w[2] + w[3]);            // the four weights sum to 1 (= 1 + x^2 (3x - 5) / 2)
}
template <class T> static __device__ inline void cr_dweights(T x, T d[4]) {
    const T h = (T)0.5;
    d[0] = h * ((T)-1.0 + x * ((T)4.0 - (T)3.0 * x)); d[1] = h * x * ((T)9.0 * x - (T)10.0);
    d[2] = h * ((T)1.0 + x * ((T)8.0 - (T)9.0 * x)); d[3] = h * x * ((T)3.0 * x - (T)2.0);
}
template <bool WITH_J>
static __device__ inline void bicubic_eval(const Taps& k, double& f, float& dfdr, float& dfdc) {
    const float4* t = k.t;
    double wc[4], wr[4]; cr_weights<double>(k.xc, wc); cr_weights<double>(k.xr, wr);
    fold_weights<double>(k.map, wc);
    double fr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fr[i] = wc[0] * (double)t[i].x + wc[1] * (double)t[i].y + wc[2] * (double)t[i].z + wc[3] * (double)t[i].w;
    f = wr[0] * fr[0] + wr[1] * fr[1] + wr[2] * fr[2] + wr[3] * fr[3];
    if (WITH_J) {
        // The derivative weights of a spline sum to ZERO, and where the image is smooth the taps agree to three or four digits: sum_j w'_j t_j in fp32 then loses those
        // digits (measured on a close-up scene: camera columns 0.3 % off the oracle's duals while the fp64 residuals agreed to 1e-8).  Applied to DIFFERENCES against
        // tap 1 — sum_{j != 1} w'_j (t_j - t_1), exact subtractions of neighbouring fp32 pixels — the result is good to fp32 round-off of the derivative itself.
        // (Folded border weights still sum to zero.)
        float dwc[4], dwr[4]; cr_dweights<float>((float)k.xc, dwc); cr_dweights<float>((float)k.xr, dwr);
        fold_weights<float>(k.map, dwc);
        dfdr = dwr[0] * (float)(fr[0] - fr[1]) + dwr[2] * (float)(fr[2] - fr[1]) + dwr[3] * (float)(fr[3] - fr[1]);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += (float)wr[i] * (dwc[0] * (t[i].x - t[i].y) + dwc[2] * (t[i].z - t[i].y) + dwc[3] * (t[i].w - t[i].y));
        dfdc = acc;
    }
}

struct PointVal {      // what phase 1 leaves for phase 2 (fp32)
    float x0, y0, iz, dfdr, dfdc;
};

// camera.h:96-116 (always distorts; y uses the distorted x) + cost.h:80-127: pixel coordinates (u, v) of P.  Returns false if the projection leaves the image.
template <bool WITH_J>
static __device__ inline bool project_point(const double P[3], const double R[9], const double t[3], const OptParams& p, double& u, double& v, PointVal& o) {
    const double X = R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0];
    const double Y = R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1];
    const double Z = R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2];
    const double ps = p.pyr_scale;
    const double iz = rcp64(Z);
    const double x0 = X * iz, y0 = Y * iz;
    const double r2 = x0 * x0 + y0 * y0, r4 = r2 * r2, r6 = r4 * r2;
    const double dc = 1.0 + p.dist[0] * r2 + p.dist[1] * r4 + p.dist[2] * r6;
    const double xd = x0 * dc + 2.0 * p.dist[3] * x0 * y0 + p.dist[4] * (r2 + 2.0 * x0 * x0);
    const double yd = y0 * dc + 2.0 * p.dist[4] * xd * y0 + p.dist[3] * (r2 + 2.0 * y0 * y0);
    u = (p.intr[0] * ps) * xd + p.intr[2] * ps; v = (p.intr[1] * ps) * yd + p.intr[3] * ps;
    if (WITH_J) { o.x0 = (float)x0; o.y0 = (float)y0; o.iz = (float)iz; }
    // outside the image -> no row (cost.h:100-105).  NaN coordinates pass the reference's test (all its comparisons are false) and end in a NaN luminance,
    // i.e. in an invalid row: the same outcome as failing here, so ONE test written to be false for NaN serves both
    return u >= 0.0 && u <= (double)(p.w - 1) && v >= 0.0 && v <= (double)(p.h - 1);
}

// operators.cpp:142-147
static __device__ inline double sdf_to_weight(double sdf, double trunc) {
    const double a = fmin(fabs(sdf), trunc) / trunc;
    return fmin(fmax(1.0 - a, 0.01), 1.0);
}
// albedo_regularizer.cpp:50-84 (float chroma weight; NaN for black voxels -> no row)
static __device__ inline float chroma_weight(uchar4 c, uchar4 cn) {
    const float s = 1.0f / 255.0f;
    const float lum = 0.299f * (float)c.x + 0.587f * (float)c.y + 0.114f * (float)c.z;
    const float lumn = 0.299f * (float)cn.x + 0.587f * (float)cn.y + 0.114f * (float)cn.z;
    const float a0 = ((float)c.x * s) / lum - ((float)cn.x * s) / lumn;
    const float a1 = ((float)c.y * s) / lum - ((float)cn.y * s) / lumn;
    const float a2 = ((float)c.z * s) / lum - ((float)cn.z * s) / lumn;
    const float d = sqrtf(a0 * a0 + (a1 * a1 + a2 * a2))      /* Vec3f::norm(): Eigen's halving reduction a0 + (a1 + a2) */;
    const float one_minus = 1.0f - d;
    // std::max(1-d, 0.01f) keeps a NaN first argument
    return (one_minus < 0.01f) ? 0.01f : one_minus;
}

// FR_LDS: the per-keyframe constants (144 B each) of ALL keyframes are staged in LDS once per workgroup — every row reads
// R, t (and Jr) of its keyframe, and with them in global memory those wave-divergent gathers keep the texture-address unit busy.
template <bool WITH_J, bool FR_LDS, int BATCH, bool PIPE = false> __global__ void k_build(GridView g, RowView r, OptParams p, const FrameConst* __restrict__ frames, double* cost_out, const double* __restrict__ cam9, const LmState* __restrict__ lm);
#ifdef I3D_BUILD_VALUE_PROBE
// Timing probe (variant build only, never shipped): the VALUE half of a two-kernel split of k_build<true> — projection, taps, spline values AND derivatives, residual, the four
// coefficients — storing what a derivative kernel would need (4 x {x0, y0, 1/z, dfdr, dfdc}, c_0..3, residual, weight, keyframe: 112 B per row, in the row buffer's own coalesced
// planes) instead of assembling the 29 partials.  I3D_BUILD_VALUE_PROBE = waves per SIMD the compiler may aim for (2: point records in LDS as shipped; 3 / 4: in registers, keyframe
// constants in LDS as in the cost kernel).  The rows it leaves are NOT rows: only the kernel's own duration means anything in such a run.
constexpr int BUILD_PROBE_WAVES = I3D_BUILD_VALUE_PROBE;
#else
constexpr int BUILD_PROBE_WAVES = 0;
#endif
template <bool WITH_J, bool FR_LDS, int BATCH, bool PIPE>
__global__ void __launch_bounds__(256, WITH_J ? (BUILD_PROBE_WAVES ? BUILD_PROBE_WAVES : 2) : (PIPE ? 3 : 4)) k_build(GridView g, RowView r, OptParams p, const FrameConst* __restrict__ frames, double* cost_out,
                                                               const double* __restrict__ cam9, const LmState* __restrict__ lm) {
    extern __shared__ double frame_lds_raw[];
    if (!WITH_J) {
        if (lm && lm->done) return;               // cost of a candidate nobody will look at (the attempt was queued before the host knew that the solve had ended)
        // camera of the evaluated point from device memory (uniform address: scalar loads into the registers the by-value copy would occupy)
        if (cam9) { for (int i = 0; i < 4; ++i) p.intr[i] = cam9[i]; for (int i = 0; i < 5; ++i) p.dist[i] = cam9[4 + i]; }
    }
    FrameHot* const flds = reinterpret_cast<FrameHot*>(frame_lds_raw);
    if (FR_LDS) {
        constexpr int WORDS = sizeof(FrameHot) / 8;
        for (int i = threadIdx.x; i < p.K * WORDS; i += blockDim.x) {
            const int f = i / WORDS, w = i - f * WORDS;
            frame_lds_raw[(size_t)f * WORDS + w] = reinterpret_cast<const double*>(&frames[f].hot)[w];
        }
        __syncthreads();
    }
#ifdef I3D_COST_MULTI_PROBE
    // Timing probe (variant build only, never shipped): what ONE launch for the B candidates of a ladder batch could cost.  The cost kernel's grid is B times as large; the
    // B workgroups of a voxel block evaluate the SAME candidate (a stand-in for B nearby candidate points: same voxel records, same observation lists, image taps within a pixel of
    // each other), land on the same XCD (workgroup ids that agree modulo 8) within 8 B consecutive ids, and only the first of them contributes to the cost — the run itself stays the
    // normal run, launch for launch, with a cost kernel that does B times the work.
    int bx = blockIdx.x; bool probe_extra = false;
#ifdef I3D_COST_MULTI_PROBE_FAR      // control: candidate-major ids — the B workgroups of a voxel block run a whole candidate apart, as B separate launches would
    if (!WITH_J) { constexpr int PB = I3D_COST_MULTI_PROBE; const int per = (int)gridDim.x / PB; probe_extra = (int)blockIdx.x / per != 0; bx = (int)blockIdx.x % per; }
#else
    if (!WITH_J) { constexpr int PB = I3D_COST_MULTI_PROBE; const int grp = (int)blockIdx.x / (8 * PB), rr = (int)blockIdx.x % (8 * PB); probe_extra = rr / 8 != 0; bx = grp * 8 + (rr % 8); }
#endif
    const int ci = bx * blockDim.x + threadIdx.x;
#else
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
#endif
    const int a = ci < r.nC ? (r.clist ? r.clist[ci] : ci) : -1;      // compute list of this rank (identity when not sharded)
    const bool owned = a >= r.own0 && a < r.own1;                     // cost / weight sums count every row once: on its owner
    double cost = 0.0;
    if (a >= 0 && (WITH_J || owned)) {
        const int N = g.N; const size_t Acap = r.Acap;
        const int s = r.alist[a];
        const uint8_t fl = r.aflags[a];
        // (cost) what the entry stored at assembly — flags of its regulariser rows, row count, Ea weights — is indexed by the entry alone: requested with its list slot, not where used
        uint8_t pre_rf = 0, pre_eafree = 0, pre_nrows = 0; float eaw[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (!WITH_J) {
            pre_rf = r.regflags[a]; pre_eafree = r.ea_free[a]; pre_nrows = r.nrows[a];
#pragma unroll
            for (int d = 0; d < 6; ++d) eaw[d] = r.ea_w[(size_t)d * Acap + a];
        }
        __builtin_amdgcn_sched_barrier(0);     // (every load above is requested before the first of them is waited for)
        if (!(fl & F_ACTIVE)) {                 // free-only entry: unknowns but no rows
            if (WITH_J) {
                r.regflags[a] = 0; r.ea_free[a] = 0; r.nrows[a] = 0;
                for (int d = 0; d < 6; ++d) r.ea_w[(size_t)d * Acap + a] = 0.0f;
            }
        } else {
        int idx[P_VOX];
        bool eligible = true;
#pragma unroll
        for (int c = 0; c < 10; ++c) { const int nb = slot_fwd_nbr(c); idx[c] = nb < 0 ? s : g.nbr[(size_t)nb * N + s]; eligible &= idx[c] >= 0; }
        idx[10] = idx[0]; idx[11] = idx[6]; idx[12] = idx[1]; idx[13] = idx[4];

        // ---- regulariser rows (optimizer.cpp:238-276) ---------------------------------------------------------
        int ring[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) ring[d] = g.nbr[(size_t)d * N + s];
        const bool ring_ok = (fl & F_RING) != 0;
        const double xs = g.x_sdf[s];
        uint8_t rf = 0;
        if (WITH_J) {
            // the ring's flags, visit ranks and colours: 18 gathers requested TOGETHER and unconditionally (an entry without a full ring reads its own voxel).  Six rounds of
            // `flags -> rank -> colour` behind conditions were six to twelve exposed round trips per entry at two waves per SIMD.
            uint8_t nfl[6]; int nrk[6]; uchar4 ncol[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) { const int nb = ring_ok ? ring[d] : s; nfl[d] = g.flags[nb]; nrk[d] = g.rank[nb]; ncol[d] = g.color[nb]; }
            const uchar4 col = g.color[s];
            const int myrank = g.rank[s];
            const double sdf0s = g.sdf0[s];
            if (p.use_er && ring_ok) {
                rf |= 1;
                bool fr = (fl & F_FREE_SDF) != 0;
#pragma unroll
                for (int d = 0; d < 6; ++d) fr |= (nfl[d] & F_FREE_SDF) != 0;
                if (fr) rf |= 8;
            }
            if (p.use_es) { rf |= 2; if ((xs - sdf0s) != 0.0) rf |= 4; if (fl & F_FREE_SDF) rf |= 16; }
            uint8_t eafree = 0;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                float w = 0.0f;
                if (p.use_ea && ring_ok) {
                    const bool added_before = (nfl[d] & F_ACTIVE) && nrk[d] < myrank;     // voxels_added, optimizer.cpp:267-279
                    if (!added_before) {
                        w = chroma_weight(col, ncol[d]);
                        if (!(w == w) || isinf(w)) w = 0.0f;
                        if (w != 0.0f && ((fl & F_FREE_ALB) || (nfl[d] & F_FREE_ALB))) eafree |= (uint8_t)(1 << d);
                    }
                }
                r.ea_w[(size_t)d * Acap + a] = w;
            }
            r.regflags[a] = rf; r.ea_free[a] = eafree;
        } else {
            // cost of the regulariser rows at this state (rows without a free parameter are not part of the reduced program).  Everything they read is requested TOGETHER and
            // unconditionally (an entry without a full ring reads its own voxel): behind their conditions the seven gathers were seven exposed round trips per entry.
            rf = pre_rf;
            const uint8_t eafree = pre_eafree;
            const double sdf0s = g.sdf0[s], xa = g.x_alb[s];
            double xsr[6], xar[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) { const int nb = ring_ok ? ring[d] : s; xsr[d] = g.x_sdf[nb]; xar[d] = g.x_alb[nb]; }
            if ((rf & 1) && (rf & 8)) {
                const double dxx = xsr[0] + xsr[1] - 2.0 * xs, dyy = xsr[2] + xsr[3] - 2.0 * xs, dzz = xsr[4] + xsr[5] - 2.0 * xs;
                const double lap = dxx + dyy + dzz; cost += 0.5 * p.type_w[1] * lap * lap;
            }
            if ((rf & 2) && (rf & 16)) { double e = xs - sdf0s; if (e == 0.0) e = 0.0000001; cost += 0.5 * p.type_w[2] * e * e; }
#pragma unroll
            for (int d = 0; d < 6; ++d) if (eafree & (1 << d)) {
                const double e = xa - xar[d];
                cost += 0.5 * (double)eaw[d] * p.type_w[3] * e * e;
            }
        }

        // ---- Eg rows ---------------------------------------------------------------------------------------
        const int nin = WITH_J ? r.slots : (int)pre_nrows;        // candidates: observation slots (assembly) or stored rows (cost)
        bool any_row = false;
        if (WITH_J) {          // (unrolled, unconditional loads: as a run-time loop every slot was a load and a wait of its own)
#pragma unroll
            for (int k = 0; k < MAX_SLOTS; ++k) { const float w = r.obs_w[(size_t)min(k, r.slots - 1) * Acap + a]; any_row |= k < r.slots && w > 0.0f; }
            any_row &= eligible;
        }
        else any_row = nin > 0;
        int nout = 0;
        if (any_row) {
            double sd[10];
#pragma unroll
            for (int c = 0; c < 10; ++c) sd[c] = g.x_sdf[idx[c]];
            float sh[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) sh[j] = g.sh[(size_t)j * N + s];
            const int cx = g.cx[s], cy = g.cy[s], cz = g.cz[s];
            const double vs = (double)g.voxel_size;
            // WITH_J: the 4 point records (72 B each) live in LDS (one 296-byte slot per lane), not in registers: the Jacobian variant is
            // register-bound (256 VGPR + AGPR spills = one wave per SIMD) and only reads them field by field
            constexpr bool Q_LDS = WITH_J && BUILD_PROBE_WAVES <= 2;
            PointShared qreg[Q_LDS ? 1 : 4];
            PointShared* q = Q_LDS ? reinterpret_cast<PointShared*>(reinterpret_cast<char*>(frame_lds_raw) + (FR_LDS ? (size_t)p.K * sizeof(FrameHot) : 0) + (size_t)threadIdx.x * Q_LDS_STRIDE) : qreg;
            // sdf slots: 0:000 1:010 2:020 3:011 4:001 5:002 6:100 7:110 8:101 9:200 (shading_cost.h:88-97)
            // the four albedos (and, at assembly, the 14 flag bytes of the stencil) are requested with the sdf values, in front of the fp64 point records they would otherwise wait behind
            double albv[4] = {0.0, 0.0, 0.0, 0.0}; uint8_t vfl[P_VOX];
#pragma unroll
            for (int j = 0; j < 4; ++j) albv[j] = g.x_alb[idx[10 + j]];
            if (WITH_J) {
#pragma unroll
                for (int c = 0; c < P_VOX; ++c) vfl[c] = g.flags[idx[c]];
            }
            shared_point(q[0], sd[0], sd[6], sd[1], sd[4], albv[0], sh, cx, cy, cz, vs);
            shared_point(q[1], sd[6], sd[9], sd[7], sd[8], albv[1], sh, cx + 1, cy, cz, vs);
            shared_point(q[2], sd[1], sd[7], sd[2], sd[3], albv[2], sh, cx, cy + 1, cz, vs);
            shared_point(q[3], sd[4], sd[8], sd[3], sd[5], albv[3], sh, cx, cy, cz + 1, vs);
            bool vox_free = false;
            if (WITH_J) {
#pragma unroll
                for (int c = 0; c < 10; ++c) vox_free |= (vfl[c] & F_FREE_SDF) != 0;
#pragma unroll
                for (int c = 10; c < 14; ++c) vox_free |= (vfl[c] & F_FREE_ALB) != 0;
                vox_free |= !p.fix_poses || !p.fix_intr || !p.fix_dist;
            }
            const double weight_sdf = sdf_to_weight(xs, (double)g.truncation);
            // slot of (s, sx, sy, sz) for each point
            constexpr int PS[4][4] = {{0, 6, 1, 4}, {6, 9, 7, 8}, {1, 7, 2, 3}, {4, 8, 3, 5}};
            const float psf = (float)p.pyr_scale;
            const float fxs = (float)(p.intr[0] * p.pyr_scale), fys = (float)(p.intr[1] * p.pyr_scale);
            const float k0 = (float)p.dist[0], k1 = (float)p.dist[1], k2 = (float)p.dist[2], k3 = (float)p.dist[3], k4 = (float)p.dist[4];

            if (!WITH_J && PIPE) {
                // ---- candidate cost, software-pipelined (round 4) --------------------------------------------------------------------------------------------
                // The row loop below exposes three dependent memory round trips per row to its wave (keyframe tag -> taps of points 0,1 -> taps of points 2,3); at four
                // waves per SIMD two thirds of the resident wave-cycles sat in s_waitcnt (profiles/r04_sq_counters.json).  Here the keyframe tags of ALL rows of the voxel
                // are requested a row ahead, and the projection + tap loads of one stencil point are always issued TWO points ahead of the
                // evaluation that consumes them — across the row boundary too — so a wave has the taps of two points in flight while it evaluates.  Same arithmetic per
                // point; a point that leaves the image still has its (sanitised) taps loaded and its row dropped as before.
                auto tag_of = [&](int k) -> unsigned { return (unsigned)__float_as_int(r.row_jt()[row_jt_index(a, k, r.slots)].y); };
                // `after`: a value the projection must wait for (an empty asm ties the point's x to it) — without it the compiler starts the next point's projection before the
                // evaluation that frees its tap registers, and keeps three sets of taps alive
                auto issue = [&](const FrameHot& fc, const PointShared& pt, Taps& t, double after) -> bool {
                    double u, v; PointVal unused;
                    double Pt[3] = {pt.P[0], pt.P[1], pt.P[2]};
                    asm volatile("" : "+v"(Pt[0]) : "v"(after));
                    const bool in = project_point<false>(Pt, fc.R, fc.t, p, u, v, unused);
                    if (!in) { u = 0.0; v = 0.0; }                // (NaN / far outside: keep the tap addresses inside the image; the row is dropped)
                    bicubic_taps(fc.lum, p.w, p.h, v, u, t);
                    return in;
                };
                // the free bit is a property of the voxel (vox_free at assembly): either every row of it is part of the reduced program or none
                const unsigned tag0 = tag_of(0);
                if (tag0 & (unsigned)ROW_FREE_BIT) {
                    Taps T0, T1;
                    int fcur = (int)(tag0 & ~(unsigned)ROW_FREE_BIT);
                    const double dB1 = q[1].B - q[0].B, dB2 = q[2].B - q[0].B, dB3 = q[3].B - q[0].B;
                    bool ok = issue(FR_LDS ? flds[fcur] : frames[fcur].hot, q[0], T0, 0.0);
                    ok = issue(FR_LDS ? flds[fcur] : frames[fcur].hot, q[1], T1, 0.0) && ok;
                    // one row: evaluate its four points while the next two are requested.  MORE (compile time) = another row follows: its first two points are requested
                    // during this row's last two evaluations.  The last row is peeled off instead of testing `k + 1 < nin` inside: after a conditional load the compiler
                    // cannot count what is in flight and falls back to draining everything.
                    auto row = [&](int k, auto more_tag) {
                        constexpr bool MORE = decltype(more_tag)::value;
                        // requested now, needed after the second / fourth evaluation (loads complete in order: they are there when the taps issued after them are)
                        const float roww = r.row_wr[row_scalar_index(a, k, r.slots)].x;
                        const unsigned tagn = MORE ? tag_of(k + 1) : 0u;
                        const FrameHot& fc = FR_LDS ? flds[fcur] : frames[fcur].hot;
                        double l0, l1, l2, l3; float u0, u1;
                        bicubic_eval<false>(T0, l0, u0, u1); ok = issue(fc, q[2], T0, l0) && ok;
                        bicubic_eval<false>(T1, l1, u0, u1); ok = issue(fc, q[3], T1, l1) && ok;
                        const int fnext = MORE ? (int)(tagn & ~(unsigned)ROW_FREE_BIT) : fcur;
                        const FrameHot& fn = FR_LDS ? flds[fnext] : frames[fnext].hot;
                        bool okn = true;
                        bicubic_eval<false>(T0, l2, u0, u1); if (MORE) okn = issue(fn, q[0], T0, l2);
                        bicubic_eval<false>(T1, l3, u0, u1); if (MORE) okn = issue(fn, q[1], T1, l3) && okn;
                        if (ok) {
                            const double d1 = dB1 - (l1 - l0), d2 = dB2 - (l2 - l0), d3 = dB3 - (l3 - l0);
                            const double res = sqrt(d1 * d1 + d2 * d2 + d3 * d3);
                            if (res > 0.0 && !isinf(res)) cost += 0.5 * (double)roww * p.type_w[0] * res * res;      // 0, NaN, inf -> NV_INVALID_RESIDUAL (shading_cost.h:186-195)
                        }
                        ok = okn; fcur = fnext;
                    };
#pragma unroll 1
                    for (int k = 0; k + 1 < nin; ++k) row(k, std::true_type{});
                    row(nin - 1, std::false_type{});
                }
            } else
            {
            // (assembly) the observation slot of row k + 1 is requested while row k is evaluated: its keyframe index is the first link of a row's chain of dependent loads
            // (slot -> keyframe constants -> taps of points 0,1 -> taps of points 2,3).  Unconditional (the last row re-reads its own slot): a conditional load would make
            // the compiler drain every load in flight at the join.
            float ow_next = 0.0f; int f_next = 0;
            if (WITH_J) { ow_next = r.obs_w[a]; f_next = r.obs_frame[a]; }
            for (int k = 0; k < nin; ++k) {
                float roww; int f;
                if (WITH_J) { const float ow = ow_next; f = f_next;
                              { const size_t kn = (size_t)min(k + 1, nin - 1) * Acap + a; ow_next = r.obs_w[kn]; f_next = r.obs_frame[kn]; }
                              roww = (ow > 0.0f) ? (float)((double)ow * weight_sdf) : 0.0f; }
                else { const size_t ro = row_scalar_index(a, k, r.slots); const int fb = __float_as_int(r.row_jt()[row_jt_index(a, k, r.slots)].y); roww = (fb & ROW_FREE_BIT) ? r.row_wr[ro].x : 0.0f; f = fb & ~ROW_FREE_BIT; }
                if (roww == 0.0f) continue;
                const FrameHot& fc = FR_LDS ? flds[f] : frames[f].hot;
                // (assembly: keyframe constants in global memory) the image pointer is requested HERE, in front of R and t: left to the scheduler it is requested where it is first
                // used — behind the projections — and the taps wait a round trip of their own for it
                const float* const lum_img = WITH_J ? fc.lum : nullptr;
                // ... and R, t are read ONCE per row: left alone, the second pair of points and the fp32 copy of R for the partials each read them again — a round trip each
                double Rl[WITH_J ? 9 : 1], tl[WITH_J ? 3 : 1];
                if (WITH_J) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rl[i] = fc.R[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) tl[i] = fc.t[i];
                }
                if (WITH_J && !FR_LDS) __builtin_amdgcn_sched_barrier(0);
                // ---- phase 1: values (fp64) ----
                double lum[4]; PointVal pv[4];
                bool ok = true;
#pragma unroll
                for (int j0 = 0; j0 < 4; j0 += BATCH) {
                    if (!ok) break;             // a row with a point outside the image is dropped (cost.h:100-105)
                    double pu[BATCH], pw[BATCH];
#pragma unroll
                    for (int j = 0; j < BATCH; ++j) ok = project_point<WITH_J>(q[j0 + j].P, WITH_J ? Rl : fc.R, WITH_J ? tl : fc.t, p, pu[j], pw[j], pv[j0 + j]) && ok;
                    if (ok) {
                        Taps tp[BATCH];
#pragma unroll
                        for (int j = 0; j < BATCH; ++j) bicubic_taps(WITH_J ? lum_img : fc.lum, p.w, p.h, pw[j], pu[j], tp[j]);
#pragma unroll
                        for (int j = 0; j < BATCH; ++j) bicubic_eval<WITH_J>(tp[j], lum[j0 + j], pv[j0 + j].dfdr, pv[j0 + j].dfdc);
                    }
                }
                double res = 0.0; float cj[4] = {0, 0, 0, 0};
                if (ok) {
                    const double d1 = (q[1].B - q[0].B) - (lum[1] - lum[0]), d2 = (q[2].B - q[0].B) - (lum[2] - lum[0]), d3 = (q[3].B - q[0].B) - (lum[3] - lum[0]);
                    res = sqrt(d1 * d1 + d2 * d2 + d3 * d3);
                    if (!(res > 0.0) || isinf(res)) { ok = false; res = 0.0; }    // 0, NaN, inf -> NV_INVALID_RESIDUAL (shading_cost.h:186-195)
                    else { const double ir = rcp64(res); cj[1] = (float)(d1 * ir); cj[2] = (float)(d2 * ir); cj[3] = (float)(d3 * ir); cj[0] = -(float)((d1 + d2 + d3) * ir); }
                }
                if (!WITH_J) { if (ok) cost += 0.5 * (double)roww * p.type_w[0] * res * res; continue; }
                if (!ok) continue;                                                 // dropped at creation (shading_cost.cpp:136-145)
                // rows are stored with the row weight folded in (Js = sqrt(w) J, common.hpp RowView).  Every partial below is linear in the four
                // coefficients c_j, so the fold costs 4 multiplications here instead of 29 at the store
                { const float sw = sqrtf(roww); cj[0] *= sw; cj[1] *= sw; cj[2] *= sw; cj[3] *= sw; }
#ifdef I3D_BUILD_VALUE_PROBE
                {
                    float I[28];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { I[5 * j] = pv[j].x0; I[5 * j + 1] = pv[j].y0; I[5 * j + 2] = pv[j].iz; I[5 * j + 3] = pv[j].dfdr; I[5 * j + 4] = pv[j].dfdc; }
                    I[20] = cj[0]; I[21] = cj[1]; I[22] = cj[2]; I[23] = cj[3]; I[24] = (float)res; I[25] = roww; I[26] = __int_as_float(f | (vox_free ? ROW_FREE_BIT : 0)); I[27] = 0.0f;
#pragma unroll
                    for (int gq = 0; gq < 7; ++gq) st_row(&r.rows[row_index(a, nout, gq, r.slots)], I[4 * gq], I[4 * gq + 1], I[4 * gq + 2], I[4 * gq + 3]);
                    ++nout;
                    continue;
                }
#endif
                // ---- phase 2: partials (fp32), accumulated point by point ----
                // (keyframe constants in global memory) the right Jacobian of the rotation is requested here and used at the end of the phase: requested where it is used it
                // cost every row a round trip of its own
                float jr[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) jr[i] = fc.Jr[i];
                if (WITH_J && !FR_LDS) __builtin_amdgcn_sched_barrier(0);
                float J[P_TOTAL];
#pragma unroll
                for (int i = 0; i < P_TOTAL; ++i) J[i] = 0.0f;
                float Wx = 0.0f, Wy = 0.0f, Wz = 0.0f;                              // sum_j c_j (P_j x M_j): rotation part before Jr
                const float R0 = (float)Rl[0], R1 = (float)Rl[1], R2 = (float)Rl[2], R3 = (float)Rl[3], R4 = (float)Rl[4], R5 = (float)Rl[5], R6 = (float)Rl[6], R7 = (float)Rl[7], R8 = (float)Rl[8];
                // Two stencil points per pass in PACKED fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes of arithmetic per instruction, the
                // fp32 half of this kernel's issue time): the chain below is the same for every point, only the scatter into J is per point
#pragma unroll
                for (int jp = 0; jp < 4; jp += 2) {
                    const int ja = jp, jb = jp + 1;
                    const f2 x0 = {pv[ja].x0, pv[jb].x0}, y0 = {pv[ja].y0, pv[jb].y0}, iz = {pv[ja].iz, pv[jb].iz};
                    const f2 dfdc = {pv[ja].dfdc, pv[jb].dfdc}, dfdr = {pv[ja].dfdr, pv[jb].dfdr};
                    const f2 r2 = x0 * x0 + y0 * y0, r4 = r2 * r2, r6 = r4 * r2;
                    const f2 dc = 1.0f + k0 * r2 + k1 * r4 + k2 * r6;
                    const f2 dcr = k0 + 2.0f * k1 * r2 + 3.0f * k2 * r4;          // d dc / d r2
                    const f2 xd = x0 * dc + 2.0f * k3 * x0 * y0 + k4 * (r2 + 2.0f * x0 * x0);
                    const f2 yd = y0 * dc + 2.0f * k4 * xd * y0 + k3 * (r2 + 2.0f * y0 * y0);
                    const f2 dxd_dx0 = dc + 2.0f * x0 * x0 * dcr + 2.0f * k3 * y0 + 6.0f * k4 * x0;
                    const f2 dxd_dy0 = 2.0f * x0 * y0 * dcr + 2.0f * k3 * x0 + 2.0f * k4 * y0;
                    const f2 dyd_dx0 = 2.0f * x0 * y0 * dcr + 2.0f * k4 * y0 * dxd_dx0 + 2.0f * k3 * x0;
                    const f2 dyd_dy0 = dc + 2.0f * y0 * y0 * dcr + 2.0f * k4 * (xd + y0 * dxd_dy0) + 6.0f * k3 * y0;
                    const f2 au = dfdc * fxs, av = dfdr * fys;
                    const f2 lx = au * dxd_dx0 + av * dyd_dx0, ly = au * dxd_dy0 + av * dyd_dy0;
                    const f2 L0 = lx * iz, L1 = ly * iz, L2 = -(lx * x0 + ly * y0) * iz;      // d lum / d Q
                    const f2 M0 = L0 * R0 + L1 * R3 + L2 * R6, M1 = L0 * R1 + L1 * R4 + L2 * R7, M2 = L0 * R2 + L1 * R5 + L2 * R8;   // d lum / d P
                    const f2 c = {cj[ja], cj[jb]};
                    // E_j = B_j - lum_j;  dE/dg = alb * N dLs + s * N M,  dE/ds (direct) = M . n
                    const f2 alb = {q[ja].alb, q[jb].alb}, sv = {q[ja].s, q[jb].s}, il = {q[ja].inv_len, q[jb].inv_len};
                    const f2 n0 = {q[ja].n[0], q[jb].n[0]}, n1 = {q[ja].n[1], q[jb].n[1]}, n2 = {q[ja].n[2], q[jb].n[2]};
                    const f2 v0 = alb * f2{q[ja].dLs[0], q[jb].dLs[0]} + sv * M0, v1 = alb * f2{q[ja].dLs[1], q[jb].dLs[1]} + sv * M1, v2 = alb * f2{q[ja].dLs[2], q[jb].dLs[2]} + sv * M2;
                    // v -> (I - n n^T) v / |g|   (or v when |g| == 0: apply_normal_jac)
                    const f2 d = n0 * v0 + n1 * v1 + n2 * v2;
                    f2 G0 = (v0 - n0 * d) * il, G1 = (v1 - n1 * d) * il, G2 = (v2 - n2 * d) * il;
                    if (il.x == 0.0f) { G0.x = v0.x; G1.x = v1.x; G2.x = v2.x; }
                    if (il.y == 0.0f) { G0.y = v0.y; G1.y = v1.y; G2.y = v2.y; }
                    const f2 direct = M0 * n0 + M1 * n1 + M2 * n2;
                    const f2 cG0 = c * G0, cG1 = c * G1, cG2 = c * G2, cD = c * (direct - (G0 + G1 + G2)), cLs = c * f2{q[ja].Ls, q[jb].Ls};
                    J[PS[ja][1]] += cG0.x; J[PS[ja][2]] += cG1.x; J[PS[ja][3]] += cG2.x; J[PS[ja][0]] += cD.x; J[P_ALB + ja] = cLs.x;
                    J[PS[jb][1]] += cG0.y; J[PS[jb][2]] += cG1.y; J[PS[jb][3]] += cG2.y; J[PS[jb][0]] += cD.y; J[P_ALB + jb] = cLs.y;
                    // pose: d lum/d t = L ; d lum/d omega = (P x M)^T Jr
                    const f2 cL0 = c * L0, cL1 = c * L1, cL2 = c * L2;
                    J[P_POSE + 3] -= cL0.x + cL0.y; J[P_POSE + 4] -= cL1.x + cL1.y; J[P_POSE + 5] -= cL2.x + cL2.y;
                    const f2 Px = {(float)q[ja].P[0], (float)q[jb].P[0]}, Py = {(float)q[ja].P[1], (float)q[jb].P[1]}, Pz = {(float)q[ja].P[2], (float)q[jb].P[2]};
                    const f2 wx = c * (Py * M2 - Pz * M1), wy = c * (Pz * M0 - Px * M2), wz = c * (Px * M1 - Py * M0);
                    Wx += wx.x + wx.y; Wy += wy.x + wy.y; Wz += wz.x + wz.y;
                    const f2 cdc = c * dfdc * psf, cdr = c * dfdr * psf;
                    const f2 i0 = cdc * xd, i1 = cdr * yd;
                    J[P_INTR + 0] -= i0.x + i0.y; J[P_INTR + 1] -= i1.x + i1.y; J[P_INTR + 2] -= cdc.x + cdc.y; J[P_INTR + 3] -= cdr.x + cdr.y;
                    const f2 dxk0 = x0 * r2, dxk1 = x0 * r4, dxk2 = x0 * r6, dxk3 = 2.0f * x0 * y0, dxk4 = r2 + 2.0f * x0 * x0;
                    const f2 c2 = 2.0f * k4 * y0;
                    const f2 e0 = c * (au * dxk0 + av * (y0 * r2 + c2 * dxk0)), e1 = c * (au * dxk1 + av * (y0 * r4 + c2 * dxk1)), e2 = c * (au * dxk2 + av * (y0 * r6 + c2 * dxk2));
                    const f2 e3 = c * (au * dxk3 + av * (c2 * dxk3 + (r2 + 2.0f * y0 * y0))), e4 = c * (au * dxk4 + av * (2.0f * xd * y0 + c2 * dxk4));
                    J[P_DIST + 0] -= e0.x + e0.y; J[P_DIST + 1] -= e1.x + e1.y; J[P_DIST + 2] -= e2.x + e2.y; J[P_DIST + 3] -= e3.x + e3.y; J[P_DIST + 4] -= e4.x + e4.y;
                }
                J[P_POSE + 0] = -(Wx * jr[0] + Wy * jr[3] + Wz * jr[6]);
                J[P_POSE + 1] = -(Wx * jr[1] + Wy * jr[4] + Wz * jr[7]);
                J[P_POSE + 2] = -(Wx * jr[2] + Wy * jr[5] + Wz * jr[8]);
                // all 29 partials finite?  0 * x is 0 for a finite x and NaN for Inf / NaN: one fma per partial instead of a class test + mask merge each
                float finz = 0.0f;
#pragma unroll
                for (int i = 0; i < P_TOTAL; ++i) finz = __builtin_fmaf(J[i], 0.0f, finz);
                if (!(finz == 0.0f)) continue;
                // rows of a voxel are compacted into its first slots (creation order = ascending observation weight)
#pragma unroll
                for (int gq = 0; gq < 7; ++gq)
                    st_row(&r.rows[row_index(a, nout, gq, r.slots)], J[4 * gq], J[4 * gq + 1], J[4 * gq + 2], J[4 * gq + 3]);
                const size_t ro = row_scalar_index(a, nout, r.slots);
                st_row2(&r.row_jt()[row_jt_index(a, nout, r.slots)], J[28], __int_as_float(f | (vox_free ? ROW_FREE_BIT : 0)));
                st_row2(&r.row_wr[ro], roww, (float)res);
                ++nout;
            }
            }
        }
        if (WITH_J) {
            r.nrows[a] = (uint8_t)nout;
        }
        }
    }
#ifdef I3D_COST_MULTI_PROBE
    if (!WITH_J && probe_extra) cost = 0.0;
#endif
    if (!WITH_J) block_partial_d(cost, cost_out, 1, 0);          // per-workgroup partial (no same-address atomics), summed by k_reduce_partials
}

void launch_build(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames, bool with_jacobian, double* cost_out, double* scratch,
                  const double* cam9, const LmState* lm) {
    if (r.nC <= 0) { if (!with_jacobian && cost_out) (void)hipMemsetAsync(cost_out, 0, sizeof(double), st); return; }      // (a rank without rows contributes 0 to the all-reduced cost: the slot is ASSIGNED below, not zeroed by the caller)
#ifdef I3D_COST_MULTI_PROBE
    const int blocks = with_jacobian ? (r.nC + 255) / 256 : (((r.nC + 255) / 256 + 7) / 8) * 8 * I3D_COST_MULTI_PROBE;
#else
    const int blocks = (r.nC + 255) / 256;
#endif
    double* const cost_dst = cost_out; cost_out = scratch;       // the kernels write per-workgroup partials
    const size_t lds = (size_t)p.K * sizeof(FrameHot), qlds = (size_t)256 * Q_LDS_STRIDE;
    if (with_jacobian) {
        // point records in LDS (74 KB per workgroup), per-keyframe constants from global memory: 247 VGPRs, no spills, TWO workgroups per CU.
        // (Measured: 2.13 -> 1.45 ms; with the keyframe constants staged in LDS as well only one workgroup fits and nothing is gained.)
#ifdef I3D_BUILD_VALUE_PROBE
        if (BUILD_PROBE_WAVES > 2) {           // (variant build only) point records in registers, keyframe constants in LDS like the cost kernel
            if (!set_dynamic_lds((const void*)k_build<true, true, 2>, "k_build<true> value probe", lds, p.K)) return;
            k_build<true, true, 2><<<blocks, 256, lds, st>>>(g, r, p, frames, cost_out, nullptr, nullptr);
            return;
        }
#endif
        if (!set_dynamic_lds((const void*)k_build<true, false, 2>, "k_build<true>", qlds, p.K)) return;
        k_build<true, false, 2><<<blocks, 256, qlds, st>>>(g, r, p, frames, cost_out, nullptr, nullptr);
    } else {
        static const bool no_pipe = [] { const char* e = std::getenv("I3D_COST_NOPIPE"); return e && e[0] == '1'; }();      // A/B runs
        if (lds <= 48 * 1024) {
            if (no_pipe) k_build<false, true, 2><<<blocks, 256, lds, st>>>(g, r, p, frames, cost_out, cam9, lm);       // (tap loads of 2 points in flight: 1 -> +3 %, 4 spills at 128 registers -> +87 %)
            else k_build<false, true, 2, true><<<blocks, 256, lds, st>>>(g, r, p, frames, cost_out, cam9, lm);
        } else if (no_pipe) k_build<false, false, 2><<<blocks, 256, 0, st>>>(g, r, p, frames, cost_out, cam9, lm);
        else k_build<false, false, 2, true><<<blocks, 256, 0, st>>>(g, r, p, frames, cost_out, cam9, lm);
    }
    if (!with_jacobian) launch_reduce_partials(st, scratch, blocks, 1, cost_dst, nullptr, true);      // *cost_dst = sum (no memset in front of every candidate)
}

// nls_solver.cpp:379-394 — per-type sums of the row weights (sums[0..3]) and row counts (sums[4..7]); [8] active voxels.
// COST, [9..12]: sum of w r^2 per row type at the state the rows were built at, over the rows of the reduced program (at least one free parameter) — the cost the trust-region
// loop starts from is 0.5 sum_t type_w[t] sums[9 + t] (the type weights are only known once sums[0..3] are: nls_solver.cpp:379-394), which saves the residual-only pass
// k_build<false> at the unchanged point (a sharded run, and I3D_GRADCOL=0; on one rank the sums ride on the gradient pass, gradcol.hip).  Eg residuals are the stored ones (row_wr.y, fp32-rounded: 2e-11 of the cost on 1e7 rows); the regulariser residuals are
// recomputed in fp64 exactly as k_build<false> does.
template <bool COST>
__global__ void __launch_bounds__(256) k_weight_sums(RowView r, GridView g, double* partials) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, n0 = 0, n3 = 0, na = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const int N = g.N;
    for (int a = r.own0 + blockIdx.x * blockDim.x + threadIdx.x; a < r.own1 && a < r.A; a += gridDim.x * blockDim.x) {      // owned range only
        if (!(r.aflags[a] & F_ACTIVE)) continue;
        na += 1.0;
        const int nr = r.nrows[a];
        for (int k = 0; k < nr; ++k) {
            const float2 wr = r.row_wr[row_scalar_index(a, k, r.slots)]; const float w = wr.x;
            s0 += (double)w; if (w != 0.0f) n0 += 1.0;
            if (COST && (__float_as_int(r.row_jt()[row_jt_index(a, k, r.slots)].y) & ROW_FREE_BIT)) c0 += (double)w * ((double)wr.y * (double)wr.y);
        }
        const uint8_t rf = r.regflags[a];
        if (rf & 1) s1 += 1.0;
        if (rf & 2) s2 += 1.0;
        const int s = COST ? r.alist[a] : 0;
        const uint8_t eafree = COST ? r.ea_free[a] : 0;
        const bool er = COST && (rf & 1) && (rf & 8), es = COST && (rf & 2) && (rf & 16);
        int ring[6];
        if (er || eafree) {
#pragma unroll
            for (int d = 0; d < 6; ++d) ring[d] = g.nbr[(size_t)d * N + s];
        }
        if (er || es) {
            const double xs = g.x_sdf[s];
            if (er) {
                const double dxx = g.x_sdf[ring[0]] + g.x_sdf[ring[1]] - 2.0 * xs, dyy = g.x_sdf[ring[2]] + g.x_sdf[ring[3]] - 2.0 * xs, dzz = g.x_sdf[ring[4]] + g.x_sdf[ring[5]] - 2.0 * xs;
                const double lap = dxx + dyy + dzz; c1 += lap * lap;
            }
            if (es) { double e = xs - g.sdf0[s]; if (e == 0.0) e = 0.0000001; c2 += e * e; }      // surface_stab_regularizer.h:62-64
        }
        const double xa = eafree ? g.x_alb[s] : 0.0;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const float w = r.ea_w[(size_t)d * r.Acap + a]; s3 += (double)w; if (w != 0.0f) n3 += 1.0;
            if (eafree & (1 << d)) { const double e = xa - g.x_alb[ring[d]]; c3 += (double)w * (e * e); }
        }
    }
    block_partial_d(s0, partials, 13, 0); block_partial_d(s1, partials, 13, 1); block_partial_d(s2, partials, 13, 2); block_partial_d(s3, partials, 13, 3);
    block_partial_d(n0, partials, 13, 4); block_partial_d(0.0, partials, 13, 5); block_partial_d(0.0, partials, 13, 6); block_partial_d(n3, partials, 13, 7);
    block_partial_d(na, partials, 13, 8);
    block_partial_d(c0, partials, 13, 9); block_partial_d(c1, partials, 13, 10); block_partial_d(c2, partials, 13, 11); block_partial_d(c3, partials, 13, 12);
}
void launch_weight_sums(hipStream_t st, RowView r, GridView g, bool with_cost, double* sums13, double* scratch) {
    const int n = r.own1 - r.own0;
    if (n <= 0) return;
    int blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
    if (with_cost) k_weight_sums<true><<<blocks, 256, 0, st>>>(r, g, scratch);
    else k_weight_sums<false><<<blocks, 256, 0, st>>>(r, g, scratch);
    launch_reduce_partials(st, scratch, blocks, 13, sums13, nullptr);
}

}  // namespace i3d
