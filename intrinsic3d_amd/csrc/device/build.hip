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
// rows; per-keyframe rotation R and dR/d(omega) come precomputed in FrameConst (wave-uniform loads when neighbouring
// voxels see the same keyframes, L1/L2 hits otherwise).
#include "kernels.hpp"

namespace i3d {

typedef double real;   // value path precision (geometry, projection, spline).  Stored partials are fp32.

struct PointShared {   // per stencil point j in {000,100,010,001}
    real n[3];         // unit normal (or raw gradient if its length is 0, operators.h:79-84)
    real inv_len;      // 1/|g| or 0 when |g| == 0 (then dn/dg = I)
    real s;            // sdf at the point
    real P[3];         // iso-projected world position
    real Ls;           // l . H(n)
    real dLs[3];       // d(l.H)/dn
    real alb;
};

static __device__ inline void shared_point(PointShared& q, real s, real sx, real sy, real sz, real alb, const real sh[9],
                                           int cx, int cy, int cz, real vs) {
    real g0 = sx - s, g1 = sy - s, g2 = sz - s;
    const real len = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
    if (len > 0.0) { q.inv_len = 1.0 / len; g0 /= len; g1 /= len; g2 /= len; } else q.inv_len = 0.0;
    q.n[0] = g0; q.n[1] = g1; q.n[2] = g2; q.s = s; q.alb = alb;
    q.P[0] = (real)cx * vs - g0 * s; q.P[1] = (real)cy * vs - g1 * s; q.P[2] = (real)cz * vs - g2 * s;
    const real nx = g0, ny = g1, nz = g2;
    // shading.h:53-67 basis order: 1, ny, nz, nx, nx*ny, ny*nz, -nx^2-ny^2+2nz^2, nx*nz, nx^2-ny^2
    q.Ls = sh[0] + sh[1] * ny + sh[2] * nz + sh[3] * nx + sh[4] * (nx * ny) + sh[5] * (ny * nz) +
           sh[6] * ((-(nx * nx)) - (ny * ny) + 2.0 * (nz * nz)) + sh[7] * (nx * nz) + sh[8] * ((nx * nx) - (ny * ny));
    q.dLs[0] = sh[3] + sh[4] * ny - 2.0 * sh[6] * nx + sh[7] * nz + 2.0 * sh[8] * nx;
    q.dLs[1] = sh[1] + sh[4] * nx + sh[5] * nz - 2.0 * sh[6] * ny - 2.0 * sh[8] * ny;
    q.dLs[2] = sh[2] + sh[5] * ny + 4.0 * sh[6] * nz + sh[7] * nx;
}

// v -> (I - n n^T) v / |g|   (or v when |g| == 0)
static __device__ inline void apply_normal_jac(const PointShared& q, const real v[3], real out[3]) {
    if (q.inv_len == 0.0) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    const real d = q.n[0] * v[0] + q.n[1] * v[1] + q.n[2] * v[2];
    out[0] = (v[0] - q.n[0] * d) * q.inv_len; out[1] = (v[1] - q.n[1] * d) * q.inv_len; out[2] = (v[2] - q.n[2] * d) * q.inv_len;
}

// [Ceres 2.1.0 cubic_interpolation.h] CubicHermiteSpline value + derivative
static __device__ inline void hermite(real p0, real p1, real p2, real p3, real x, real& f, real& df) {
    const real a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const real b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const real c = 0.5 * (-p0 + p2);
    f = p1 + x * (c + x * (b + x * a));
    df = c + x * (2.0 * b + 3.0 * a * x);
}
// BiCubicInterpolator::Evaluate(r = v, c = u) on a clamped Grid2D<float,1> (cost.h:108-127)
static __device__ inline void bicubic(const float* __restrict__ img, int w, int h, real r, real c, real& f, real& dfdr, real& dfdc) {
    const int row = (int)floor(r), col = (int)floor(c);
    const real xc = c - (real)col, xr = r - (real)row;
    real fr[4], dc[4];
    const int c0 = min(max(col - 1, 0), w - 1), c1 = min(max(col, 0), w - 1), c2 = min(max(col + 1, 0), w - 1), c3 = min(max(col + 2, 0), w - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = min(max(row - 1 + i, 0), h - 1);
        const float* line = img + (size_t)rr * w;
        hermite((real)line[c0], (real)line[c1], (real)line[c2], (real)line[c3], xc, fr[i], dc[i]);
    }
    real dummy;
    hermite(fr[0], fr[1], fr[2], fr[3], xr, f, dfdr);
    hermite(dc[0], dc[1], dc[2], dc[3], xr, dfdc, dummy);
}

struct PointFrame {     // per (point, keyframe)
    real lum;
    real dpose[6];      // d lum / d (omega, t)
    real M[3];          // d lum / d P (world)
    real dintr[4];      // d lum / d (fx,fy,cx,cy) level-0 parameters
    real ddist[5];      // d lum / d (k1,k2,k3,p1,p2)
};

// camera.h:96-116 (always distorts; y uses the distorted x) + cost.h:80-127.  Returns false if the projection leaves the image.
template <bool WITH_J>
static __device__ inline bool eval_point(const PointShared& q, const FrameConst& fc, const OptParams& p, PointFrame& o) {
    const real X = fc.R[0] * q.P[0] + fc.R[1] * q.P[1] + fc.R[2] * q.P[2] + fc.t[0];
    const real Y = fc.R[3] * q.P[0] + fc.R[4] * q.P[1] + fc.R[5] * q.P[2] + fc.t[1];
    const real Z = fc.R[6] * q.P[0] + fc.R[7] * q.P[1] + fc.R[8] * q.P[2] + fc.t[2];
    const real ps = p.pyr_scale;
    const real fx = p.intr[0] * ps, fy = p.intr[1] * ps, cxs = p.intr[2] * ps, cys = p.intr[3] * ps;
    const real iz = 1.0 / Z;
    const real x0 = X * iz, y0 = Y * iz;
    const real r2 = x0 * x0 + y0 * y0, r4 = r2 * r2, r6 = r4 * r2;
    const real k0 = p.dist[0], k1 = p.dist[1], k2 = p.dist[2], k3 = p.dist[3], k4 = p.dist[4];
    const real dc = 1.0 + k0 * r2 + k1 * r4 + k2 * r6;
    const real xd = x0 * dc + 2.0 * k3 * x0 * y0 + k4 * (r2 + 2.0 * x0 * x0);
    const real yd = y0 * dc + 2.0 * k4 * xd * y0 + k3 * (r2 + 2.0 * y0 * y0);
    const real u = fx * xd + cxs, v = fy * yd + cys;
    if (u < 0.0 || u > (real)(fc.w - 1) || v < 0.0 || v > (real)(fc.h - 1)) return false;
    if (!(u == u) || !(v == v)) return false;      // NaN coordinates: Ceres' comparisons are all false -> in-bounds -> NaN lum -> invalid row
    real f, dfdr, dfdc;
    bicubic(fc.lum, fc.w, fc.h, v, u, f, dfdr, dfdc);
    o.lum = f;
    if (!WITH_J) return true;
    const real dcr = k0 + 2.0 * k1 * r2 + 3.0 * k2 * r4;                 // d dc / d r2
    const real dxd_dx0 = dc + 2.0 * x0 * x0 * dcr + 2.0 * k3 * y0 + 6.0 * k4 * x0;
    const real dxd_dy0 = 2.0 * x0 * y0 * dcr + 2.0 * k3 * x0 + 2.0 * k4 * y0;
    const real dyd_dx0 = 2.0 * x0 * y0 * dcr + 2.0 * k4 * y0 * dxd_dx0 + 2.0 * k3 * x0;
    const real dyd_dy0 = dc + 2.0 * y0 * y0 * dcr + 2.0 * k4 * (xd + y0 * dxd_dy0) + 6.0 * k3 * y0;
    const real au = dfdc * fx, av = dfdr * fy;
    const real lx = au * dxd_dx0 + av * dyd_dx0, ly = au * dxd_dy0 + av * dyd_dy0;
    const real Lq[3] = {lx * iz, ly * iz, -(lx * x0 + ly * y0) * iz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const real* d = fc.dR[i];
        const real dX = d[0] * q.P[0] + d[1] * q.P[1] + d[2] * q.P[2];
        const real dY = d[3] * q.P[0] + d[4] * q.P[1] + d[5] * q.P[2];
        const real dZ = d[6] * q.P[0] + d[7] * q.P[1] + d[8] * q.P[2];
        o.dpose[i] = Lq[0] * dX + Lq[1] * dY + Lq[2] * dZ;
        o.dpose[3 + i] = Lq[i];
        o.M[i] = Lq[0] * fc.R[i] + Lq[1] * fc.R[3 + i] + Lq[2] * fc.R[6 + i];
    }
    o.dintr[0] = dfdc * ps * xd; o.dintr[1] = dfdr * ps * yd; o.dintr[2] = dfdc * ps; o.dintr[3] = dfdr * ps;
    const real dxk[5] = {x0 * r2, x0 * r4, x0 * r6, 2.0 * x0 * y0, r2 + 2.0 * x0 * x0};
    const real c2 = 2.0 * k4 * y0;
    const real dyk[5] = {y0 * r2 + c2 * dxk[0], y0 * r4 + c2 * dxk[1], y0 * r6 + c2 * dxk[2],
                         c2 * dxk[3] + (r2 + 2.0 * y0 * y0), 2.0 * xd * y0 + c2 * dxk[4]};
#pragma unroll
    for (int i = 0; i < 5; ++i) o.ddist[i] = au * dxk[i] + av * dyk[i];
    return true;
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
    const float d = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float one_minus = 1.0f - d;
    // std::max(1-d, 0.01f) keeps a NaN first argument
    return (one_minus < 0.01f) ? 0.01f : one_minus;
}

// block-level sum of a double into one atomic
static __device__ inline void block_add(double v, double* dst) {
    __shared__ double sm[4];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i]; if (t != 0.0) atomicAdd(dst, t); }
    __syncthreads();
}

template <bool WITH_J>
__global__ void __launch_bounds__(256) k_build(GridView g, RowView r, OptParams p, const FrameConst* __restrict__ frames, double* cost_out) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    double cost = 0.0;
    if (a < r.A) {
        const int N = g.N; const size_t Acap = r.Acap;
        const int s = r.alist[a];
        const uint8_t fl = r.aflags[a];
        if (!(fl & F_ACTIVE)) {                 // free-only entry: unknowns but no rows
            if (WITH_J) {
                r.regflags[a] = 0; r.ea_free[a] = 0; r.nrows[a] = 0;
                for (int d = 0; d < 6; ++d) r.ea_w[(size_t)d * Acap + a] = 0.0f;
                for (int k = 0; k < r.slots; ++k) r.rows[row_index(a, k, 7, r.slots)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
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
            if (p.use_er && ring_ok) {
                rf |= 1;
                bool fr = (fl & F_FREE_SDF) != 0;
#pragma unroll
                for (int d = 0; d < 6; ++d) fr |= (g.flags[ring[d]] & F_FREE_SDF) != 0;
                if (fr) rf |= 8;
            }
            if (p.use_es) { rf |= 2; if ((xs - g.sdf0[s]) != 0.0) rf |= 4; if (fl & F_FREE_SDF) rf |= 16; }
            uint8_t eafree = 0;
            const uchar4 col = g.color[s];
            const int myrank = g.rank[s];
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                float w = 0.0f;
                if (p.use_ea && ring_ok) {
                    const int nb = ring[d];
                    const bool added_before = (g.flags[nb] & F_ACTIVE) && g.rank[nb] < myrank;     // voxels_added, optimizer.cpp:267-279
                    if (!added_before) {
                        w = chroma_weight(col, g.color[nb]);
                        if (!(w == w) || isinf(w)) w = 0.0f;
                        if (w != 0.0f && ((fl & F_FREE_ALB) || (g.flags[nb] & F_FREE_ALB))) eafree |= (uint8_t)(1 << d);
                    }
                }
                r.ea_w[(size_t)d * Acap + a] = w;
            }
            r.regflags[a] = rf; r.ea_free[a] = eafree;
        } else {
            rf = r.regflags[a];
            // cost of the regulariser rows at this state (rows without a free parameter are not part of the reduced program)
            if ((rf & 1) && (rf & 8)) {
                const double dxx = g.x_sdf[ring[0]] + g.x_sdf[ring[1]] - 2.0 * xs, dyy = g.x_sdf[ring[2]] + g.x_sdf[ring[3]] - 2.0 * xs,
                             dzz = g.x_sdf[ring[4]] + g.x_sdf[ring[5]] - 2.0 * xs;
                const double lap = dxx + dyy + dzz; cost += 0.5 * p.type_w[1] * lap * lap;
            }
            if ((rf & 2) && (rf & 16)) { double e = xs - g.sdf0[s]; if (e == 0.0) e = 0.0000001; cost += 0.5 * p.type_w[2] * e * e; }
            const uint8_t eafree = r.ea_free[a];
            const double xa = g.x_alb[s];
#pragma unroll
            for (int d = 0; d < 6; ++d) if (eafree & (1 << d)) {
                const double e = xa - g.x_alb[ring[d]];
                cost += 0.5 * (double)r.ea_w[(size_t)d * Acap + a] * p.type_w[3] * e * e;
            }
        }

        // ---- Eg rows ---------------------------------------------------------------------------------------
        const int nin = WITH_J ? r.slots : (int)r.nrows[a];       // candidates: observation slots (assembly) or stored rows (cost)
        bool any_row = false;
        if (WITH_J) { for (int k = 0; k < r.slots; ++k) any_row |= r.obs_w[(size_t)k * Acap + a] > 0.0f; any_row &= eligible; }
        else any_row = nin > 0;
        int nout = 0;
        if (any_row) {
            real sd[10];
#pragma unroll
            for (int c = 0; c < 10; ++c) sd[c] = g.x_sdf[idx[c]];
            real sh[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) sh[j] = (real)g.sh[(size_t)j * N + s];
            const int cx = g.cx[s], cy = g.cy[s], cz = g.cz[s];
            const real vs = (real)g.voxel_size;
            PointShared q[4];
            // sdf slots: 0:000 1:010 2:020 3:011 4:001 5:002 6:100 7:110 8:101 9:200 (shading_cost.h:88-97)
            shared_point(q[0], sd[0], sd[6], sd[1], sd[4], g.x_alb[idx[10]], sh, cx, cy, cz, vs);
            shared_point(q[1], sd[6], sd[9], sd[7], sd[8], g.x_alb[idx[11]], sh, cx + 1, cy, cz, vs);
            shared_point(q[2], sd[1], sd[7], sd[2], sd[3], g.x_alb[idx[12]], sh, cx, cy + 1, cz, vs);
            shared_point(q[3], sd[4], sd[8], sd[3], sd[5], g.x_alb[idx[13]], sh, cx, cy, cz + 1, vs);
            bool vox_free = false;
            if (WITH_J) {
#pragma unroll
                for (int c = 0; c < 10; ++c) vox_free |= (g.flags[idx[c]] & F_FREE_SDF) != 0;
#pragma unroll
                for (int c = 10; c < 14; ++c) vox_free |= (g.flags[idx[c]] & F_FREE_ALB) != 0;
                vox_free |= !p.fix_poses || !p.fix_intr || !p.fix_dist;
            }
            const double weight_sdf = sdf_to_weight(xs, (double)g.truncation);
            // slot of (s, sx, sy, sz) for each point
            constexpr int PS[4][4] = {{0, 6, 1, 4}, {6, 9, 7, 8}, {1, 7, 2, 3}, {4, 8, 3, 5}};

            for (int k = 0; k < nin; ++k) {
                const size_t ka = (size_t)k * Acap + a;
                float roww; int f;
                if (WITH_J) { const float ow = r.obs_w[ka]; f = r.obs_frame[ka]; roww = (ow > 0.0f) ? (float)((double)ow * weight_sdf) : 0.0f; }
                else { const float4 m = r.rows[row_index(a, k, 7, r.slots)]; const int fb = __float_as_int(m.z); roww = (fb & ROW_FREE_BIT) ? m.x : 0.0f; f = fb & ~ROW_FREE_BIT; }
                if (roww == 0.0f) continue;
                const FrameConst& fc = frames[f];
                PointFrame pf[4];
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) ok = ok && eval_point<WITH_J>(q[j], fc, p, pf[j]);
                real res = 0.0, c1 = 0, c2 = 0, c3 = 0;
                if (ok) {
                    const real B0 = q[0].alb * q[0].Ls, B1 = q[1].alb * q[1].Ls, B2 = q[2].alb * q[2].Ls, B3 = q[3].alb * q[3].Ls;
                    const real d1 = (B1 - B0) - (pf[1].lum - pf[0].lum), d2 = (B2 - B0) - (pf[2].lum - pf[0].lum), d3 = (B3 - B0) - (pf[3].lum - pf[0].lum);
                    res = sqrt(d1 * d1 + d2 * d2 + d3 * d3);
                    if (!(res > 0.0) || isinf(res)) { ok = false; res = 0.0; }    // 0, NaN, inf -> NV_INVALID_RESIDUAL (shading_cost.h:186-195)
                    else { const real ir = 1.0 / res; c1 = d1 * ir; c2 = d2 * ir; c3 = d3 * ir; }
                }
                if (!WITH_J) { if (ok) cost += 0.5 * (double)roww * p.type_w[0] * res * res; continue; }
                if (!ok) continue;                                                 // dropped at creation (shading_cost.cpp:136-145)
                const real cj[4] = {-(c1 + c2 + c3), c1, c2, c3};
                real J[P_TOTAL];
#pragma unroll
                for (int i = 0; i < P_TOTAL; ++i) J[i] = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // E_j = B_j - lum_j;  dE/dg = alb * N dLs + s * N M,  dE/ds (direct) = M . n
                    real v[3] = {q[j].alb * q[j].dLs[0] + q[j].s * pf[j].M[0], q[j].alb * q[j].dLs[1] + q[j].s * pf[j].M[1], q[j].alb * q[j].dLs[2] + q[j].s * pf[j].M[2]};
                    real G[3]; apply_normal_jac(q[j], v, G);
                    const real direct = pf[j].M[0] * q[j].n[0] + pf[j].M[1] * q[j].n[1] + pf[j].M[2] * q[j].n[2];
                    J[PS[j][1]] += cj[j] * G[0]; J[PS[j][2]] += cj[j] * G[1]; J[PS[j][3]] += cj[j] * G[2];
                    J[PS[j][0]] += cj[j] * (direct - (G[0] + G[1] + G[2]));
                    J[P_ALB + j] = cj[j] * q[j].Ls;
#pragma unroll
                    for (int i = 0; i < 6; ++i) J[P_POSE + i] -= cj[j] * pf[j].dpose[i];
#pragma unroll
                    for (int i = 0; i < 4; ++i) J[P_INTR + i] -= cj[j] * pf[j].dintr[i];
#pragma unroll
                    for (int i = 0; i < 5; ++i) J[P_DIST + i] -= cj[j] * pf[j].ddist[i];
                }
                bool fin = true;
#pragma unroll
                for (int i = 0; i < P_TOTAL; ++i) fin = fin && !(isnan(J[i]) || isinf(J[i]));
                if (!fin) continue;
                // rows of a voxel are compacted into its first slots (creation order = ascending observation weight)
#pragma unroll
                for (int gq = 0; gq < 7; ++gq)
                    r.rows[row_index(a, nout, gq, r.slots)] = make_float4((float)J[4 * gq], (float)J[4 * gq + 1], (float)J[4 * gq + 2], (float)J[4 * gq + 3]);
                r.rows[row_index(a, nout, 7, r.slots)] = make_float4(roww, (float)res, __int_as_float(f | (vox_free ? ROW_FREE_BIT : 0)), (float)J[28]);
                ++nout;
            }
        }
        if (WITH_J) {
            r.nrows[a] = (uint8_t)nout;
            for (int k = nout; k < r.slots; ++k) r.rows[row_index(a, k, 7, r.slots)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        }
    }
    if (!WITH_J) block_add(cost, cost_out);
}

void launch_build(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames, bool with_jacobian, double* cost_out) {
    if (r.A <= 0) return;
    const int blocks = (r.A + 255) / 256;
    if (with_jacobian) k_build<true><<<blocks, 256, 0, st>>>(g, r, p, frames, cost_out);
    else k_build<false><<<blocks, 256, 0, st>>>(g, r, p, frames, cost_out);
}

// nls_solver.cpp:379-394 — per-type sums of the row weights (sums[0..3]) and row counts (sums[4..7])
__global__ void __launch_bounds__(256) k_weight_sums(RowView r, double* sums) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, n0 = 0, n3 = 0, na = 0;
    if (a < r.A && (r.aflags[a] & F_ACTIVE)) {
        na = 1.0;
        const int nr = r.nrows[a];
        for (int k = 0; k < nr; ++k) { const float w = r.rows[row_index(a, k, 7, r.slots)].x; s0 += (double)w; if (w != 0.0f) n0 += 1.0; }
        const uint8_t rf = r.regflags[a];
        if (rf & 1) s1 = 1.0;
        if (rf & 2) s2 = 1.0;
        for (int d = 0; d < 6; ++d) { const float w = r.ea_w[(size_t)d * r.Acap + a]; s3 += (double)w; if (w != 0.0f) n3 += 1.0; }
    }
    block_add(s0, sums + 0); block_add(s1, sums + 1); block_add(s2, sums + 2); block_add(s3, sums + 3);
    block_add(n0, sums + 4); block_add(n3, sums + 7); block_add(na, sums + 8);
}
void launch_weight_sums(hipStream_t st, RowView r, double* sums9) {
    if (r.A > 0) k_weight_sums<<<(r.A + 255) / 256, 256, 0, st>>>(r, sums9);
}

}  // namespace i3d
