// K5/K6 — matrix-free normal-equation operator of the Gauss-Newton / LM step:  y = J^T W J x,  g = J^T W r,
// diag(J^T W J) and the dense pose / intrinsics / distortion blocks of the block-Jacobi preconditioner.
//
// Replaces Ceres' BlockSparseMatrix Jacobian + CgnrLinearOperator + BlockJacobiPreconditioner [Ceres 2.1.0, not in
// /root/reference; selected at nls_solver.cpp:307] for this problem's FIXED row structure:
//   * Eg rows are stored (29 fp32 partials, [29][slots][A] planes, one coalesced stream per column);
//   * Er / Es / Ea rows have constant coefficients (volumetric_regularizer.h:59-72, surface_stab_regularizer.h:59-66,
//     albedo_regularizer.h:59-66) and are never stored — their action is recomputed from per-voxel flags;
//   * column indices are implicit: a row's voxel columns are the centre voxel's neighbour-table entries.
//
// J^T is a GATHER, not a scatter: pass 1 (k_eg_pass, one lane per active voxel) reads the voxel's <= slots rows ONCE,
// forms t = W (J x) per row and immediately the 14 per-voxel column sums  C[c] = sum_k J[c][k] t_k  (all rows of a voxel
// share the same 14 voxel columns), which go to a staging plane; pass 2 (k_gather, one lane per voxel) pulls the 10+4
// staged sums of the voxels whose stencil contains it, plus the regulariser terms.  No fp32 atomics on voxel unknowns,
// deterministic.  Only the 6K+9 shared camera unknowns are reduced with LDS atomics -> one fp64 atomic per block/entry.
#include "kernels.hpp"

namespace i3d {

template <int MODE>
__global__ void __launch_bounds__(256) k_eg_pass(GridView g, RowView r, OptParams p, const float* __restrict__ u, PassBuffers b, int lds_floats) {
    extern __shared__ float lds[];        // [6K+9] (+ [21K+25] in COLNORM)
    for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) lds[i] = 0.0f;
    __syncthreads();
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = g.N, Acap = r.Acap, K = p.K;
    const int nshared = 6 * K + 9;
    float cam9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cam9[i] = 0.0f;
    if (a < r.A) {
        const int s = r.alist[a];
        int idx[P_VOX];
#pragma unroll
        for (int c = 0; c < 10; ++c) { const int nb = slot_fwd_nbr(c); idx[c] = nb < 0 ? s : g.nbr[(size_t)nb * N + s]; }
        idx[10] = idx[0]; idx[11] = idx[6]; idx[12] = idx[1]; idx[13] = idx[4];
        int ring[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) ring[d] = g.nbr[(size_t)d * N + s];

        float uv[P_VOX];
        float acc[P_VOX];
#pragma unroll
        for (int c = 0; c < P_VOX; ++c) acc[c] = 0.0f;
#pragma unroll
        for (int i = 0; i < 9; ++i) cam9[i] = 0.0f;
        bool loaded = false;
        for (int k = 0; k < r.slots; ++k) {
            const size_t ka = (size_t)k * Acap + a;
            const float w = r.roww[ka];
            if (w == 0.0f) continue;
            const float rho = w * (float)p.type_w[0];
            const int f = r.obs_frame[ka];
            float J[P_TOTAL];
#pragma unroll
            for (int i = 0; i < P_TOTAL; ++i) J[i] = r.J[((size_t)i * r.slots + k) * Acap + a];
            if (MODE == PASS_COLNORM) {
#pragma unroll
                for (int c = 0; c < P_VOX; ++c) acc[c] += rho * J[c] * J[c];
                float* bl = lds + nshared;
                int o = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    atomicAdd(&lds[6 * f + i], rho * J[P_POSE + i] * J[P_POSE + i]);
#pragma unroll
                    for (int j = i; j < 6; ++j) { atomicAdd(&bl[21 * f + o], rho * J[P_POSE + i] * J[P_POSE + j]); ++o; }
                }
                o = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    atomicAdd(&lds[6 * K + i], rho * J[P_INTR + i] * J[P_INTR + i]);
#pragma unroll
                    for (int j = i; j < 4; ++j) { atomicAdd(&bl[21 * K + o], rho * J[P_INTR + i] * J[P_INTR + j]); ++o; }
                }
                o = 0;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    atomicAdd(&lds[6 * K + 4 + i], rho * J[P_DIST + i] * J[P_DIST + i]);
#pragma unroll
                    for (int j = i; j < 5; ++j) { atomicAdd(&bl[21 * K + 10 + o], rho * J[P_DIST + i] * J[P_DIST + j]); ++o; }
                }
            } else {
                float t;
                if (MODE == PASS_GRAD) t = rho * r.res[ka];
                else {
                    if (!loaded) {
#pragma unroll
                        for (int c = 0; c < 10; ++c) uv[c] = idx[c] >= 0 ? u[idx[c]] : 0.0f;
#pragma unroll
                        for (int c = 10; c < P_VOX; ++c) uv[c] = idx[c] >= 0 ? u[N + idx[c]] : 0.0f;
                        loaded = true;
                    }
                    float d = 0.0f;
#pragma unroll
                    for (int c = 0; c < P_VOX; ++c) d += J[c] * uv[c];
                    const float* up = u + 2 * (size_t)N + 6 * f;
#pragma unroll
                    for (int i = 0; i < 6; ++i) d += J[P_POSE + i] * up[i];
                    const float* ui = u + 2 * (size_t)N + 6 * K;
#pragma unroll
                    for (int i = 0; i < 9; ++i) d += J[P_INTR + i] * ui[i];
                    t = rho * d;
                }
#pragma unroll
                for (int c = 0; c < P_VOX; ++c) acc[c] += J[c] * t;
#pragma unroll
                for (int i = 0; i < 6; ++i) atomicAdd(&lds[6 * f + i], J[P_POSE + i] * t);
#pragma unroll
                for (int i = 0; i < 9; ++i) cam9[i] += J[P_INTR + i] * t;     // intrinsics + distortion: registers, reduced per wave below
            }
        }
#pragma unroll
        for (int c = 0; c < P_VOX; ++c) b.C[(size_t)c * Acap + a] = acc[c];

        // ---- regulariser rows: tr (Er), ts (Es, Jacobian folded in), ta[6] (Ea) ---------------------------------
        const uint8_t rf = r.regflags[a];
        float tr = 0.0f, ts = 0.0f;
        if (rf & 1) {
            const float rho = (float)p.type_w[1];
            if (MODE == PASS_COLNORM) tr = rho;
            else if (MODE == PASS_GRAD) {
                const double xs = g.x_sdf[s];
                const double dxx = g.x_sdf[ring[0]] + g.x_sdf[ring[1]] - 2.0 * xs, dyy = g.x_sdf[ring[2]] + g.x_sdf[ring[3]] - 2.0 * xs,
                             dzz = g.x_sdf[ring[4]] + g.x_sdf[ring[5]] - 2.0 * xs;
                tr = rho * (float)(dxx + dyy + dzz);
            } else {
                float d = -6.0f * u[s];
#pragma unroll
                for (int q = 0; q < 6; ++q) d += u[ring[q]];
                tr = rho * d;
            }
        }
        if ((rf & 2) && (rf & 4)) {        // Es row with unit Jacobian (residual != 0, surface_stab_regularizer.h:62-64)
            const float rho = (float)p.type_w[2];
            if (MODE == PASS_COLNORM) ts = rho;
            else if (MODE == PASS_GRAD) ts = rho * (float)(g.x_sdf[s] - g.sdf0[s]);
            else ts = rho * u[s];
        }
        b.treg[a] = tr; b.treg[(size_t)Acap + a] = ts;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const float w = r.ea_w[(size_t)d * Acap + a];
            float ta = 0.0f;
            if (w != 0.0f) {
                const float rho = w * (float)p.type_w[3];
                if (MODE == PASS_COLNORM) ta = rho;
                else if (MODE == PASS_GRAD) ta = rho * (float)(g.x_alb[s] - g.x_alb[ring[d]]);
                else ta = rho * (u[N + s] - u[N + ring[d]]);
            }
            b.treg[(size_t)(2 + d) * Acap + a] = ta;
        }
    }
    if (MODE != PASS_COLNORM) {
        // the 9 intrinsics/distortion columns are shared by every row: wave-shuffle reduction, one LDS atomic per wave
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            float v = cam9[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((threadIdx.x & 63) == 0 && v != 0.0f) atomicAdd(&lds[6 * K + i], v);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nshared; i += blockDim.x) { const float v = lds[i]; if (v != 0.0f) atomicAdd(&b.shared[i], (double)v); }
    if (MODE == PASS_COLNORM)
        for (int i = threadIdx.x; i < 21 * K + 25; i += blockDim.x) { const float v = lds[nshared + i]; if (v != 0.0f) atomicAdd(&b.blocks[i], (double)v); }
}

void launch_eg_pass(hipStream_t st, PassMode mode, GridView g, RowView r, OptParams p, const float* u, PassBuffers b) {
    if (r.A <= 0) return;
    const int blocks = (r.A + 255) / 256;
    const int nshared = 6 * p.K + 9;
    if (mode == PASS_GRAD) k_eg_pass<PASS_GRAD><<<blocks, 256, nshared * sizeof(float), st>>>(g, r, p, u, b, nshared);
    else if (mode == PASS_JTJP) k_eg_pass<PASS_JTJP><<<blocks, 256, nshared * sizeof(float), st>>>(g, r, p, u, b, nshared);
    else { const int n = nshared + 21 * p.K + 25; k_eg_pass<PASS_COLNORM><<<blocks, 256, n * sizeof(float), st>>>(g, r, p, u, b, n); }
}

// pass 2: one lane per stored voxel pulls what the rows contribute to its two unknowns
template <bool SQUARED>
__global__ void __launch_bounds__(256) k_gather(GridView g, RowView r, PassBuffers b, float* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = g.N;
    if (s >= N) return;
    const uint8_t fl = g.flags[s];
    const size_t Acap = r.Acap;
    float osdf = 0.0f, oalb = 0.0f;
    if (fl & (F_FREE_SDF | F_FREE_ALB)) {
        const int as = g.aidx[s];
        int ringa[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) { const int nb = g.nbr[(size_t)d * N + s]; ringa[d] = nb >= 0 ? g.aidx[nb] : -1; }
        if (fl & F_FREE_SDF) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                const int rn = slot_rev_nbr(c);
                int av;
                if (rn < 0) av = as; else if (rn < 6) av = ringa[rn]; else { const int nb = g.nbr[(size_t)rn * N + s]; av = nb >= 0 ? g.aidx[nb] : -1; }
                if (av >= 0) acc += b.C[(size_t)c * Acap + av];
            }
            if (as >= 0) acc += b.treg[Acap + as] + (SQUARED ? 36.0f : -6.0f) * b.treg[as];
#pragma unroll
            for (int d = 0; d < 6; ++d) if (ringa[d] >= 0) acc += b.treg[ringa[d]];
            osdf = acc;
        }
        if (fl & F_FREE_ALB) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 10; c < P_VOX; ++c) {
                const int rn = slot_rev_nbr(c);
                const int av = rn < 0 ? as : ringa[rn];
                if (av >= 0) acc += b.C[(size_t)c * Acap + av];
            }
            if (as >= 0) {
#pragma unroll
                for (int d = 0; d < 6; ++d) acc += b.treg[(size_t)(2 + d) * Acap + as];
            }
#pragma unroll
            for (int d = 0; d < 6; ++d) if (ringa[d] >= 0) {
                const float t = b.treg[(size_t)(2 + (d ^ 1)) * Acap + ringa[d]];     // neighbour's edge pointing back at this voxel
                acc += SQUARED ? t : -t;
            }
            oalb = acc;
        }
    }
    out[s] = osdf; out[(size_t)N + s] = oalb;
}
void launch_gather(hipStream_t st, PassMode mode, GridView g, RowView r, PassBuffers b, float* out) {
    if (g.N <= 0) return;
    const int blocks = (g.N + 255) / 256;
    if (mode == PASS_COLNORM) k_gather<true><<<blocks, 256, 0, st>>>(g, r, b, out);
    else k_gather<false><<<blocks, 256, 0, st>>>(g, r, b, out);
}

__global__ void k_shared_finalize(int K, OptParams p, const double* __restrict__ shared, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * K + 9) return;
    const bool fixed = i < 6 * K ? p.fix_poses : (i < 6 * K + 4 ? p.fix_intr : p.fix_dist);
    out[i] = fixed ? 0.0f : (float)shared[i];
}
void launch_shared_finalize(hipStream_t st, int K, OptParams p, const double* shared, float* out) {
    k_shared_finalize<<<(6 * K + 9 + 255) / 256, 256, 0, st>>>(K, p, shared, out);
}

// ---- vector helpers ----------------------------------------------------------------------------------------
#define GRID_STRIDE(n) const int stride = gridDim.x * blockDim.x; for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += stride)
static inline int vblocks(int n) { int b = (n + 255) / 256; return b < 1 ? 1 : (b > 4096 ? 4096 : b); }

__global__ void k_fill(int n, float* x, float v) { GRID_STRIDE(n) x[i] = v; }
__global__ void k_fill_d(int n, double* x, double v) { GRID_STRIDE(n) x[i] = v; }
__global__ void k_mul(int n, const float* a, const float* b, float* o) { GRID_STRIDE(n) o[i] = a[i] * b[i]; }
__global__ void k_scale(int n, const float* c, const float* m, float* S) { GRID_STRIDE(n) S[i] = m[i] != 0.0f ? 1.0f / (1.0f + sqrtf(c[i])) : 0.0f; }
__global__ void k_lm_diag(int n, const float* c, const float* S, float inv_radius, float* D2, float* Minv) {
    GRID_STRIDE(n) {
        const float s = S[i];
        if (s == 0.0f) { D2[i] = 0.0f; Minv[i] = 0.0f; continue; }
        const float cs = c[i] * s * s;
        const float d2 = fminf(fmaxf(cs, 1e-6f), 1e32f) * inv_radius;
        D2[i] = d2; Minv[i] = 1.0f / (cs + d2);
    }
}
__global__ void k_op_tail(int n, const float* S, const float* acc, const float* D2, const float* p, float* q) { GRID_STRIDE(n) q[i] = S[i] * acc[i] + D2[i] * p[i]; }
__global__ void k_axpy(int n, float a, const float* x, float* y) { GRID_STRIDE(n) y[i] += a * x[i]; }
__global__ void k_xpay(int n, const float* x, float a, float* y) { GRID_STRIDE(n) y[i] = x[i] + a * y[i]; }
__global__ void k_sub(int n, const float* a, const float* b, float* o) { GRID_STRIDE(n) o[i] = a[i] - b[i]; }

static __device__ inline void block_add_d(double v, double* dst) {
    __shared__ double sm[4];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i]; atomicAdd(dst, t); }
    __syncthreads();
}
__global__ void __launch_bounds__(256) k_dot(int n, const float* a, const float* b, double* out) {
    double s = 0.0; GRID_STRIDE(n) s += (double)a[i] * (double)b[i];
    block_add_d(s, out);
}
__global__ void __launch_bounds__(256) k_dot3(int n, const float* x, const float* b, const float* r, const float* D2, double* out) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    GRID_STRIDE(n) { const double xi = x[i]; s0 += xi * ((double)b[i] + (double)r[i]); s1 += xi * (double)r[i]; s2 += (double)D2[i] * xi * xi; }
    block_add_d(s0, out); block_add_d(s1, out + 1); block_add_d(s2, out + 2);
}

void launch_fill(hipStream_t st, int n, float* x, float v) { if (n > 0) k_fill<<<vblocks(n), 256, 0, st>>>(n, x, v); }
void launch_fill_d(hipStream_t st, int n, double* x, double v) { if (n > 0) k_fill_d<<<vblocks(n), 256, 0, st>>>(n, x, v); }
void launch_mul(hipStream_t st, int n, const float* a, const float* b, float* o) { if (n > 0) k_mul<<<vblocks(n), 256, 0, st>>>(n, a, b, o); }
void launch_scale_from_colnorm(hipStream_t st, int n, const float* c, const float* m, float* S) { if (n > 0) k_scale<<<vblocks(n), 256, 0, st>>>(n, c, m, S); }
void launch_lm_diag(hipStream_t st, int n, const float* c, const float* S, float ir, float* D2, float* Minv) { if (n > 0) k_lm_diag<<<vblocks(n), 256, 0, st>>>(n, c, S, ir, D2, Minv); }
void launch_apply_op_tail(hipStream_t st, int n, const float* S, const float* acc, const float* D2, const float* p, float* q) { if (n > 0) k_op_tail<<<vblocks(n), 256, 0, st>>>(n, S, acc, D2, p, q); }
void launch_axpy(hipStream_t st, int n, float a, const float* x, float* y) { if (n > 0) k_axpy<<<vblocks(n), 256, 0, st>>>(n, a, x, y); }
void launch_xpay(hipStream_t st, int n, const float* x, float a, float* y) { if (n > 0) k_xpay<<<vblocks(n), 256, 0, st>>>(n, x, a, y); }
void launch_sub(hipStream_t st, int n, const float* a, const float* b, float* o) { if (n > 0) k_sub<<<vblocks(n), 256, 0, st>>>(n, a, b, o); }
void launch_dot(hipStream_t st, int n, const float* a, const float* b, double* out) { if (n > 0) k_dot<<<vblocks(n) > 1024 ? 1024 : vblocks(n), 256, 0, st>>>(n, a, b, out); }
void launch_dot3(hipStream_t st, int n, const float* x, const float* b, const float* r, const float* D2, double* out3) { if (n > 0) k_dot3<<<vblocks(n) > 1024 ? 1024 : vblocks(n), 256, 0, st>>>(n, x, b, r, D2, out3); }

// z_shared = Minv_block * r_shared for the K 6x6 pose blocks, the 4x4 intrinsics block and the 5x5 distortion block
__global__ void k_precond_shared(int K, const float* __restrict__ Minv, const float* __restrict__ rs, float* __restrict__ zs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * K + 9) return;
    int base, n, row; const float* M;
    if (i < 6 * K) { const int f = i / 6; base = 6 * f; n = 6; row = i - base; M = Minv + 36 * f; }
    else if (i < 6 * K + 4) { base = 6 * K; n = 4; row = i - base; M = Minv + 36 * K; }
    else { base = 6 * K + 4; n = 5; row = i - base; M = Minv + 36 * K + 16; }
    float s = 0.0f;
    for (int j = 0; j < n; ++j) s += M[row * n + j] * rs[base + j];
    zs[i] = s;
}
void launch_precond_shared(hipStream_t st, int K, const float* Minv, const float* rs, float* zs) {
    k_precond_shared<<<(6 * K + 9 + 255) / 256, 256, 0, st>>>(K, Minv, rs, zs);
}

__global__ void k_freemask(GridView g, OptParams p, float* __restrict__ mask) {
    const int N = g.N, NP = 2 * N + 6 * p.K + 9;
    GRID_STRIDE(NP) {
        float m;
        if (i < N) m = (g.flags[i] & F_FREE_SDF) ? 1.0f : 0.0f;
        else if (i < 2 * N) m = (g.flags[i - N] & F_FREE_ALB) ? 1.0f : 0.0f;
        else if (i < 2 * N + 6 * p.K) m = p.fix_poses ? 0.0f : 1.0f;
        else if (i < 2 * N + 6 * p.K + 4) m = p.fix_intr ? 0.0f : 1.0f;
        else m = p.fix_dist ? 0.0f : 1.0f;
        mask[i] = m;
    }
}
void launch_freemask(hipStream_t st, GridView g, OptParams p, float* mask) { k_freemask<<<vblocks(2 * g.N + 6 * p.K + 9), 256, 0, st>>>(g, p, mask); }

// candidate point x + S*step (TrustRegionMinimizer: delta = step .* jacobian_scaling), squared norms of delta and x over the free parameters
__global__ void __launch_bounds__(256) k_candidate(GridView g, int K, float sign, const float* __restrict__ step, const float* __restrict__ S,
                                                   const double* __restrict__ xsh, double* xc_sdf, double* xc_alb, double* xc_sh,
                                                   double* norms2, const float* __restrict__ mask) {
    const int N = g.N, NP = 2 * N + 6 * K + 9;
    double d2 = 0.0, x2 = 0.0;
    GRID_STRIDE(NP) {
        const double delta = (double)sign * (double)step[i] * (double)S[i];
        double x;
        if (i < N) { x = g.x_sdf[i]; xc_sdf[i] = x + delta; }
        else if (i < 2 * N) { x = g.x_alb[i - N]; xc_alb[i - N] = x + delta; }
        else { x = xsh[i - 2 * N]; xc_sh[i - 2 * N] = x + delta; }
        if (mask[i] != 0.0f) { d2 += delta * delta; x2 += x * x; }
    }
    block_add_d(d2, norms2); block_add_d(x2, norms2 + 1);
}
void launch_candidate(hipStream_t st, GridView g, int K, float sign, const float* step, const float* S, const double* xsh, double* xc_sdf, double* xc_alb,
                      double* xc_sh, double* norms2, const float* mask) {
    int b = vblocks(2 * g.N + 6 * K + 9); if (b > 1024) b = 1024;
    k_candidate<<<b, 256, 0, st>>>(g, K, sign, step, S, xsh, xc_sdf, xc_alb, xc_sh, norms2, mask);
}
__global__ void k_accept(GridView g, const double* __restrict__ xc_sdf, const double* __restrict__ xc_alb) {
    GRID_STRIDE(g.N) { const double a = xc_sdf[i], b = xc_alb[i]; g.x_sdf[i] = a; g.x_alb[i] = b; g.f_sdf[i] = (float)a; g.f_alb[i] = (float)b; }
}
void launch_accept(hipStream_t st, GridView g, const double* xc_sdf, const double* xc_alb) { if (g.N > 0) k_accept<<<vblocks(g.N), 256, 0, st>>>(g, xc_sdf, xc_alb); }

}  // namespace i3d
