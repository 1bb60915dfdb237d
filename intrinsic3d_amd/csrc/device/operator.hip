// K5/K6 — matrix-free normal-equation operator of the Gauss-Newton / LM step and the fused PCG iteration.
//
// Replaces Ceres' BlockSparseMatrix Jacobian + CgnrLinearOperator + BlockJacobiPreconditioner + ConjugateGradientsSolver
// [Ceres 2.1.0, not in /root/reference; selected at nls_solver.cpp:307] for this problem's FIXED row structure:
//   * Eg rows are stored wave-tiled (common.hpp RowView / row_index: one 7680 B block per [group of 64 entries][slot] = seven float4 planes + one
//     float2 plane, 120 B per row, the row weight folded into the partials), so a wave reads one contiguous block per slot; Er / Es / Ea rows have constant coefficients (volumetric_regularizer.h:59-72,
//     surface_stab_regularizer.h:59-66, albedo_regularizer.h:59-66) and are never stored;
//   * column indices are implicit: a row's voxel columns are the centre voxel's neighbour-table entries;
//   * all solver vectors live in WORK-LIST space (entries = voxels that own rows or unknowns): [sdf A | albedo A | poses 6K |
//     intrinsics 4 | distortion 5]; a neighbour outside the list is a fixed parameter and contributes 0.
//
// J^T is a GATHER, not a scatter: pass 1 (k_eg_jtjp for the PCG operator, k_eg_pass<GRAD|COLNORM> for the gradient and the column
// norms; one lane per entry) reads the entry's rows ONCE, forms t = W (J u) per
// row and immediately the 14 per-voxel column sums C[c] = sum_k J[c][k] t_k (all rows of a voxel share the same 14 voxel
// columns) into a staging plane; pass 2 (k_gather) pulls the 10+4 staged sums of the entries whose stencil contains it plus
// the regulariser terms, applies the Jacobi scaling / LM diagonal and accumulates p.q.  No fp32 atomics on voxel unknowns.
// The 6K+9 camera columns: poses are summed across the wave per distinct keyframe (slots are keyframe-ordered) and one lane issues the
// LDS atomics; intrinsics / distortion accumulate in registers; each workgroup then adds its totals with one fp64 atomic per entry.
// Every other fp64 reduction goes through per-workgroup partials (reduce_device.hpp) — no same-address atomics from thousands of workgroups.
//
// The PCG scalars (rho, p.q, alpha, beta, Q) never leave the device inside a solve: the two camera-tail kernels turn the fp64 partial
// sums into alpha / beta / termination flags, every kernel of the iteration starts with `if (state->done) return`, and the host only
// polls the state one pass behind (no pipeline bubble).
#include "kernels.hpp"
#include "reduce_device.hpp"
#include "wave_ops.hpp"
#include <cstdlib>

namespace i3d {

#define GRID_STRIDE(n) const int stride = gridDim.x * blockDim.x; for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += stride)
static inline int vblocks(int n) { int b = (n + 255) / 256; return b < 1 ? 1 : (b > 2048 ? 2048 : b); }

// dst[k] += (assign: =) sum over workgroups of partials[b * ncomp + k]   (one workgroup).  A thread's partials are requested eight at a time (clamped index) and added in the order
// of the plain loop: one load and one wait per partial made this kernel nine round trips long on the cost kernel's 9 k partials.
__global__ void __launch_bounds__(1024) k_reduce_partials(const double* __restrict__ partials, int nblk, int ncomp, double* dst, const PcgState* __restrict__ state, int assign) {
    if (state && state->done) return;
    for (int k = 0; k < ncomp; ++k) {
        double s = 0.0;
        for (int i0 = threadIdx.x; i0 < nblk; i0 += 8 * 1024) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = partials[(size_t)min(i0 + q * 1024, nblk - 1) * ncomp + k];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (i0 + q * 1024 < nblk) s += v[q];
        }
        const double t = block_sum_d(s);
        if (threadIdx.x == 0) dst[k] = assign ? t : dst[k] + t;
    }
}
void launch_reduce_partials(hipStream_t st, const double* partials, int nblk, int ncomp, double* dst, const PcgState* state, bool assign) {
    if (nblk > 0) k_reduce_partials<<<1, 1024, 0, st>>>(partials, nblk, ncomp, dst, state, assign ? 1 : 0);
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------------
// Persistent 1024-thread workgroups (one per CU, 16 waves): each walks a contiguous chunk of 1024-entry tiles.  The pose
// columns are accumulated in LDS: aggregated per wave and keyframe (wave_accumulate), with `reps` replicas of the [6K] accumulator
// (odd stride: different banks) for the rare slots that hold more than 3 keyframes.  LDS is zeroed / flushed once per workgroup.
constexpr int EG_THREADS = 1024;

template <int MODE>
__global__ void __launch_bounds__(EG_THREADS) k_eg_pass(GridView g, RowView r, OptParams p, const float* __restrict__ u, PassBuffers b,
                                                        int reps, int tiles_per_block, const PcgState* __restrict__ state) {
    if (state && state->done) return;
    // Fixed-order sums (round 4): the pose columns of a wave's rows go into a table private to the wave (wave_ops.hpp: wave_table_add_quads since round 5 — the wave sums of a round taken four to a tree), merged into the workgroup's
    // dense accumulator in wave order at the end of every tile; the intrinsics / distortion sums leave through per-wave slots; the workgroup's totals leave as ONE
    // float row (b.part), summed over the workgroups in a fixed order by k_sum_rows — no LDS accumulator shared by waves, no global atomics: the gradient and the
    // column norms of an outer iteration are bit-reproducible.
    extern __shared__ float lds[];        // [ticket | 3] [NW][NCAM] | [NW][TC] tags | [NW][TC][NPV] | dense [rs] | [NCAM] | JTJP: staged camera part of u [6K+9]
    const int K = p.K; const size_t Acap = r.Acap;
    const int nshared = 6 * K + 9;
    constexpr int NPVT = (MODE == PASS_COLNORM) ? 21 : 6, TCT = (MODE == PASS_COLNORM) ? 16 : 32, NWT = EG_THREADS / 64;
    const int rs = ((MODE == PASS_COLNORM) ? 21 * K : 6 * K) | 1;
    constexpr int NCAM = (MODE == PASS_COLNORM) ? 34 : 9;       // COLNORM: 9 squared columns + 10 + 15 block entries of intrinsics / distortion
    constexpr int D_CAMW = 4, D_TAG = D_CAMW + ((NWT * NCAM + 3) & ~3), D_VAL = D_TAG + NWT * TCT, D0 = D_VAL + NWT * TCT * NPVT;
    const int nacc = D0 + rs + NCAM;
    for (int i = threadIdx.x; i < nacc; i += EG_THREADS) lds[i] = (i >= D_TAG && i < D_VAL) ? __int_as_float(-1) : 0.0f;
    float* upose = lds + nacc;            // JTJP: the 6K+9 camera entries of u (every row reads 6+9 of them)
    const size_t tail = 2 * (size_t)r.chunk;          // camera part of every solver vector
    const int chunk = r.chunk;
    if (MODE == PASS_JTJP) for (int i = threadIdx.x; i < nshared; i += EG_THREADS) upose[i] = u[tail + i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int o_tag = D_TAG + wave * TCT, o_val = D_VAL + wave * (TCT * NPVT);
    int tcount = 0;
    float* const cam_acc = lds + D0 + rs;
    (void)reps;
    float cam9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cam9[i] = 0.0f;
    float camb[NCAM];
#pragma unroll
    for (int i = 0; i < NCAM; ++i) camb[i] = 0.0f;
    const float tw0 = (float)p.type_w[0];
    const int nC = r.nC;
    const int ntiles = (nC + EG_THREADS - 1) / EG_THREADS;
    const int tile0 = blockIdx.x * tiles_per_block;

    for (int tile = tile0; tile < tile0 + tiles_per_block && tile < ntiles; ++tile) {
        const int ci = tile * EG_THREADS + threadIdx.x;
        const bool in = ci < nC;
        const int a = in ? (r.clist ? r.clist[ci] : ci) : 0;         // compute list of this rank (identity when not sharded)
        const bool owned = in && a >= r.own0 && a < r.own1;           // camera columns are accumulated once: by the row's owner
        const uint8_t fl = in ? r.aflags[a] : 0;
        const int nr = (in && (fl & F_ACTIVE)) ? (int)r.nrows[a] : 0;
        // wave-uniform row count so that the row loads are unconditional and batched
        int nr_max = nr;
        for (int o = 32; o > 0; o >>= 1) nr_max = max(nr_max, __shfl_xor(nr_max, o, 64));

        float acc[P_VOX];
#pragma unroll
        for (int c = 0; c < P_VOX; ++c) acc[c] = 0.0f;
        float uv[P_VOX];
#pragma unroll
        for (int c = 0; c < P_VOX; ++c) uv[c] = 0.0f;
        if (MODE == PASS_JTJP && nr > 0) {
#pragma unroll
            for (int c = 0; c < P_VOX; ++c) {
                const int nb = slot_fwd_nbr(c);
                const int la = nb < 0 ? a : r.anbr[(size_t)nb * Acap + a];
                uv[c] = la >= 0 ? u[c < 10 ? vec_sdf(la, chunk) : vec_alb(la, chunk)] : 0.0f;
            }
        }
        const size_t ac = in ? (size_t)a : 0;
        for (int k = 0; k < nr_max; ++k) {
            const float4* __restrict__ row = r.rows + row_index(ac, k, 0, r.slots);     // 7 planes, 64 float4 apart: one contiguous 7 KB block per wave
            float4 j4[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) j4[q] = ld_row(row + q * 64);
            const size_t ro = row_scalar_index(ac, k, r.slots);
            const float2 jt = r.row_jt()[row_jt_index(ac, k, r.slots)];
            constexpr int NPV = (MODE == PASS_COLNORM) ? 21 : 6;
            float pv[NPV]; int fsel = 0; bool pvalid = false;
#pragma unroll
            for (int i = 0; i < NPV; ++i) pv[i] = 0.0f;
            if (k < nr) {
                const float rho = tw0;                       // the row weight is folded into the stored partials (Js = sqrt(w) J)
                const int f = __float_as_int(jt.y) & ~ROW_FREE_BIT;
                float J[P_TOTAL];
#pragma unroll
                for (int q = 0; q < 7; ++q) { J[4 * q] = j4[q].x; J[4 * q + 1] = j4[q].y; J[4 * q + 2] = j4[q].z; J[4 * q + 3] = j4[q].w; }
                J[28] = jt.x;
                if (MODE == PASS_COLNORM) {
#pragma unroll
                    for (int c = 0; c < P_VOX; ++c) acc[c] += rho * J[c] * J[c];
                  if (owned) {
                    int o = 0;                                  // upper triangle of this keyframe's 6x6 block
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
#pragma unroll
                        for (int j = i; j < 6; ++j) { pv[o] = rho * J[P_POSE + i] * J[P_POSE + j]; ++o; }
                    }
                    fsel = f; pvalid = true;
                    // intrinsics / distortion blocks are shared by every row: registers, reduced once per workgroup
#pragma unroll
                    for (int i = 0; i < 9; ++i) camb[i] += rho * J[P_INTR + i] * J[P_INTR + i];
                    o = 9;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = i; j < 4; ++j) { camb[o] += rho * J[P_INTR + i] * J[P_INTR + j]; ++o; }
                    }
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
#pragma unroll
                        for (int j = i; j < 5; ++j) { camb[o] += rho * J[P_DIST + i] * J[P_DIST + j]; ++o; }
                    }
                  }
                } else {
                    float t;
                    if (MODE == PASS_GRAD) { const float2 wr = r.row_wr[ro]; t = rho * (sqrtf(wr.x) * wr.y); }      // Js^T (tw sqrt(w) r) = J^T W r
                    else {
                        float d = 0.0f;
#pragma unroll
                        for (int c = 0; c < P_VOX; ++c) d += J[c] * uv[c];
                        const float* up = upose + 6 * f;
#pragma unroll
                        for (int i = 0; i < 6; ++i) d += J[P_POSE + i] * up[i];
                        const float* ui = upose + 6 * K;
#pragma unroll
                        for (int i = 0; i < 9; ++i) d += J[P_INTR + i] * ui[i];
                        t = rho * d;
                    }
#pragma unroll
                    for (int c = 0; c < P_VOX; ++c) acc[c] += J[c] * t;
                    if (!p.fix_poses && owned) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) pv[i] = J[P_POSE + i] * t;
                        fsel = f; pvalid = true;
                    }
                    if (owned) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) cam9[i] += J[P_INTR + i] * t;
                    }     // intrinsics + distortion: registers across all tiles, reduced once below
                }
            }
            // pose columns of this slot: every lane of the wave takes part (the loop bound nr_max is wave-uniform)
            wave_table_add_quads<NPV, TCT, (MODE == PASS_COLNORM ? 2 : 3)>(pvalid, fsel, [&](int q) { return pv[q]; }, lds, o_tag, o_val, tcount, D0, NPV);
        }
        // the waves' tables -> the dense accumulator, in wave order (once per tile of 1024 entries: this kernel runs once per outer iteration)
        __syncthreads();
        for (int w = 0; w < NWT; ++w) { if (wave == w) wave_table_merge<NPVT, TCT>(lds, o_tag, o_val, tcount, D0, NPVT); __syncthreads(); }
        if (in) {
#pragma unroll
            for (int c = 0; c < P_VOX; ++c) b.C[(size_t)c * Acap + a] = acc[c];
            // ---- regulariser rows: tr (Er), ts (Es, Jacobian folded in), ta[6] (Ea) ---------------------------------
            const uint8_t rf = (fl & F_ACTIVE) ? r.regflags[a] : 0;
            float tr = 0.0f, ts = 0.0f;
            const int s = r.alist[a];
            const int N = g.N;
            if (rf & 1) {
                const float rho = (float)p.type_w[1];
                if (MODE == PASS_COLNORM) tr = rho;
                else if (MODE == PASS_GRAD) {
                    const double xs = g.x_sdf[s];
                    double nbv[6];
#pragma unroll
                    for (int d = 0; d < 6; ++d) nbv[d] = g.x_sdf[g.nbr[(size_t)d * N + s]];
                    const double dxx = nbv[0] + nbv[1] - 2.0 * xs, dyy = nbv[2] + nbv[3] - 2.0 * xs, dzz = nbv[4] + nbv[5] - 2.0 * xs;
                    tr = rho * (float)(dxx + dyy + dzz);
                } else {
                    float d = -6.0f * u[vec_sdf(a, chunk)];
#pragma unroll
                    for (int q = 0; q < 6; ++q) { const int la = r.anbr[(size_t)q * Acap + a]; if (la >= 0) d += u[vec_sdf(la, chunk)]; }
                    tr = rho * d;
                }
            }
            if ((rf & 2) && (rf & 4)) {        // Es row with unit Jacobian (residual != 0, surface_stab_regularizer.h:62-64)
                const float rho = (float)p.type_w[2];
                if (MODE == PASS_COLNORM) ts = rho;
                else if (MODE == PASS_GRAD) ts = rho * (float)(g.x_sdf[s] - g.sdf0[s]);
                else ts = rho * u[vec_sdf(a, chunk)];
            }
            b.treg[a] = tr; b.treg[Acap + a] = ts;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const float w = (fl & F_ACTIVE) ? r.ea_w[(size_t)d * Acap + a] : 0.0f;
                float ta = 0.0f;
                if (w != 0.0f) {
                    const float rho = w * (float)p.type_w[3];
                    if (MODE == PASS_COLNORM) ta = rho;
                    else if (MODE == PASS_GRAD) ta = rho * (float)(g.x_alb[s] - g.x_alb[g.nbr[(size_t)d * N + s]]);
                    else { const int la = r.anbr[(size_t)d * Acap + a]; ta = rho * (u[vec_alb(a, chunk)] - (la >= 0 ? u[vec_alb(la, chunk)] : 0.0f)); }
                }
                b.treg[(size_t)(2 + d) * Acap + a] = ta;
            }
        }
    }
    // columns shared by every row: wave-shuffle reduction into the wave's slot, the slots added in wave order
#pragma unroll
    for (int i = 0; i < NCAM; ++i) {
        float v = (MODE == PASS_COLNORM) ? camb[i] : cam9[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) lds[D_CAMW + wave * NCAM + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NCAM) { float v = 0.0f; for (int w = 0; w < NWT; ++w) v += lds[D_CAMW + w * NCAM + threadIdx.x]; cam_acc[threadIdx.x] = v; }
    __syncthreads();
    // this workgroup's totals as one float row: [pose part (6K | 21K) | NCAM]; k_sum_rows adds the rows of all workgroups in a fixed order
    const int npose = (MODE == PASS_COLNORM) ? 21 * K : 6 * K;
    float* const row = b.part + (size_t)blockIdx.x * b.part_stride;
    for (int i = threadIdx.x; i < npose + NCAM; i += EG_THREADS) row[i] = i < npose ? lds[D0 + i] : cam_acc[i - npose];
}

// shared / blocks <- the workgroups' rows of a k_eg_pass launch (or of k_eg_tile's cam_part), added in workgroup order (fp64).  mode: PASS_COLNORM distributes the
// upper triangles into `blocks` and their diagonals into `shared`; otherwise the row is the camera block itself.
// 256 threads = 32 columns x 8 row groups: a thread adds every 8th row of its column (fp64), the 8 group sums are added in group order — a fixed association, and
// 8 independent chains per column instead of one serial walk over up to 512 rows.
__global__ void __launch_bounds__(256) k_sum_rows(int mode, int K, const float* __restrict__ part, int nrows, int stride, double* __restrict__ shared, double* __restrict__ blocks) {
    __shared__ double grp[8][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    const int ncol = mode == PASS_COLNORM ? 21 * K + 34 : 6 * K + 9;
    double v = 0.0;
    if (i < ncol) for (int w = rg; w < nrows; w += 8) v += (double)part[(size_t)w * stride + i];
    grp[rg][col] = v;
    __syncthreads();
    if (rg != 0 || i >= ncol) return;
    v = grp[0][col];
    for (int q = 1; q < 8; ++q) v += grp[q][col];
    if (mode != PASS_COLNORM) { shared[i] = v; return; }
    if (i < 21 * K) {
        blocks[i] = v;
        const int f = i / 21, o = i - 21 * f;            // diagonal entries of the block are the squared column norms
        const int di = o == 0 ? 0 : o == 6 ? 1 : o == 11 ? 2 : o == 15 ? 3 : o == 18 ? 4 : o == 20 ? 5 : -1;
        if (di >= 0) shared[6 * f + di] = v;
    } else { const int c = i - 21 * K; if (c < 9) shared[6 * K + c] = v; else blocks[21 * K + (c - 9)] = v; }
}
void launch_sum_rows(hipStream_t st, PassMode mode, int K, const float* part, int nrows, int stride, double* shared, double* blocks) {
    const int ncol = mode == PASS_COLNORM ? 21 * K + 34 : 6 * K + 9;
    k_sum_rows<<<(ncol + 31) / 32, 256, 0, st>>>((int)mode, K, part, nrows, stride, shared, blocks);
}

// ---- the PCG operator pass: k_eg_pass<PASS_JTJP> specialised for memory-level parallelism --------------------------------------
// Same arithmetic as k_eg_pass<PASS_JTJP>.  Differences: the 14 operator-input values of the lane's voxel are parked in LDS instead of
// registers, which makes room for TWO row blocks in flight per lane: the loads of slot k+1 are issued before slot k is consumed.
// (The pass is bound by memory round trips: 16 / 12 / 8 waves per CU run 0.42 / 0.46 / 0.56 ms; a second row in flight per wave acts
// like twice the waves without the registers for them.)
__global__ void __launch_bounds__(EG_THREADS) k_eg_jtjp(GridView g, RowView r, OptParams p, const float* __restrict__ u, PassBuffers b,
                                                        int reps, int tiles_per_block, const PcgState* __restrict__ state) {
    if (state && state->done) return;
    extern __shared__ float lds[];        // [reps][rs] pose accumulators | [9] | camera part of u [6K+9] | uv [14][EG_THREADS]
    const int K = p.K; const size_t Acap = r.Acap;
    const int nshared = 6 * K + 9;
    const int rs = (6 * K) | 1;
    const int nacc = reps * rs + 9;
    for (int i = threadIdx.x; i < nacc; i += EG_THREADS) lds[i] = 0.0f;
    float* const upose = lds + nacc;
    float* const uvl = upose + nshared + threadIdx.x;                    // this lane's column of the [14][EG_THREADS] staging
    const size_t tail = 2 * (size_t)r.chunk;
    const int chunk = r.chunk;
    for (int i = threadIdx.x; i < nshared; i += EG_THREADS) upose[i] = u[tail + i];
    __syncthreads();
    float* const cam_acc = lds + reps * rs;
    float* const pose_acc = lds + (threadIdx.x & (reps - 1)) * rs;
    float* const wave_acc = lds + ((threadIdx.x >> 6) & (reps - 1)) * rs;
    float cam9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cam9[i] = 0.0f;
    const float tw0 = (float)p.type_w[0];
    const int nC = r.nC;
    const int ntiles = (nC + EG_THREADS - 1) / EG_THREADS;
    const int tile0 = blockIdx.x * tiles_per_block;
    const float* const ui = upose + 6 * K;

    for (int tile = tile0; tile < tile0 + tiles_per_block && tile < ntiles; ++tile) {
        const int ci = tile * EG_THREADS + threadIdx.x;
        const bool in = ci < nC;
        const int a = in ? (r.clist ? r.clist[ci] : ci) : 0;
        const bool owned = in && a >= r.own0 && a < r.own1;
        const uint8_t fl = in ? r.aflags[a] : 0;
        const int nr = (in && (fl & F_ACTIVE)) ? (int)r.nrows[a] : 0;
        int nr_max = nr;
        for (int o = 32; o > 0; o >>= 1) nr_max = max(nr_max, __shfl_xor(nr_max, o, 64));
        const size_t ac = in ? (size_t)a : 0;
        // first row block on its way before anything else of the tile is touched
        float4 rwA[8], rwB[8];                          // planes 0..6 + (column 28, keyframe tag) in [7].xy
        auto load_block = [&](float4 (&rw)[8], int k) {
            const float4* __restrict__ row = r.rows + row_index(ac, k, 0, r.slots);
#pragma unroll
            for (int q = 0; q < 7; ++q) rw[q] = ld_row(row + q * 64);
            { const float2 jt = r.row_jt()[row_jt_index(ac, k, r.slots)]; rw[7].x = jt.x; rw[7].y = jt.y; }
        };
        if (nr_max > 0) load_block(rwA, 0);
        // operator input at the 14 stencil unknowns of the voxel -> LDS (0 where the neighbour is not in the list = fixed parameter)
        if (in) {
#pragma unroll
            for (int c = 0; c < P_VOX; ++c) {
                const int nb = slot_fwd_nbr(c);
                const int la = nb < 0 ? a : r.anbr[(size_t)nb * Acap + a];
                uvl[c * EG_THREADS] = la >= 0 ? u[c < 10 ? vec_sdf(la, chunk) : vec_alb(la, chunk)] : 0.0f;
            }
        }
        if (nr_max > 1) load_block(rwB, 1);
        // ---- regulariser rows: tr (Er), ts (Es, Jacobian folded in), ta[6] (Ea).  They do not depend on the Eg rows: computed HERE, while
        //      the first two row blocks are in flight, so their index / vector gathers cost no round trip of their own ----
        if (in) {
            const uint8_t rf = (fl & F_ACTIVE) ? r.regflags[a] : 0;
            // the +x,+y,+z ring values are already staged (sdf slots 6,1,4 / albedo slots 11,12,13; 0 when outside the list); only -x,-y,-z are fetched
            const float us = uvl[0], ua = uvl[10 * EG_THREADS];
            float rs_[6], ra_[6];
            rs_[0] = uvl[6 * EG_THREADS]; rs_[2] = uvl[1 * EG_THREADS]; rs_[4] = uvl[4 * EG_THREADS];
            ra_[0] = uvl[11 * EG_THREADS]; ra_[2] = uvl[12 * EG_THREADS]; ra_[4] = uvl[13 * EG_THREADS];
#pragma unroll
            for (int q = 1; q < 6; q += 2) {
                const int la = (rf || (fl & F_ACTIVE)) ? r.anbr[(size_t)q * Acap + a] : -1;
                rs_[q] = la >= 0 ? u[vec_sdf(la, chunk)] : 0.0f; ra_[q] = la >= 0 ? u[vec_alb(la, chunk)] : 0.0f;
            }
            float tr = 0.0f, ts = 0.0f;
            if (rf & 1) tr = (float)p.type_w[1] * (((((((-6.0f * us) + rs_[0]) + rs_[1]) + rs_[2]) + rs_[3]) + rs_[4]) + rs_[5]);
            if ((rf & 2) && (rf & 4)) ts = (float)p.type_w[2] * us;
            b.treg[a] = tr; b.treg[Acap + a] = ts;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const float w = (fl & F_ACTIVE) ? r.ea_w[(size_t)d * Acap + a] : 0.0f;
                b.treg[(size_t)(2 + d) * Acap + a] = (w != 0.0f) ? w * (float)p.type_w[3] * (ua - ra_[d]) : 0.0f;
            }
        }
        float acc[P_VOX];
#pragma unroll
        for (int c = 0; c < P_VOX; ++c) acc[c] = 0.0f;

        auto consume = [&](const float4 (&rw)[8], int k) {
            float pv[6]; int fsel = 0; bool pvalid = false;
#pragma unroll
            for (int i = 0; i < 6; ++i) pv[i] = 0.0f;
            if (k < nr) {
                const float rho = tw0;                       // the row weight is folded into the stored partials
                const int f = __float_as_int(rw[7].y) & ~ROW_FREE_BIT;
                float J[P_TOTAL];
#pragma unroll
                for (int q = 0; q < 7; ++q) { J[4 * q] = rw[q].x; J[4 * q + 1] = rw[q].y; J[4 * q + 2] = rw[q].z; J[4 * q + 3] = rw[q].w; }
                J[28] = rw[7].x;
                float d = 0.0f;
#pragma unroll
                for (int c = 0; c < P_VOX; ++c) d += J[c] * uvl[c * EG_THREADS];
                const float* up = upose + 6 * f;
#pragma unroll
                for (int i = 0; i < 6; ++i) d += J[P_POSE + i] * up[i];
#pragma unroll
                for (int i = 0; i < 9; ++i) d += J[P_INTR + i] * ui[i];
                const float t = rho * d;
#pragma unroll
                for (int c = 0; c < P_VOX; ++c) acc[c] += J[c] * t;
                if (!p.fix_poses && owned) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) pv[i] = J[P_POSE + i] * t;
                    fsel = f; pvalid = true;
                }
                if (owned) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) cam9[i] += J[P_INTR + i] * t;
                }
            }
            wave_accumulate<6>(pvalid, fsel, pv, pose_acc, wave_acc, 6);
        };
        for (int k = 0; k < nr_max; k += 2) {                 // slot k is in rwA, slot k+1 (if any) in rwB; a buffer is refilled as soon as it is consumed
            consume(rwA, k);
            if (k + 2 < nr_max) load_block(rwA, k + 2);
            if (k + 1 < nr_max) {
                consume(rwB, k + 1);
                if (k + 3 < nr_max) load_block(rwB, k + 3);
            }
        }
        if (in) {
#pragma unroll
            for (int c = 0; c < P_VOX; ++c) b.C[(size_t)c * Acap + a] = acc[c];
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        float v = cam9[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v != 0.0f) atomicAdd(&cam_acc[i], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nshared; i += EG_THREADS) {
        float v;
        if (i < 6 * K) { v = 0.0f; for (int q = 0; q < reps; ++q) v += lds[q * rs + i]; }
        else v = cam_acc[i - 6 * K];
        if (v != 0.0f) atomicAdd(&b.shared[i], (double)v);
    }
}


// returns the number of workgroups = rows written to b.part (GRAD / COLNORM; 0 for the untiled operator, which adds its camera block into b.shared)
int launch_eg_pass(hipStream_t st, PassMode mode, GridView g, RowView r, OptParams p, const float* u, PassBuffers b, const PcgState* state) {
    if (r.nC <= 0) return 0;
    static int num_cu = 0;
    if (!num_cu) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev); num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    const int ntiles = (r.nC + EG_THREADS - 1) / EG_THREADS;
    const int blocks = ntiles < num_cu ? ntiles : num_cu;                 // one persistent workgroup per CU
    const int tiles_per_block = (ntiles + blocks - 1) / blocks;
    const int nshared = 6 * p.K + 9;
    const int rs = (6 * p.K) | 1;
    constexpr int NW = EG_THREADS / 64;
    auto det_words = [](int ncam, int tc, int npv) { return 4 + ((NW * ncam + 3) & ~3) + NW * tc + NW * tc * npv; };
    if (mode == PASS_GRAD) {
        const size_t lds_g = (size_t)(det_words(9, 32, 6) + rs + 9) * sizeof(float);
        if (!set_dynamic_lds((const void*)k_eg_pass<PASS_GRAD>, "k_eg_pass<GRAD>", lds_g, p.K)) return 0;
        k_eg_pass<PASS_GRAD><<<blocks, EG_THREADS, lds_g, st>>>(g, r, p, u, b, 1, tiles_per_block, state);
        return blocks;
    } else if (mode == PASS_JTJP) {
        int rj = 8;                                        // replicas only serve the rare > 3-keyframe fallback; LDS goes to the uv staging
        while (rj > 1 && (size_t)(rj * rs + 9 + nshared + P_VOX * EG_THREADS) * sizeof(float) > 150 * 1024) rj >>= 1;
        const size_t lds_j = (size_t)(rj * rs + 9 + nshared + P_VOX * EG_THREADS) * sizeof(float);
        if (!set_dynamic_lds((const void*)k_eg_jtjp, "k_eg_jtjp", lds_j, p.K)) return 0;
        k_eg_jtjp<<<blocks, EG_THREADS, lds_j, st>>>(g, r, p, u, b, rj, tiles_per_block, state);
        return 0;
    }
    const int rs2 = (21 * p.K) | 1;
    const size_t n = (size_t)(det_words(34, 16, 21) + rs2 + 34) * sizeof(float);
    if (!set_dynamic_lds((const void*)k_eg_pass<PASS_COLNORM>, "k_eg_pass<COLNORM>", n, p.K)) return 0;
    k_eg_pass<PASS_COLNORM><<<blocks, EG_THREADS, n, st>>>(g, r, p, u, b, 1, tiles_per_block, state);
    return blocks;
}

// ---- pass 2: one lane per work-list entry pulls what the rows contribute to its two unknowns ------------------------------
// TAIL: out = S*acc + D2*v (the CGNR operator applied to v) and, if dot_out, dot_out += v.out
template <bool SQUARED, bool TAIL>
__global__ void __launch_bounds__(256) k_gather(RowView r, PassBuffers b, float* __restrict__ out, const float* __restrict__ S,
                                                const float* __restrict__ D2, const float* __restrict__ v, double* dot_out, int dot_atomic,
                                                const PcgState* __restrict__ state) {
    if (state && state->done) return;
    const int a = r.own0 + blockIdx.x * blockDim.x + threadIdx.x;       // owned range of this rank
    const int A = r.A; const size_t Acap = r.Acap;
    double dotp = 0.0;
    if (a < r.own1 && a < A) {
        const uint8_t fl = r.aflags[a];
        float osdf = 0.0f, oalb = 0.0f;
        if (fl & (F_FREE_SDF | F_FREE_ALB)) {
            int ringa[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) ringa[d] = r.anbr[(size_t)d * Acap + a];
            // every gather below is UNCONDITIONAL (a missing neighbour reads this entry's own slot and contributes nothing): a load behind `if (av >= 0)` is a load the
            // compiler cannot count — it drained everything in flight at each of them, one round trip per stencil slot (tools/isa_drains.py: 20 of the kernel's 23 waits)
            if (fl & F_FREE_SDF) {
                float acc = 0.0f;
                int avs[10];
#pragma unroll
                for (int c = 0; c < 10; ++c) { const int rn = slot_rev_nbr(c); avs[c] = rn < 0 ? a : (rn < 6 ? ringa[rn] : r.anbr[(size_t)rn * Acap + a]); }
                float cv[10];
#pragma unroll
                for (int c = 0; c < 10; ++c) cv[c] = b.C[(size_t)c * Acap + (avs[c] >= 0 ? avs[c] : a)];
                float tv[6];
#pragma unroll
                for (int d = 0; d < 6; ++d) tv[d] = b.treg[ringa[d] >= 0 ? ringa[d] : a];
                const float t1 = b.treg[Acap + a], t0 = b.treg[a];
#pragma unroll
                for (int c = 0; c < 10; ++c) if (avs[c] >= 0) acc += cv[c];
                acc += t1 + (SQUARED ? 36.0f : -6.0f) * t0;
#pragma unroll
                for (int d = 0; d < 6; ++d) if (ringa[d] >= 0) acc += tv[d];
                osdf = acc;
            }
            if (fl & F_FREE_ALB) {
                float acc = 0.0f;
                float cv[4], te[6], tn[6];
#pragma unroll
                for (int c = 10; c < P_VOX; ++c) { const int rn = slot_rev_nbr(c); const int av = rn < 0 ? a : ringa[rn]; cv[c - 10] = b.C[(size_t)c * Acap + (av >= 0 ? av : a)]; }
#pragma unroll
                for (int d = 0; d < 6; ++d) { te[d] = b.treg[(size_t)(2 + d) * Acap + a]; tn[d] = b.treg[(size_t)(2 + (d ^ 1)) * Acap + (ringa[d] >= 0 ? ringa[d] : a)]; }     // tn: neighbour's edge pointing back at this entry
#pragma unroll
                for (int c = 10; c < P_VOX; ++c) { const int rn = slot_rev_nbr(c); const int av = rn < 0 ? a : ringa[rn]; if (av >= 0) acc += cv[c - 10]; }
#pragma unroll
                for (int d = 0; d < 6; ++d) acc += te[d];
#pragma unroll
                for (int d = 0; d < 6; ++d) if (ringa[d] >= 0) acc += SQUARED ? tn[d] : -tn[d];
                oalb = acc;
            }
        }
        const int js = vec_sdf(a, r.chunk), ja = js + r.chunk;
        if (TAIL) {
            const float v0 = v[js], v1 = v[ja];
            osdf = S[js] * osdf + D2[js] * v0; oalb = S[ja] * oalb + D2[ja] * v1;
            dotp = (double)v0 * (double)osdf + (double)v1 * (double)oalb;
        }
        out[js] = osdf; out[ja] = oalb;
    }
    if (TAIL && dot_out) {
        if (!dot_atomic) block_partial_d(dotp, dot_out, 1, 0);      // per-workgroup partial of v.out (summed by k_pcg_tail_b)
        else { const double t = block_sum_d(dotp); if (threadIdx.x == 0 && t != 0.0) atomicAdd(dot_out, t); }     // sharded: ~1/world of the workgroups, no extra dispatch
    }
}
void launch_gather(hipStream_t st, PassMode mode, RowView r, PassBuffers b, float* out) {
    const int n = (r.own1 < r.A ? r.own1 : r.A) - r.own0;
    if (n <= 0) return;
    const int blocks = (n + 255) / 256;
    if (mode == PASS_COLNORM) k_gather<true, false><<<blocks, 256, 0, st>>>(r, b, out, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    else k_gather<false, false><<<blocks, 256, 0, st>>>(r, b, out, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
}
int launch_gather_tail(hipStream_t st, RowView r, PassBuffers b, float* out, const float* S, const float* D2, const float* v, double* dot_partials, bool dot_atomic,
                       const PcgState* state) {
    const int n = (r.own1 < r.A ? r.own1 : r.A) - r.own0;
    if (n <= 0) return 0;
    const int blocks = (n + 255) / 256;
    k_gather<false, true><<<blocks, 256, 0, st>>>(r, b, out, S, D2, v, dot_partials, dot_atomic ? 1 : 0, state);
    return blocks;                       // number of partials written (when dot_partials != nullptr)
}

// camera tail of the vectors: out[2A + i] from the fp64 accumulators; TAIL as above
__global__ void k_shared_finalize(size_t tail_off, int K, OptParams p, const double* __restrict__ shared, float* __restrict__ out, int tail,
                                  const float* __restrict__ S, const float* __restrict__ D2, const float* __restrict__ v, double* dot_out,
                                  const PcgState* __restrict__ state) {
    if (state && state->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double dotp = 0.0;
    if (i < 6 * K + 9) {
        const bool fixed = i < 6 * K ? p.fix_poses : (i < 6 * K + 4 ? p.fix_intr : p.fix_dist);
        float o = fixed ? 0.0f : (float)shared[i];
        const size_t j = tail_off + i;
        if (tail) { const float vv = v[j]; o = S[j] * o + D2[j] * vv; dotp = (double)vv * (double)o; }
        out[j] = o;
    }
    if (tail && dot_out) { const double t = block_sum_d(dotp); if (threadIdx.x == 0 && t != 0.0) atomicAdd(dot_out, t); }     // a handful of workgroups
}
void launch_shared_finalize(hipStream_t st, size_t tail_off, int K, OptParams p, const double* shared, float* out, bool tail, const float* S, const float* D2,
                            const float* v, double* dot_out, const PcgState* state) {
    k_shared_finalize<<<(6 * K + 9 + 255) / 256, 256, 0, st>>>(tail_off, K, p, shared, out, tail ? 1 : 0, S, D2, v, dot_out, state);
}

// ---- vector helpers -------------------------------------------------------------------------------------------------
__global__ void k_fill(int n, float* x, float v) { GRID_STRIDE(n) x[i] = v; }
__global__ void k_fill_d(int n, double* x, double v) { GRID_STRIDE(n) x[i] = v; }
__global__ void k_mul(int n, const float* a, const float* b, float* o) { GRID_STRIDE(n) o[i] = a[i] * b[i]; }
__global__ void k_mul2(int n, size_t seg, const float* a, const float* b, float* o) { GRID_STRIDE(2 * n) { const size_t j = i < n ? (size_t)i : (size_t)(i - n) + seg; o[j] = a[j] * b[j]; } }
__global__ void k_scale(int n, const float* c, const float* m, float* S, float* cm) { GRID_STRIDE(n) { const float v = m[i] != 0.0f ? c[i] : -1.0f; cm[i] = v; S[i] = lm_scale(v); } }
__global__ void k_lm_diag(int n, const float* c, const float* S, float inv_radius, float* D2, float* Minv) {
    GRID_STRIDE(n) { float d2, mi; lm_diag(c[i], S[i], inv_radius, d2, mi); D2[i] = d2; Minv[i] = mi; }
}
__global__ void __launch_bounds__(256) k_dot(int n, const float* a, const float* b, double* out) {
    double s = 0.0; GRID_STRIDE(n) s += (double)a[i] * (double)b[i];
    block_partial_d(s, out, 1, 0);
}
__global__ void __launch_bounds__(256) k_dot2(int n, size_t seg, const float* a, const float* b, double* out) {
    double s = 0.0; GRID_STRIDE(2 * n) { const size_t j = i < n ? (size_t)i : (size_t)(i - n) + seg; s += (double)a[j] * (double)b[j]; }
    block_partial_d(s, out, 1, 0);
}
// number of entries with |a_i m_i| > tol (as a double: it travels through the same sum reductions / all-reduces as the dot products).  Ceres' gradient test is a
// max-norm test, max_i |g_i| <= gradient_tolerance over the free parameters: true exactly when this count is 0 — and a count, unlike a maximum, is sum-reducible.
__global__ void __launch_bounds__(256) k_count_above(int n, const float* a, const float* m, float tol, double* out) {
    double s = 0.0; GRID_STRIDE(n) s += (fabsf(a[i] * m[i]) > tol) ? 1.0 : 0.0;
    block_partial_d(s, out, 1, 0);
}
__global__ void __launch_bounds__(256) k_count_above2(int n, size_t seg, const float* a, const float* m, float tol, double* out) {
    double s = 0.0; GRID_STRIDE(2 * n) { const size_t j = i < n ? (size_t)i : (size_t)(i - n) + seg; s += (fabsf(a[j] * m[j]) > tol) ? 1.0 : 0.0; }
    block_partial_d(s, out, 1, 0);
}
void launch_fill(hipStream_t st, int n, float* x, float v) { if (n > 0) k_fill<<<vblocks(n), 256, 0, st>>>(n, x, v); }
void launch_fill_d(hipStream_t st, int n, double* x, double v) { if (n > 0) k_fill_d<<<vblocks(n), 256, 0, st>>>(n, x, v); }
__global__ void k_int_to_double(const int* src, double* dst) { *dst = (double)*src; }
void launch_int_to_double(hipStream_t st, const int* src, double* dst) { k_int_to_double<<<1, 1, 0, st>>>(src, dst); }
void launch_mul(hipStream_t st, int n, const float* a, const float* b, float* o) { if (n > 0) k_mul<<<vblocks(n), 256, 0, st>>>(n, a, b, o); }
void launch_mul2(hipStream_t st, Seg2 sg, const float* a, const float* b, float* o) {
    if (sg.n > 0) k_mul2<<<vblocks(2 * sg.n), 256, 0, st>>>(sg.n, sg.off1 - sg.off0, a + sg.off0, b + sg.off0, o + sg.off0);
}
void launch_scale_from_colnorm(hipStream_t st, int n, const float* c, const float* m, float* S, float* cm) { if (n > 0) k_scale<<<vblocks(n), 256, 0, st>>>(n, c, m, S, cm); }
void launch_lm_diag(hipStream_t st, int n, const float* c, const float* S, float ir, float* D2, float* Minv) { if (n > 0) k_lm_diag<<<vblocks(n), 256, 0, st>>>(n, c, S, ir, D2, Minv); }
void launch_dot2(hipStream_t st, Seg2 sg, const float* a, const float* b, double* out, double* scratch) {      // out += a.b over both segments
    if (sg.n <= 0) return;
    const int blocks = vblocks(2 * sg.n) > 1024 ? 1024 : vblocks(2 * sg.n);
    k_dot2<<<blocks, 256, 0, st>>>(sg.n, sg.off1 - sg.off0, a + sg.off0, b + sg.off0, scratch);
    launch_reduce_partials(st, scratch, blocks, 1, out, nullptr);
}
void launch_count_above2(hipStream_t st, Seg2 sg, const float* a, const float* m, float tol, double* out, double* scratch) {      // out += #{|a m| > tol} over both segments
    if (sg.n <= 0) return;
    const int blocks = vblocks(2 * sg.n) > 1024 ? 1024 : vblocks(2 * sg.n);
    k_count_above2<<<blocks, 256, 0, st>>>(sg.n, sg.off1 - sg.off0, a + sg.off0, m + sg.off0, tol, scratch);
    launch_reduce_partials(st, scratch, blocks, 1, out, nullptr);
}
void launch_count_above(hipStream_t st, int n, const float* a, const float* m, float tol, double* out, double* scratch) {
    if (n <= 0) return;
    const int blocks = vblocks(n) > 1024 ? 1024 : vblocks(n);
    k_count_above<<<blocks, 256, 0, st>>>(n, a, m, tol, scratch);
    launch_reduce_partials(st, scratch, blocks, 1, out, nullptr);
}
void launch_dot(hipStream_t st, int n, const float* a, const float* b, double* out, double* scratch) {      // out += a.b
    if (n <= 0) return;
    const int blocks = vblocks(n) > 1024 ? 1024 : vblocks(n);
    k_dot<<<blocks, 256, 0, st>>>(n, a, b, scratch);
    launch_reduce_partials(st, scratch, blocks, 1, out, nullptr);
}

__global__ void k_freemask(RowView r, OptParams p, float* __restrict__ mask) {
    const int A = r.A, chunk = r.chunk, nv = 2 * chunk, NP = nv + 6 * p.K + 9;
    GRID_STRIDE(NP) {
        float m;
        if (i < nv) {                                   // [sdf chunk | alb chunk]; padding entries are fixed
            const int a = i < chunk ? i : i - chunk;
            m = (a < A && (r.aflags[a] & (i < chunk ? F_FREE_SDF : F_FREE_ALB))) ? 1.0f : 0.0f;
        }
        else if (i < nv + 6 * p.K) m = p.fix_poses ? 0.0f : 1.0f;
        else if (i < nv + 6 * p.K + 4) m = p.fix_intr ? 0.0f : 1.0f;
        else m = p.fix_dist ? 0.0f : 1.0f;
        mask[i] = m;
    }
}
void launch_freemask(hipStream_t st, RowView r, OptParams p, float* mask) { k_freemask<<<vblocks(2 * r.chunk + 6 * p.K + 9), 256, 0, st>>>(r, p, mask); }

// ---- fused PCG iteration (conjugate_gradients_solver.cc) -------------------------------------------------------------
// One iteration = 6 launches:  tail_a | direction | eg_tile | halo_fold | tail_b | step   (+ the rim push of the operator input when sharded).
//   k_pcg_step   (a rank's slice of the voxel unknowns, 16 B per lane):  x += alpha p ; q = S acc + D^2 p ; r -= alpha q ; z = M^-1 r ;
//                per-workgroup partial sums of r.z, x.(b+r), x.r, sum D^2 x^2
//   k_pcg_tail_a (camera tail, replicated, one workgroup): adds up the partials (sharded over the mailbox transport: and sums them over the
//                ranks, p2p_allreduce_wg), the same update on the 6K+9 camera unknowns, block-Jacobi z, then the scalar logic of the iteration
//                boundary: quadratic-model stop test (eta = 0.1) of the iteration just finished, rho / beta of the next one
//   k_pcg_direction  p = z + beta p ; u = S p on the rank's slice + the camera tail; partial sums of D^2 p^2
//   k_pcg_tail_b (one workgroup) p.q from the row and D^2 p^2 partials (sharded: [camera block | p.q] summed over the ranks), camera tail of
//                q = A p from the fp64 block, alpha
enum { STEP_INIT = 0, STEP_NORMAL = 1, STEP_XONLY = 2, STEP_RESET = 3 };

// QINLINE: `q` holds the raw operator accumulators J^T W J u of the tiled pass (k_eg_tile + k_halo_fold); the vector q = S acc + D^2 v
// (v = p, or x for the residual reset) is formed here instead of being written and read back.
template <int MODE, bool QINLINE>
__global__ void __launch_bounds__(256) k_pcg_step(int n4, int seg4 /* float4 distance between the sdf and the albedo segment */, const float4* __restrict__ p, const float4* __restrict__ q, float4* __restrict__ x, float4* __restrict__ r,
                                                  const float4* __restrict__ b, const float4* __restrict__ D2, const float4* __restrict__ Minv, float4* __restrict__ z,
                                                  const float4* __restrict__ S, double* __restrict__ partials /* [gridDim.x][4] */, PcgState* state) {
    if (state->done) return;
    const float alpha = (float)state->alpha;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < 2 * n4; j += gridDim.x * blockDim.x) {
        const int i = j < n4 ? j : j - n4 + seg4;          // a rank's slice = one segment of the sdf part + the same segment of the albedo part
        float xv[4], rv[4];
        if (MODE == STEP_INIT) { const float4 t = r[i]; rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w; }
        else {
            const float4 xo = x[i];
            xv[0] = xo.x; xv[1] = xo.y; xv[2] = xo.z; xv[3] = xo.w;
            if (MODE != STEP_RESET) {
                const float4 pp = p[i];
                xv[0] += alpha * pp.x; xv[1] += alpha * pp.y; xv[2] += alpha * pp.z; xv[3] += alpha * pp.w;
                x[i] = make_float4(xv[0], xv[1], xv[2], xv[3]);
            }
            if (MODE == STEP_XONLY) continue;
            float4 qq = q[i];                   // q = A p, or (RESET) tmp = A x
            const float4 bb = b[i], dd = D2[i];
            if (QINLINE) {
                const float4 sv = S[i];
                if (MODE == STEP_NORMAL) { const float4 pp = p[i]; qq = make_float4(sv.x * qq.x + dd.x * pp.x, sv.y * qq.y + dd.y * pp.y, sv.z * qq.z + dd.z * pp.z, sv.w * qq.w + dd.w * pp.w); }
                else qq = make_float4(sv.x * qq.x + dd.x * xv[0], sv.y * qq.y + dd.y * xv[1], sv.z * qq.z + dd.z * xv[2], sv.w * qq.w + dd.w * xv[3]);
            }
            if (MODE == STEP_NORMAL) { const float4 ro = r[i]; rv[0] = ro.x - alpha * qq.x; rv[1] = ro.y - alpha * qq.y; rv[2] = ro.z - alpha * qq.z; rv[3] = ro.w - alpha * qq.w; }
            if (MODE == STEP_RESET) { rv[0] = bb.x - qq.x; rv[1] = bb.y - qq.y; rv[2] = bb.z - qq.z; rv[3] = bb.w - qq.w; }
            r[i] = make_float4(rv[0], rv[1], rv[2], rv[3]);
            const float bv[4] = {bb.x, bb.y, bb.z, bb.w}, dv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double xd = xv[k]; s1 += xd * ((double)bv[k] + (double)rv[k]); s2 += xd * (double)rv[k]; s3 += (double)dv[k] * xd * xd; }
        }
        const float4 mm = Minv[i];
        const float zv[4] = {mm.x * rv[0], mm.y * rv[1], mm.z * rv[2], mm.w * rv[3]};
        z[i] = make_float4(zv[0], zv[1], zv[2], zv[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) s0 += (double)rv[k] * (double)zv[k];
    }
    if (MODE == STEP_XONLY) return;
    block_partial_d(s0, partials, 4, 0); block_partial_d(s1, partials, 4, 1); block_partial_d(s2, partials, 4, 2); block_partial_d(s3, partials, 4, 3);
}

// camera tail, x only (residual-reset iterations: x is needed before r = b - A x can be formed)
__global__ void k_pcg_tail_x(size_t to, int NS, const float* __restrict__ p, float* __restrict__ x, const PcgState* __restrict__ state) {
    if (state->done) return;
    const float alpha = (float)state->alpha;
    for (int i = threadIdx.x; i < NS; i += blockDim.x) x[to + i] += alpha * p[to + i];
}

__global__ void __launch_bounds__(1024) k_pcg_tail_a(int mode, size_t to, int K, const float* __restrict__ Mblk, const float* __restrict__ p, const float* __restrict__ q,
                                                    float* __restrict__ x, float* __restrict__ r, const float* __restrict__ b, const float* __restrict__ D2,
                                                    float* __restrict__ z, const double* __restrict__ partials, int nblk, PcgState* st,
                                                    double* __restrict__ shared_zero, int nzero, int* host_flags, int seq, P2PDev pd) {
    // (seq, done) goes to a 2-slot ring in pinned host memory: the host polls it instead of queueing a copy + event per pass
    auto publish = [&]() { if (host_flags) { __hip_atomic_store(&host_flags[2 * (seq & 1) + 1], st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                             __hip_atomic_store(&host_flags[2 * (seq & 1)], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } };
    if (st->done) { if (threadIdx.x == 0) publish(); return; }        // (the same decision on every rank: the scalars are replicated bit for bit)
    for (int i = threadIdx.x; i < nzero; i += blockDim.x) shared_zero[i] = 0.0;      // camera block + p.q slot of the pass that starts here
    __shared__ double red[16][4];
    if (pd.on) {        // sharded, peer-to-peer transport: this rank's slice sums -> acc, summed over the ranks right here (no reduction launches)
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = threadIdx.x; i < nblk; i += blockDim.x) { for (int k = 0; k < 4; ++k) v[k] += partials[4 * (size_t)i + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
        if ((threadIdx.x & 63) == 0) { for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = v[k]; }
        __syncthreads();
        if (threadIdx.x < 4) { double t = st->acc[threadIdx.x]; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w][threadIdx.x]; st->acc[threadIdx.x] = t; }
        __syncthreads();
        p2p_allreduce_wg(pd, st->acc, 4);
        nblk = 0;
    }
    const int NS = 6 * K + 9;
    const float alpha = (float)st->alpha;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (mode != STEP_INIT) {
        for (int i = threadIdx.x; i < NS; i += blockDim.x) {
            const size_t j = to + i;
            float xi = x[j], ri;
            if (mode == STEP_NORMAL) { xi += alpha * p[j]; x[j] = xi; ri = r[j] - alpha * q[j]; }
            else ri = b[j] - q[j];                                   // RESET: q holds tmp = A x, x was updated by k_pcg_tail_x
            r[j] = ri;
            const double xd = xi; s1 += xd * ((double)b[j] + (double)ri); s2 += xd * (double)ri; s3 += (double)D2[j] * xd * xd;
        }
    }
    __syncthreads();                                                 // the whole tail of r is final: the block preconditioner mixes entries
    for (int i = threadIdx.x; i < NS; i += blockDim.x) {
        int base, n, row; const float* M;
        if (i < 6 * K) { const int f = i / 6; base = 6 * f; n = 6; row = i - base; M = Mblk + 36 * f; }
        else if (i < 6 * K + 4) { base = 6 * K; n = 4; row = i - base; M = Mblk + 36 * K; }
        else { base = 6 * K + 4; n = 5; row = i - base; M = Mblk + 36 * K + 16; }
        const float* rs = r + to;
        float s = 0.0f;
        for (int j = 0; j < n; ++j) s += M[row * n + j] * rs[base + j];
        z[to + i] = s; s0 += (double)rs[i] * (double)s;
    }
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {          // slice sums of k_pcg_step (single rank; sharded ranks reduce + all-reduce them into acc first)
        s0 += partials[4 * (size_t)i]; s1 += partials[4 * (size_t)i + 1]; s2 += partials[4 * (size_t)i + 2]; s3 += partials[4 * (size_t)i + 3];
    }
    double v[4] = {s0, s1, s2, s3};
#pragma unroll
    for (int k = 0; k < 4; ++k) for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    if ((threadIdx.x & 63) == 0) { for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = v[k]; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot[4];
    for (int k = 0; k < 4; ++k) { double t = st->acc[k]; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w][k]; tot[k] = t; st->acc[k] = 0.0; }
    st->pq = 0.0;
    bool stop = false;
    if (mode != STEP_INIT) {                                         // end of iteration it: quadratic-model termination (eta = 0.1)
        st->xbr = tot[1]; st->xr = tot[2]; st->d2xx = tot[3];
        const int it = st->it + 1; st->it = it;
        const double Q1 = -tot[1]; st->Q1 = Q1;
        if (st->fixed_iterations >= 0) { st->Q0 = Q1; if (it >= st->fixed_iterations) { st->done = 1; stop = true; } }
        else {
            const double zeta = (double)it * (Q1 - st->Q0) / Q1;
            if (zeta < 0.1) { st->done = 1; stop = true; }
            else { st->Q0 = Q1; if (it >= st->max_iterations) { st->done = 1; stop = true; } }
        }
    }
    if (!stop) {                                                     // start of the next iteration: rho = r.z, beta
        const double rho = tot[0];
        if (rho == 0.0 || isinf(rho) || isnan(rho)) { st->done = 2; stop = true; }
        else {
            if (st->it > 0) { const double beta = rho / st->rho; if (beta == 0.0 || isinf(beta) || isnan(beta)) { st->done = 2; stop = true; } else st->beta = beta; }
            else st->beta = 0.0;
            if (!stop) { st->last_rho = st->rho; st->rho = rho; }
        }
    }
    publish();
}

// p = z + beta p ; u = S p over the two segments of a slice (+ `ntail` trailing scalars at `tail_off`: the camera tail, when it is
// handled here); d2_partials (optional): per-workgroup sums of D^2 p^2 (the diagonal part of p.q, see tile_pass.hip)
__global__ void __launch_bounds__(256) k_pcg_direction(int n4, int seg4, size_t tail_rel, int ntail, const float* __restrict__ z, float* __restrict__ p, const float* __restrict__ S,
                                                       float* __restrict__ u, const float* __restrict__ D2, double* __restrict__ d2_partials, const PcgState* __restrict__ state) {
    if (state->done) return;
    const float beta = (float)state->beta; const bool first = state->it == 0;
    const float4* z4 = reinterpret_cast<const float4*>(z); float4* p4 = reinterpret_cast<float4*>(p);
    const float4* S4 = reinterpret_cast<const float4*>(S); float4* u4 = reinterpret_cast<float4*>(u);
    const float4* D4 = reinterpret_cast<const float4*>(D2);
    double d2 = 0.0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < 2 * n4; j += gridDim.x * blockDim.x) {
        const int i = j < n4 ? j : j - n4 + seg4;
        float4 pi = z4[i];
        if (!first) { const float4 po = p4[i]; pi.x += beta * po.x; pi.y += beta * po.y; pi.z += beta * po.z; pi.w += beta * po.w; }
        p4[i] = pi;
        const float4 sv = S4[i];
        u4[i] = make_float4(sv.x * pi.x, sv.y * pi.y, sv.z * pi.z, sv.w * pi.w);
        if (d2_partials) { const float4 dd = D4[i]; d2 += (double)dd.x * (double)pi.x * (double)pi.x + (double)dd.y * (double)pi.y * (double)pi.y + (double)dd.z * (double)pi.z * (double)pi.z + (double)dd.w * (double)pi.w * (double)pi.w; }
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntail; t += gridDim.x * blockDim.x) {
        const size_t i = tail_rel + t;
        const float pi = first ? z[i] : z[i] + beta * p[i]; p[i] = pi; u[i] = S[i] * pi;      // (the tail's D^2 p^2 is added by k_pcg_tail_b: it is replicated when sharded)
    }
    if (d2_partials) block_partial_d(d2, d2_partials, 1, 0);
}

// camera tail of q = (S J^T W J S + D^2) p from the (all-reduced) fp64 camera block, p.q, alpha = rho / p.q
__global__ void __launch_bounds__(1024) k_pcg_tail_b(size_t to, int K, OptParams p, double* shared, const double* pq_slice,
                                                    const double* __restrict__ pq_partials, int nblk, const double* __restrict__ pq_partials2, int nblk2, int rowwise,
                                                    float* __restrict__ q, const float* __restrict__ S, const float* __restrict__ D2, const float* __restrict__ v, PcgState* st, P2PDev pd) {
    if (st->done) return;
    __shared__ double red[16];
    const int NS = 6 * K + 9;
    if (pd.on) {        // sharded, peer-to-peer transport: this rank's p.q (rows + D^2 p^2 of its slice) joins the camera block, [6K+9 | p.q] is summed over the ranks here
        double d = 0.0;
        for (int i = threadIdx.x; i < nblk; i += blockDim.x) d += pq_partials[i];
        for (int i = threadIdx.x; i < nblk2; i += blockDim.x) d += pq_partials2[i];
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) { double t = shared[NS]; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w]; shared[NS] = t; }
        __syncthreads();
        p2p_allreduce_wg(pd, shared, NS + 1);
        nblk = 0; nblk2 = 0;
    }
    double dotp = 0.0;
    for (int i = threadIdx.x; i < NS; i += blockDim.x) {
        const bool fixed = i < 6 * K ? p.fix_poses : (i < 6 * K + 4 ? p.fix_intr : p.fix_dist);
        const size_t j = to + i;
        const float vv = v[j];
        const float o = S[j] * (fixed ? 0.0f : (float)shared[i]) + D2[j] * vv;
        q[j] = o;
        if (!rowwise) dotp += (double)vv * (double)o;      // rowwise: p.q = sum_rows t (J u) [partials, camera columns included] + sum D^2 p^2 [partials2: voxel part; camera part here]
        else dotp += (double)D2[j] * (double)vv * (double)vv;
    }
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) dotp += pq_partials[i];       // workgroup partials of the voxel part (k_gather) / of the rows (k_eg_tile)
    for (int i = threadIdx.x; i < nblk2; i += blockDim.x) dotp += pq_partials2[i];     // workgroup partials of sum D^2 p^2 (k_pcg_direction)
    for (int o = 32; o > 0; o >>= 1) dotp += __shfl_down(dotp, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dotp;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double pq = *pq_slice; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) pq += red[w];
    st->pq = pq;
    if (!(pq > 0.0) || isinf(pq)) { st->done = 2; return; }
    const double alpha = st->rho / pq;
    if (isinf(alpha)) { st->done = 2; return; }
    st->alpha = alpha;
}

__global__ void k_pcg_init(PcgState* st, int fixed_iterations, int max_iterations, const LmState* lm) {
    const bool over = lm && lm->done;
    for (int k = 0; k < 4; ++k) st->acc[k] = 0.0;
    st->rho = 0.0; st->last_rho = 1.0; st->pq = 0.0; st->alpha = 0.0; st->beta = 0.0; st->xbr = 0.0; st->xr = 0.0; st->d2xx = 0.0;
    st->Q0 = 0.0; st->Q1 = 0.0; st->it = 0; st->done = (fixed_iterations == 0 || over) ? 1 : 0; st->fixed_iterations = fixed_iterations; st->max_iterations = max_iterations;
}

void launch_pcg_init(hipStream_t st, PcgState* state, int fixed_iterations, int max_iterations, const LmState* lm) { k_pcg_init<<<1, 1, 0, st>>>(state, fixed_iterations, max_iterations, lm); }
static inline int step_blocks(int n4) { int b = (n4 + 255) / 256; return b < 1 ? 1 : (b > 2048 ? 2048 : b); }      // 64 VGPRs: 8 waves per SIMD = 2048 workgroups of 256 in flight
// off and n must be multiples of 4 (the rank-major layout pads every slice to a multiple of 8 floats)
int launch_pcg_step(hipStream_t st, int mode, Seg2 sg, const float* p, const float* q, float* x, float* r, const float* b, const float* D2, const float* Minv,
                    float* z, const float* S_for_inline_q, double* partials, PcgState* state) {
    if (sg.n <= 0) return 0;
    const int n4 = sg.n >> 2, seg4 = (int)((sg.off1 - sg.off0) >> 2);
    const size_t off = sg.off0;
    auto c4 = [off](const float* v) { return reinterpret_cast<const float4*>(v ? v + off : nullptr); };
    auto m4 = [off](float* v) { return reinterpret_cast<float4*>(v + off); };
    const float* S = S_for_inline_q;
    const int blocks = step_blocks(2 * n4);
#define I3D_STEP(MODE, QI) k_pcg_step<MODE, QI><<<blocks, 256, 0, st>>>(n4, seg4, c4(p), c4(q), m4(x), m4(r), c4(b), c4(D2), c4(Minv), m4(z), c4(S), partials, state)
    switch (mode) {
        case STEP_INIT:   I3D_STEP(STEP_INIT, false); break;
        case STEP_NORMAL: if (S) I3D_STEP(STEP_NORMAL, true); else I3D_STEP(STEP_NORMAL, false); break;
        case STEP_XONLY:  I3D_STEP(STEP_XONLY, false); break;
        default:          if (S) I3D_STEP(STEP_RESET, true); else I3D_STEP(STEP_RESET, false); break;
    }
#undef I3D_STEP
    return mode == STEP_XONLY ? 0 : blocks;      // number of [4]-partials written
}
void launch_pcg_tail_x(hipStream_t st, size_t tail_off, int K, const float* p, float* x, const PcgState* state) { k_pcg_tail_x<<<1, 256, 0, st>>>(tail_off, 6 * K + 9, p, x, state); }
void launch_pcg_tail_a(hipStream_t st, int mode, size_t tail_off, int K, const float* Minv_blocks, const float* p, const float* q, float* x, float* r, const float* b,
                       const float* D2, float* z, const double* partials, int nblk, PcgState* state, double* shared_zero, int nzero, int* host_flags, int seq, const P2PDev& pd) {
    k_pcg_tail_a<<<1, 1024, 0, st>>>(mode, tail_off, K, Minv_blocks, p, q, x, r, b, D2, z, partials, nblk, state, shared_zero, nzero, host_flags, seq, pd);
}
int launch_pcg_direction(hipStream_t st, Seg2 sg, size_t tail_off, int ntail, const float* z, float* p, const float* S, float* u, const float* D2, double* d2_partials,
                         const PcgState* state) {
    if (sg.n <= 0 && ntail <= 0) return 0;
    const int n4 = sg.n >> 2, seg4 = (int)((sg.off1 - sg.off0) >> 2);
    const int blocks = step_blocks(2 * n4 > 0 ? 2 * n4 : 1);
    const size_t o = sg.off0;
    k_pcg_direction<<<blocks, 256, 0, st>>>(n4, seg4, tail_off - o, ntail, z + o, p + o, S + o, u + o, D2 + o, d2_partials, state);
    return d2_partials ? blocks : 0;
}
void launch_pcg_tail_b(hipStream_t st, size_t tail_off, int K, OptParams p, double* shared, const double* pq_slice, const double* pq_partials, int nblk,
                       const double* pq_partials2, int nblk2, bool rowwise, float* q, const float* S, const float* D2, const float* v, PcgState* state, const P2PDev& pd) {
    k_pcg_tail_b<<<1, 1024, 0, st>>>(tail_off, K, p, shared, pq_slice, pq_partials, nblk, pq_partials2, nblk2, rowwise ? 1 : 0, q, S, D2, v, state, pd);
}

// ---- LM candidate / acceptance ------------------------------------------------------------------------------------------
// candidate point x + S*step (TrustRegionMinimizer: delta = step .* jacobian_scaling), squared norms of delta and x over the free
// parameters.  Replicated: every rank holds the full step and S vectors (all-gathered) and updates the whole list.
__global__ void __launch_bounds__(256) k_candidate(GridView g, RowView r, int K, float sign, const float* __restrict__ step, const float* __restrict__ S,
                                                   const double* __restrict__ xsh, double* xc_sdf, double* xc_alb, double* xc_sh,
                                                   double* norms2, const float* __restrict__ mask, const LmState* __restrict__ lm) {
    if (lm && lm->done) return;           // (a later attempt queued before the host knew that the solve had ended: the accepted candidate must survive)
    const int A = r.A, NS = 6 * K + 9, chunk = r.chunk;
    const size_t tail = 2 * (size_t)chunk;
    double d2 = 0.0, x2 = 0.0;
    GRID_STRIDE(A + NS) {
        if (i < A) {
            const int s = r.alist[i]; const int js = vec_sdf(i, chunk), ja = js + chunk;
            const double ds = (double)sign * (double)step[js] * (double)S[js], da = (double)sign * (double)step[ja] * (double)S[ja];
            const double xs = g.x_sdf[s], xa = g.x_alb[s];
            xc_sdf[s] = xs + ds; xc_alb[s] = xa + da;
            if (mask[js] != 0.0f) { d2 += ds * ds; x2 += xs * xs; }
            if (mask[ja] != 0.0f) { d2 += da * da; x2 += xa * xa; }
        } else {
            const int t = i - A; const size_t j = tail + t;
            const double delta = (double)sign * (double)step[j] * (double)S[j];
            const double x = xsh[t]; xc_sh[t] = x + delta;
            if (mask[j] != 0.0f) { d2 += delta * delta; x2 += x * x; }
        }
    }
    block_partial_d(d2, norms2, 2, 0); block_partial_d(x2, norms2, 2, 1);
}
void launch_candidate(hipStream_t st, GridView g, RowView r, int K, float sign, const float* step, const float* S, const double* xsh, double* xc_sdf, double* xc_alb,
                      double* xc_sh, double* norms2, const float* mask, double* scratch, const LmState* lm) {
    int b = vblocks(r.A + 6 * K + 9); if (b > 1024) b = 1024;
    k_candidate<<<b, 256, 0, st>>>(g, r, K, sign, step, S, xsh, xc_sdf, xc_alb, xc_sh, scratch, mask, lm);
    launch_reduce_partials(st, scratch, b, 2, norms2, nullptr, true);      // (after a finished solve: stale partials into a slot nobody reads)
}
// x <- candidate on the work list (everything else never moves), refresh the fp32 shadows
__global__ void k_accept(GridView g, RowView r, const double* __restrict__ xc_sdf, const double* __restrict__ xc_alb, const LmState* __restrict__ lm) {
    if (lm && !lm->accepted) return;
    GRID_STRIDE(r.A) { const int s = r.alist[i]; const double a = xc_sdf[s], b = xc_alb[s]; g.x_sdf[s] = a; g.x_alb[s] = b; g.f_sdf[s] = (float)a; g.f_alb[s] = (float)b; }
}
void launch_accept(hipStream_t st, GridView g, RowView r, const double* xc_sdf, const double* xc_alb, const LmState* lm) { if (r.A > 0) k_accept<<<vblocks(r.A), 256, 0, st>>>(g, r, xc_sdf, xc_alb, lm); }

// halo exchange of the operator input (Comm::push_halo): gather the rim values a peer needs / scatter what the peers sent
__global__ void k_halo_pack(int n, const int* __restrict__ idx, const float* __restrict__ vec, int chunk, float* __restrict__ buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int e = idx[i]; buf[2 * (size_t)i] = vec[e]; buf[2 * (size_t)i + 1] = vec[(size_t)chunk + e]; }
}
__global__ void k_halo_unpack(int n, const int* __restrict__ idx, const float* __restrict__ buf, int chunk, float* __restrict__ vec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int e = idx[i]; vec[e] = buf[2 * (size_t)i]; vec[(size_t)chunk + e] = buf[2 * (size_t)i + 1]; }
}
// the rim of nsys vectors of a ladder batch in one buffer: entry j carries [system][sdf value, albedo value] side by side, so a peer's block stays contiguous
__global__ void k_halo_pack_multi(int n, const int* __restrict__ idx, const float* __restrict__ vec0, size_t stride, HaloSys sys, int nsys, int chunk, float* __restrict__ buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nsys) return;
    const int j = i / nsys, s = i - j * nsys, e = idx[j];
    const float* v = vec0 + (size_t)sys.id[s] * stride;
    buf[2 * (size_t)i] = v[e]; buf[2 * (size_t)i + 1] = v[(size_t)chunk + e];
}
__global__ void k_halo_unpack_multi(int n, const int* __restrict__ idx, const float* __restrict__ buf, size_t stride, HaloSys sys, int nsys, int chunk, float* __restrict__ vec0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nsys) return;
    const int j = i / nsys, s = i - j * nsys, e = idx[j];
    float* v = vec0 + (size_t)sys.id[s] * stride;
    v[e] = buf[2 * (size_t)i]; v[(size_t)chunk + e] = buf[2 * (size_t)i + 1];
}
void launch_halo_pack_multi(hipStream_t st, int n, const int* idx, const float* vec0, size_t stride, HaloSys sys, int nsys, int chunk, float* buf) {
    if (n > 0 && nsys > 0) k_halo_pack_multi<<<(n * nsys + 255) / 256, 256, 0, st>>>(n, idx, vec0, stride, sys, nsys, chunk, buf);
}
void launch_halo_unpack_multi(hipStream_t st, int n, const int* idx, const float* buf, size_t stride, HaloSys sys, int nsys, int chunk, float* vec0) {
    if (n > 0 && nsys > 0) k_halo_unpack_multi<<<(n * nsys + 255) / 256, 256, 0, st>>>(n, idx, buf, stride, sys, nsys, chunk, vec0);
}
void launch_halo_pack(hipStream_t st, int n, const int* idx, const float* vec, int chunk, float* buf) { if (n > 0) k_halo_pack<<<(n + 255) / 256, 256, 0, st>>>(n, idx, vec, chunk, buf); }
void launch_halo_unpack(hipStream_t st, int n, const int* idx, const float* buf, int chunk, float* vec) { if (n > 0) k_halo_unpack<<<(n + 255) / 256, 256, 0, st>>>(n, idx, buf, chunk, vec); }

// compute list of a rank: owned entries + every entry whose Eg rows (forward stencil) or regulariser rows (ring) touch an owned entry
__global__ void k_mark_compute(RowView r, int* __restrict__ flag) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= r.A) return;
    flag[a] = shard_needs_entry(a, r.own0, r.own1, (r.aflags[a] & F_ACTIVE) != 0, r.anbr, r.Acap) ? 1 : 0;
}
void launch_mark_compute(hipStream_t st, RowView r, int* flag) { if (r.A > 0) k_mark_compute<<<(r.A + 255) / 256, 256, 0, st>>>(r, flag); }
__global__ void k_compact_list(int A, const int* __restrict__ flag, const int* __restrict__ scan, int* __restrict__ list) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < A && flag[a]) list[scan[a]] = a;
}
void launch_compact_list(hipStream_t st, int A, const int* flag, const int* scan, int* list) { if (A > 0) k_compact_list<<<(A + 255) / 256, 256, 0, st>>>(A, flag, scan, list); }

}  // namespace i3d
