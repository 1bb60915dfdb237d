// The gradient J^T W r AND the column norms diag(J^T W J) (+ the camera blocks of J^T W J) of an outer iteration from ONE stream of the Eg rows.
//
// TrustRegionMinimizer::Init needs both before the first LM attempt [Ceres 2.1.0 trust_region_minimizer.cc: the Jacobi scaling from the column norms, then the
// gradient of the scaled problem; selected at nls_solver.cpp:307].  k_eg_pass<PASS_COLNORM> and k_eg_pass<PASS_GRAD> (operator.hip) each stream the
// 120 B rows once (0.48 + 0.42 ms on the 8 M-voxel workload); here a row is read once and feeds both sets of sums.  Outputs keep the two-pass layout
// (staging planes C / treg per pass for k_gather, one float row of camera totals per workgroup for k_sum_rows) so that everything downstream is unchanged.
//
// 384-thread workgroups, two per CU (3 waves per SIMD, 153 registers): the kernel carries 28 voxel-column sums, 25 + 9 shared-column sums and one row (29 partials) per lane, which does not fit the
// 128 registers a 1024-thread workgroup leaves a lane.  Sums are fixed-order as in k_eg_pass (wave-private keyframe tables merged in wave order, per-wave slots
// for the shared columns, one row per workgroup): bit-reproducible.
#include "kernels.hpp"
#include "reduce_device.hpp"
#include "wave_ops.hpp"

namespace i3d {

constexpr int GC_THREADS = 384, GC_NW = GC_THREADS / 64, GC_TC = 16, GC_NPV = 27, GC_NCAM = 34;      // 6 gradient + 21 block entries per keyframe; 9 gradient + 10 + 15 block entries of intrinsics / distortion
constexpr int GC_D_CAMW = 4, GC_D_TAG = GC_D_CAMW + ((GC_NW * (GC_NCAM + 0) + 3) & ~3), GC_D_VAL = GC_D_TAG + GC_NW * GC_TC, GC_D0 = GC_D_VAL + GC_NW * GC_TC * GC_NPV;

// upper triangle of a 6 x 6 block, row-major: entry o = (TRI_I[o], TRI_J[o])
__device__ constexpr int8_t GC_TRI_I[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__device__ constexpr int8_t GC_TRI_J[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};

#ifndef GC_OCC
#define GC_OCC 2
#endif
__global__ void __launch_bounds__(GC_THREADS, GC_OCC) k_eg_gradcol(GridView g, RowView r, OptParams p, GradColBuffers b, int tiles_per_block) {
    extern __shared__ float lds[];        // [4] | [NW][NCAM] shared-column slots | [NW][TC] tags | [NW][TC][27] | dense [K][27] | [NCAM]
    const int K = p.K; const size_t Acap = r.Acap;
    const int rs = (GC_NPV * K) | 1;
    const int nacc = GC_D0 + rs + GC_NCAM;
    for (int i = threadIdx.x; i < nacc; i += GC_THREADS) lds[i] = (i >= GC_D_TAG && i < GC_D_VAL) ? __int_as_float(-1) : 0.0f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int o_tag = GC_D_TAG + wave * GC_TC, o_val = GC_D_VAL + wave * (GC_TC * GC_NPV);
    int tcount = 0;
    float* const cam_acc = lds + GC_D0 + rs;
    float cam9[9];                        // gradient: intrinsics + distortion columns
#pragma unroll
    for (int i = 0; i < 9; ++i) cam9[i] = 0.0f;
    float camb[25];                       // J^T W J blocks of the intrinsics (10) and the distortion (15), upper triangles (their diagonals are the squared column norms)
#pragma unroll
    for (int i = 0; i < 25; ++i) camb[i] = 0.0f;
    const float tw0 = (float)p.type_w[0];
    const bool free_poses = !p.fix_poses;
    double cost = 0.0;                    // 0.5 sum w r^2 over the rows of the reduced program (what k_build<false> returns at this point), on the rows' owner
    const int nC = r.nC;
    const int ntiles = (nC + GC_THREADS - 1) / GC_THREADS;
    const int tile0 = blockIdx.x * tiles_per_block;

    for (int tile = tile0; tile < tile0 + tiles_per_block && tile < ntiles; ++tile) {
        const int ci = tile * GC_THREADS + threadIdx.x;
        const bool in = ci < nC;
        const int a = in ? (r.clist ? r.clist[ci] : ci) : 0;
        const bool owned = in && a >= r.own0 && a < r.own1;           // camera columns are accumulated once: by the row's owner
        const uint8_t fl = in ? r.aflags[a] : 0;
        const int nr = (in && (fl & F_ACTIVE)) ? (int)r.nrows[a] : 0;
        int nr_max = nr;
        for (int o = 32; o > 0; o >>= 1) nr_max = max(nr_max, __shfl_xor(nr_max, o, 64));
        float accg[P_VOX], accc[P_VOX];
#pragma unroll
        for (int c = 0; c < P_VOX; ++c) { accg[c] = 0.0f; accc[c] = 0.0f; }
        const size_t ac = in ? (size_t)a : 0;
        // the row behind the one being consumed is always in flight (round 6: the loop exposed one memory round trip per row slot — wait share 0.77 at 3 waves per SIMD): slot k + 1 is
        // requested before slot k is used; the last slot asks for itself again (unconditional loads: the compiler's wait counts stay exact)
        float4 n4[7]; float2 njt, nwr;
        auto request = [&](int k) {
            const float4* __restrict__ row = r.rows + row_index(ac, k, 0, r.slots);
#pragma unroll
            for (int q = 0; q < 7; ++q) n4[q] = ld_row(row + q * 64);
            njt = r.row_jt()[row_jt_index(ac, k, r.slots)];
            nwr = r.row_wr[row_scalar_index(ac, k, r.slots)];
        };
        if (nr_max > 0) request(0);
        for (int k = 0; k < nr_max; ++k) {
            float4 j4[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) j4[q] = n4[q];
            const float2 jt = njt, wr_k = nwr;
            request(k + 1 < nr_max ? k + 1 : k);
            const bool live = k < nr;
            float J[P_TOTAL];
#pragma unroll
            for (int q = 0; q < 7; ++q) { J[4 * q] = j4[q].x; J[4 * q + 1] = j4[q].y; J[4 * q + 2] = j4[q].z; J[4 * q + 3] = j4[q].w; }
            J[28] = jt.x;
            float t = 0.0f; int fsel = 0;
            if (live) {
                const float2 wr = wr_k;
                t = tw0 * (sqrtf(wr.x) * wr.y);               // Js^T (tw sqrt(w) r) = J^T W r  (the row weight is folded into the stored partials)
                fsel = __float_as_int(jt.y) & ~ROW_FREE_BIT;
#pragma unroll
                for (int c = 0; c < P_VOX; ++c) { accg[c] += J[c] * t; accc[c] += tw0 * J[c] * J[c]; }
                if (owned) {
                    if (__float_as_int(jt.y) & ROW_FREE_BIT) cost += 0.5 * (double)wr.x * p.type_w[0] * ((double)wr.y * (double)wr.y);      // (the stored residual: fp32-rounded)
#pragma unroll
                    for (int i = 0; i < 9; ++i) cam9[i] += J[P_INTR + i] * t;
                    int o = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = i; j < 4; ++j) { camb[o] += tw0 * J[P_INTR + i] * J[P_INTR + j]; ++o; }
                    }
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
#pragma unroll
                        for (int j = i; j < 5; ++j) { camb[o] += tw0 * J[P_DIST + i] * J[P_DIST + j]; ++o; }
                    }
                }
            }
            // pose columns of this slot: 6 gradient entries + the upper triangle of the keyframe's 6 x 6 block
            wave_table_add_quads<GC_NPV, GC_TC, 3>(live && owned, fsel, [&](int q) {
                return q < 6 ? (free_poses ? J[P_POSE + (q < 6 ? q : 0)] * t : 0.0f) : tw0 * J[P_POSE + GC_TRI_I[q < 6 ? 0 : q - 6]] * J[P_POSE + GC_TRI_J[q < 6 ? 0 : q - 6]];
            }, lds, o_tag, o_val, tcount, GC_D0, GC_NPV);
        }
        __syncthreads();
        for (int w = 0; w < GC_NW; ++w) { if (wave == w) wave_table_merge<GC_NPV, GC_TC>(lds, o_tag, o_val, tcount, GC_D0, GC_NPV); __syncthreads(); }
        if (in) {
#pragma unroll
            for (int c = 0; c < P_VOX; ++c) { b.Cg[(size_t)c * Acap + a] = accg[c]; b.Cc[(size_t)c * Acap + a] = accc[c]; }
            // ---- regulariser rows: tr (Er), ts (Es, Jacobian folded in), ta[6] (Ea): weighted residual (gradient) / weight (column norms) ----
            // Everything they read is requested in TWO batches, unconditionally (an entry without the row reads its own voxel; the flags select afterwards): the entry's bytes, Ea
            // weights and ring, then the ring's sdf / albedo values.  Behind their conditions the Ea rows alone were a chain of weight -> neighbour -> albedo per direction: up to 18
            // round trips per entry at ONE workgroup per CU.
            const int s = r.alist[a];
            const int N = g.N;
            const bool act = (fl & F_ACTIVE) != 0;
            const uint8_t rf_ld = r.regflags[a], eaf_ld = r.ea_free[a];
            float eaw[6]; int rg[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) { eaw[d] = r.ea_w[(size_t)d * Acap + a]; rg[d] = g.nbr[(size_t)d * N + s]; }
            const double xs = g.x_sdf[s], xa = g.x_alb[s], s0 = g.sdf0[s];
            double xsn[6], xan[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) { const int nb = rg[d] >= 0 ? rg[d] : s; xsn[d] = g.x_sdf[nb]; xan[d] = g.x_alb[nb]; }
            const uint8_t rf = act ? rf_ld : 0;
            float trg = 0.0f, tsg = 0.0f, trc = 0.0f, tsc = 0.0f;
            const uint8_t eafree = (owned && act) ? eaf_ld : 0;
            if (rf & 1) {
                const float rho = (float)p.type_w[1];
                const double dxx = xsn[0] + xsn[1] - 2.0 * xs, dyy = xsn[2] + xsn[3] - 2.0 * xs, dzz = xsn[4] + xsn[5] - 2.0 * xs;
                const double lap = dxx + dyy + dzz;
                trg = rho * (float)lap; trc = rho;
                if (owned && (rf & 8)) cost += 0.5 * p.type_w[1] * lap * lap;
            }
            if (rf & 2) {
                const double e0 = xs - s0;
                if (rf & 4) { const float rho = (float)p.type_w[2]; tsg = rho * (float)e0; tsc = rho; }
                if (owned && (rf & 16)) { const double e = e0 == 0.0 ? 0.0000001 : e0; cost += 0.5 * p.type_w[2] * e * e; }      // surface_stab_regularizer.h:62-64
            }
            b.tregg[a] = trg; b.tregg[Acap + a] = tsg; b.tregc[a] = trc; b.tregc[Acap + a] = tsc;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const float w = act ? eaw[d] : 0.0f;
                float tag = 0.0f, tac = 0.0f;
                if (w != 0.0f) {
                    const float rho = w * (float)p.type_w[3]; const double e = xa - xan[d];
                    tag = rho * (float)e; tac = rho;
                    if (eafree & (1 << d)) cost += 0.5 * (double)w * p.type_w[3] * e * e;
                }
                b.tregg[(size_t)(2 + d) * Acap + a] = tag; b.tregc[(size_t)(2 + d) * Acap + a] = tac;
            }
        }
    }
    block_partial_d(cost, b.cost_partials, 1, 0);          // per-workgroup partial, summed by k_reduce_partials
    // columns shared by every row: wave-shuffle reduction into the wave's slot, the slots added in wave order
#pragma unroll
    for (int i = 0; i < GC_NCAM; ++i) {
        float v = i < 9 ? cam9[i] : camb[i - 9];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) lds[GC_D_CAMW + wave * GC_NCAM + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < GC_NCAM) { float v = 0.0f; for (int w = 0; w < GC_NW; ++w) v += lds[GC_D_CAMW + w * GC_NCAM + threadIdx.x]; cam_acc[threadIdx.x] = v; }
    __syncthreads();
    // this workgroup's totals as two float rows in the layout of the two-pass kernels: gradient [6K | 9] at b.part, column norms [21K | 9 squares | 10 | 15] at b.part + b.col_off
    float* const rowg = b.part + (size_t)blockIdx.x * b.part_stride;
    float* const rowc = rowg + b.col_off;
    for (int i = threadIdx.x; i < 6 * K; i += GC_THREADS) { const int f = i / 6; rowg[i] = lds[GC_D0 + GC_NPV * f + (i - 6 * f)]; }
    for (int i = threadIdx.x; i < 21 * K; i += GC_THREADS) { const int f = i / 21; rowc[i] = lds[GC_D0 + GC_NPV * f + 6 + (i - 21 * f)]; }
    if (threadIdx.x < 9) rowg[6 * K + threadIdx.x] = cam_acc[threadIdx.x];
    if (threadIdx.x < 34) {
        const int i = threadIdx.x;
        // the squared norms of the 9 shared columns are the diagonals of the two blocks: intrinsics entries 0, 4, 7, 9 of 10, distortion entries 0, 5, 9, 12, 14 of 15
        constexpr int8_t DIAG[9] = {0, 4, 7, 9, 10, 15, 19, 22, 24};
        rowc[21 * K + i] = i < 9 ? cam_acc[9 + DIAG[i]] : cam_acc[i];
    }
}

// returns the number of workgroups = rows written to b.part
int launch_eg_gradcol(hipStream_t st, GridView g, RowView r, OptParams p, GradColBuffers b) {
    if (r.nC <= 0) return 0;
    static int num_cu = 0;
    if (!num_cu) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev); num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    const int ntiles = (r.nC + GC_THREADS - 1) / GC_THREADS;
    const int want = 2 * num_cu;                                          // two persistent workgroups per CU
    const int blocks = ntiles < want ? ntiles : want;
    const int tiles_per_block = (ntiles + blocks - 1) / blocks;
    const size_t n = (size_t)(GC_D0 + ((GC_NPV * p.K) | 1) + GC_NCAM) * sizeof(float);
    if (!set_dynamic_lds((const void*)k_eg_gradcol, "k_eg_gradcol", n, p.K)) return 0;
    const int used = (ntiles + tiles_per_block - 1) / tiles_per_block;    // workgroups that own a tile (the others would write zero rows)
    k_eg_gradcol<<<used, GC_THREADS, n, st>>>(g, r, p, b, tiles_per_block);
    launch_reduce_partials(st, b.cost_partials, used, 1, b.cost_out, nullptr);      // *cost_out += sum
    return used;
}

}  // namespace i3d
