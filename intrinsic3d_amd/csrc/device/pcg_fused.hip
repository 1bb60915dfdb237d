// The PCG iteration in THREE launches (single rank; conjugate_gradients_solver.cc of Ceres 2.1.0 as driven by nls_solver.cpp:296-337):
//
//     k_pcg_dir3  ->  k_eg_tile  ->  k_pcg_step3
//
// Round 2 ran six (tail_a | direction | eg_tile | halo_fold | tail_b | step): the two single-workgroup boundary kernels and the halo fold
// cost 82 us of a 410 us pass for 0.3 GB of traffic.  Here every grid-wide reduction is finished REDUNDANTLY in the prologue of the kernel
// that needs its result — each workgroup adds up the same few thousand per-workgroup partial sums in the same fixed order, so all of them
// hold bit-identical scalars without a launch or an in-kernel barrier:
//   * k_pcg_dir3  : [r.z, x.(b+r), x.r, sum D^2 x^2] of the iteration just finished -> Ceres' quadratic-model stop test, rho, beta (workgroup 0
//                   also records them and publishes (pass, done) to the host); then p = z + beta p, u = S p, partial sums of D^2 p^2;
//   * k_eg_tile   : q_acc = J^T W J u (tile_pass.hip), p.q row by row; the camera block leaves as one float partial per workgroup (no atomics);
//   * k_pcg_step3 : p.q = rows + D^2 p^2 -> alpha; folds the tiles' halo sums into the accumulators (the former k_halo_fold: pairs sorted by
//                   entry, CSR offsets from the plan), x += alpha p, q = S acc + D^2 p, r -= alpha q, z = M^-1 r and the four partial sums; a
//                   handful of extra workgroups do the same for the 6K+9 camera unknowns: column sums of the camera partials in a fixed
//                   order, block-Jacobi z with the 6x6 / 4x4 / 5x5 inverses.
// The scalar state is double-buffered by pass parity (a kernel never writes the PcgState it reads), so there is no same-kernel race on it.
// Everything here is deterministic: no floating-point atomics, fixed summation orders.
#include "kernels.hpp"
#include "reduce_device.hpp"
#include <cstring>

namespace i3d {

constexpr int PF_THREADS = 512;
constexpr int PF_MAX_WG = 512;            // slice workgroups of dir3 / step3 (grid-stride beyond)
constexpr int PF_POSES_PER_WG = 10;        // camera tail: poses handled by one tail workgroup of k_pcg_step3

// sum of nblk partial [NC]-tuples, identical (bit for bit) in every workgroup of PF_THREADS threads that calls it
template <int NC>
static __device__ inline void reduce_partials_all(const double* __restrict__ P, int nblk, double (&tot)[NC], double* sm /* [NC * 8] */,
                                                  const double* __restrict__ P2 = nullptr, int nblk2 = 0 /* a second list added to the same totals */) {
    double v[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) v[k] = 0.0;
    for (int i = threadIdx.x; i < nblk; i += PF_THREADS) {
#pragma unroll
        for (int k = 0; k < NC; ++k) v[k] += P[(size_t)i * NC + k];
    }
    for (int i = threadIdx.x; i < nblk2; i += PF_THREADS) {
#pragma unroll
        for (int k = 0; k < NC; ++k) v[k] += P2[(size_t)i * NC + k];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NC; ++k) sm[k * 8 + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NC; ++k) { double t = 0.0; for (int w = 0; w < PF_THREADS / 64; ++w) t += sm[k * 8 + w]; tot[k] = t; }
    __syncthreads();
}

__global__ void k_pcg_init3(PcgState* st2, int fixed_iterations, int max_iterations, const LmState* lm) {
    const bool over = lm && lm->done;           // the LM loop ended before this (speculatively queued) solve: every kernel of it returns at once
    for (int b = 0; b < 2; ++b) {
        PcgState* st = st2 + b;
        for (int k = 0; k < 4; ++k) st->acc[k] = 0.0;
        st->rho = 0.0; st->last_rho = 1.0; st->pq = 0.0; st->alpha = 0.0; st->beta = 0.0; st->xbr = 0.0; st->xr = 0.0; st->d2xx = 0.0;
        st->Q0 = 0.0; st->Q1 = 0.0; st->it = 0; st->done = (fixed_iterations == 0 || over) ? 1 : 0; st->fixed_iterations = fixed_iterations; st->max_iterations = max_iterations;
    }
}
void launch_pcg_init3(hipStream_t st, PcgState* st2, int fixed_iterations, int max_iterations, const LmState* lm) { k_pcg_init3<<<1, 1, 0, st>>>(st2, fixed_iterations, max_iterations, lm); }

// Jacobi scale S, LM diagonal D^2 and 1x1 block-Jacobi inverse M^-1 of a voxel unknown from its masked squared column norm (k_scale + k_lm_diag,
// operator.hip, expression for expression): the two vector kernels of a pass read ONE array instead of S, D^2 and M^-1 (24 B less per entry and pass)
static __device__ inline void lm_from_colnorm(float cm, float inv_radius, float& s, float& d2, float& minv) { s = lm_scale(cm); lm_diag(cm, s, inv_radius, d2, minv); }

// iteration boundary + direction.  `prev` was written by the previous boundary, `next` is read by the operator / step kernels of this pass.
// SH (sharded run over the peer-to-peer mailboxes, p2p_device.hpp): the ONE exchange of the boundary happens in here —
//   * workgroup 0 stores this rank's four slice sums into every rank's mailbox, every workgroup of every rank reads all contributions and adds them in rank
//     order (bit-identical scalars everywhere, no launch, no grid barrier); the sums of the replicated camera tail are added once, locally;
//   * the rim rides with it: a few extra workgroups store z on the owned entries the peers' rows read straight into the peers' mailboxes BEFORE they wait for
//     anything, and — once beta is known — rebuild p = z + beta p and u = S p on the foreign entries this rank's rows read (every rank holds the column norms of
//     all entries, so S is local).  Round 3 pushed u = S p with a launch of its own after the direction kernel, between two all-reduce launches.
// Workgroup roles (SH), by blockIdx: [0, n_rim) rim | n_rim .. n_rim + n_main - 1 the slice | the last one the camera tail.  Their partial sums of D^2 p^2 land
// at [0, n_main) (slice) and [n_main] (tail, counted once by k_pcg_step3 after ITS exchange).
// (the body of the kernel: shared by the single-system launch and the ladder launch, whose workgroups pick their system by blockIdx.y — the SAME arithmetic in the
// same order, so a system iterated in a ladder batch goes through bit for bit the states it goes through alone)
// SH = 2 (round 6: sharded run whose exchanges are LAUNCHES of the transport — RCCL, the rank simulation): the workgroup roles of SH = 1 without rim workgroups; the four
// slice sums arrive already summed over the ranks in sa_red4 (k_lad_reduce_step + one all-reduce in front of this kernel), the camera tail's partial rows are added locally.
template <int SH>
static __device__ __forceinline__ void pcg_dir3_body(int init, int n4, int seg4, size_t tail_rel, int ntail, const float* __restrict__ z, float* __restrict__ p,
                                                        const float* __restrict__ S, float* __restrict__ u, const float* __restrict__ tD2 /* LM diagonal of the camera tail, [ntail] */,
                                                        const float* __restrict__ cm, const float inv_radius, const double* __restrict__ step_partials, int n_step, double* __restrict__ d2_partials,
                                                        const PcgState* __restrict__ prev, PcgState* __restrict__ next, int* host_flags, int seq, const ShardArgs& sa,
                                                        const double* __restrict__ red4 = nullptr /* SH = 2: [4] slice sums over all ranks */) {
    __shared__ double sm[4 * 8];
    __shared__ double smx[SH == 1 ? 4 * P2P_MAX_RANKS : 1];
    // (seq, done) goes to a 2-slot ring in pinned host memory: the host polls it one pass behind
    auto publish = [&](int done) { if (host_flags) { __hip_atomic_store(&host_flags[2 * (seq & 1) + 1], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                                     __hip_atomic_store(&host_flags[2 * (seq & 1)], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } };
    // logical workgroup: 0 .. n_main - 1 slice, n_main the camera tail (SH only), beyond: rim
    const int n_rim = SH == 1 ? sa.n_rim_wg : 0, n_main = SH ? (int)gridDim.x - n_rim - 1 : (int)gridDim.x;
    const int lid = SH == 1 ? ((int)blockIdx.x < n_rim ? n_main + 1 + (int)blockIdx.x : (int)blockIdx.x - n_rim) : (int)blockIdx.x;
    const bool rim_wg = SH == 1 && lid > n_main, tail_wg = SH ? lid == n_main : lid == n_main - 1;
    const bool writer = lid == 0 && threadIdx.x == 0;
    const float4* z4 = reinterpret_cast<const float4*>(z); float4* p4 = reinterpret_cast<float4*>(p);
    const float4* C4 = reinterpret_cast<const float4*>(cm); float4* u4 = reinterpret_cast<float4*>(u);
    if (prev->done) { if (writer) { *next = *prev; publish(prev->done); } return; }      // (the same decision on every rank: the state is replicated bit for bit)
    double tot[4];
    if (!SH) reduce_partials_all<4>(step_partials, n_step, tot, sm);
    else if (SH == 2) {
        reduce_partials_all<4>(step_partials + 4 * (size_t)sa.n_slice_partials, n_step - sa.n_slice_partials, tot, sm);           // the camera tail (replicated)
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] += red4[k];
    } else {
        const unsigned e32 = p2p_pass_epoch(seq, P2P_X_DIR);
        if (rim_wg) {                                                // z on the owned rim -> the peers' mailboxes, before any wait
            const RimLists& rl = sa.rim;
            for (int j = (lid - n_main - 1) * PF_THREADS + threadIdx.x; j < rl.n_send; j += n_rim * PF_THREADS) {
                const int e = rl.send_idx[j], k = rl.send_peer[j];
                p2p_put_rim(sa.pd, 0, e32, k, j - rl.send_off[k], sa.zb[e], sa.zb[(size_t)sa.chunk + e]);
            }
        }
        {   double loc[4];
            reduce_partials_all<4>(step_partials, sa.n_slice_partials, loc, sm);                                               // this rank's slice
            if (lid == 0 && threadIdx.x < 4) p2p_put_double_all(sa.pd, 0, e32, threadIdx.x, threadIdx.x == 0 ? loc[0] : threadIdx.x == 1 ? loc[1] : threadIdx.x == 2 ? loc[2] : loc[3]); }
        double tl[4];
        reduce_partials_all<4>(step_partials + 4 * (size_t)sa.n_slice_partials, n_step - sa.n_slice_partials, tl, sm);         // the camera tail (replicated)
        p2p_sum_all<4>(sa.pd, 0, e32, tot, smx);
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] += tl[k];
    }
    int it = prev->it, done = 0; bool stop = false;
    double Q0 = prev->Q0, Q1 = prev->Q1, xbr = prev->xbr, xr = prev->xr, d2xx = prev->d2xx;
    if (!init) {                                                     // end of iteration it: quadratic-model termination (eta = 0.1)
        xbr = tot[1]; xr = tot[2]; d2xx = tot[3];
        it = it + 1;
        Q1 = -tot[1];
        if (prev->fixed_iterations >= 0) { Q0 = Q1; if (it >= prev->fixed_iterations) { done = 1; stop = true; } }
        else {
            const double zeta = (double)it * (Q1 - Q0) / Q1;
            if (zeta < 0.1) { done = 1; stop = true; }
            else { Q0 = Q1; if (it >= prev->max_iterations) { done = 1; stop = true; } }
        }
    }
    double rho = prev->rho, last_rho = prev->last_rho, beta = prev->beta;
    if (!stop) {                                                     // start of the next iteration: rho = r.z, beta
        const double rho_new = tot[0];
        if (rho_new == 0.0 || isinf(rho_new) || isnan(rho_new)) { done = 2; stop = true; }
        else {
            if (it > 0) { const double bt = rho_new / prev->rho; if (bt == 0.0 || isinf(bt) || isnan(bt)) { done = 2; stop = true; } else beta = bt; }
            else beta = 0.0;
            if (!stop) { last_rho = prev->rho; rho = rho_new; }
        }
    }
    if (writer) {
        PcgState s = *prev;
        s.rho = rho; s.last_rho = last_rho; s.beta = beta; s.xbr = xbr; s.xr = xr; s.d2xx = d2xx; s.Q0 = Q0; s.Q1 = Q1; s.it = it; s.done = done; s.pq = 0.0;
        *next = s;
        publish(done);
    }
    if (stop) return;                                                // (rim words already sent stay unread: their epoch is never asked for again)
    const float betaf = (float)beta; const bool first = it == 0;
    if (rim_wg) {                                                    // p = z + beta p, u = S p on the foreign entries this rank's rows read
        const RimLists& rl = sa.rim; const unsigned e32 = p2p_pass_epoch(seq, P2P_X_DIR);
        for (int j = (lid - n_main - 1) * PF_THREADS + threadIdx.x; j < rl.n_recv; j += n_rim * PF_THREADS) {
            const int e = rl.recv_idx[j], k = rl.recv_peer[j];
            float zs, za; p2p_get_rim(sa.pd, 0, e32, k, j - rl.recv_off[k], zs, za);
            const size_t ea = (size_t)sa.chunk + e;
            const float ps = first ? zs : zs + betaf * sa.pb[e], pa = first ? za : za + betaf * sa.pb[ea];
            sa.pb[e] = ps; sa.pb[ea] = pa;
            sa.ub[e] = lm_scale(sa.cmb[e]) * ps; sa.ub[ea] = lm_scale(sa.cmb[ea]) * pa;
        }
        return;
    }
    double d2 = 0.0;
    if (!SH || !tail_wg) {
        for (int j = lid * PF_THREADS + threadIdx.x; j < 2 * n4; j += n_main * PF_THREADS) {
            const int i = j < n4 ? j : j - n4 + seg4;
            float4 pi = z4[i]; const float4 cv = C4[i];
            if (!first) { const float4 po = p4[i]; pi.x += betaf * po.x; pi.y += betaf * po.y; pi.z += betaf * po.z; pi.w += betaf * po.w; }
            p4[i] = pi;
            float4 sv, dd; float mi;
            lm_from_colnorm(cv.x, inv_radius, sv.x, dd.x, mi); lm_from_colnorm(cv.y, inv_radius, sv.y, dd.y, mi);
            lm_from_colnorm(cv.z, inv_radius, sv.z, dd.z, mi); lm_from_colnorm(cv.w, inv_radius, sv.w, dd.w, mi);
            u4[i] = make_float4(sv.x * pi.x, sv.y * pi.y, sv.z * pi.z, sv.w * pi.w);
            d2 += (double)dd.x * (double)pi.x * (double)pi.x + (double)dd.y * (double)pi.y * (double)pi.y + (double)dd.z * (double)pi.z * (double)pi.z + (double)dd.w * (double)pi.w * (double)pi.w;
        }
    }
    if (tail_wg) {                                                   // the camera tail (6K+9 unknowns): with the last slice workgroup, or (SH) a workgroup of its own
        for (int t = threadIdx.x; t < ntail; t += PF_THREADS) {
            const size_t i = tail_rel + t;
            const float pi = first ? z[i] : z[i] + betaf * p[i]; p[i] = pi; u[i] = S[i] * pi;
            d2 += (double)tD2[t] * (double)pi * (double)pi;
        }
    }
    { const double t = block_sum_d(d2); if (threadIdx.x == 0) d2_partials[lid] = t; }
}
template <int SH>
__global__ void __launch_bounds__(PF_THREADS, SH ? 4 : 1) k_pcg_dir3(int init, int n4, int seg4, size_t tail_rel, int ntail, const float* __restrict__ z, float* __restrict__ p,
                                                        const float* __restrict__ S, float* __restrict__ u, const float* __restrict__ D2 /* S, D2: the camera tail */,
                                                        const float* __restrict__ cm, const LmState* __restrict__ lm, const double* __restrict__ step_partials, int n_step, double* __restrict__ d2_partials,
                                                        const PcgState* __restrict__ prev, PcgState* __restrict__ next, int* host_flags, int seq, ShardArgs sa) {
    pcg_dir3_body<SH>(init, n4, seg4, tail_rel, ntail, z, p, S, u, D2 + tail_rel, cm, lm->inv_radius /* (uniform: a scalar load) */, step_partials, n_step, d2_partials, prev, next, host_flags, seq, sa);
}
// ladder batch: blockIdx.y picks the system; its vectors, partial sums, scalar states and host ring lie at fixed strides behind system 0's.  COLL: a sharded run
// (SH = 2 of the body): the last workgroup is the camera tail's, the slice sums of system j come from red4_0 + 4 j.
template <bool COLL>
__global__ void __launch_bounds__(PF_THREADS, 1) k_pcg_dir3_lad(int init, int n4, int seg4, size_t tail_rel, int ntail, const float* __restrict__ z0, float* __restrict__ p0,
                                                        const float* __restrict__ S, float* __restrict__ u0, const float* __restrict__ tD2_0, const float* __restrict__ cm, const LmState* __restrict__ lm,
                                                        const double* __restrict__ step_partials0, int n_step, double* __restrict__ d2_partials0, PcgState* __restrict__ st2_0 /* [system][2] */, int prev_parity,
                                                        int* host_flags0, int seq, LadVec lv, int n_slice_partials, const double* __restrict__ red4_0) {
    const int j = lv.sysid[blockIdx.y];
    ShardArgs none; none.n_rim_wg = 0; none.n_slice_partials = n_slice_partials;
    const size_t vo = (size_t)j * lv.vec;
    pcg_dir3_body<COLL ? 2 : 0>(init, n4, seg4, tail_rel, ntail, z0 + vo, p0 + vo, S, u0 + vo, tD2_0 + (size_t)j * lv.tail, cm, lm->lad_inv_radius[j], step_partials0 + (size_t)j * lv.part, n_step,
                                d2_partials0 + (size_t)j * lv.part, st2_0 + 2 * j + prev_parity, st2_0 + 2 * j + (prev_parity ^ 1), host_flags0 + 4 * j, seq, none, COLL ? red4_0 + 4 * j : nullptr);
}

static int dir3_main_wgs(int n_entries, int cap) { int b = (n_entries / 2 + PF_THREADS - 1) / PF_THREADS; b = b < 1 ? 1 : b; return b > cap ? cap : b; }      // 2 * (n / 4) float4 items
// returns the number of D^2 p^2 partials of the SLICE (sharded: the tail's own partial follows at that index)
int launch_pcg_dir3(hipStream_t st, bool init, Seg2 sg, size_t tail_off, int ntail, const float* z, float* p, const float* S, float* u, const float* D2, const float* cm, const LmState* lm,
                    const double* step_partials, int n_step, double* d2_partials, const PcgState* prev, PcgState* next, int* host_flags, int seq, const ShardArgs* sa) {
    const int n4 = sg.n >> 2, seg4 = (int)((sg.off1 - sg.off0) >> 2);
    const size_t o = sg.off0;
    if (!sa) {
        const int blocks = dir3_main_wgs(sg.n, PF_MAX_WG);
        ShardArgs none; std::memset(&none, 0, sizeof(none));
        k_pcg_dir3<0><<<blocks, PF_THREADS, 0, st>>>(init ? 1 : 0, n4, seg4, tail_off - o, ntail, z + o, p + o, S + o, u + o, D2 + o, cm + o, lm, step_partials, n_step, d2_partials, prev, next, host_flags, seq, none);
        return blocks;
    }
    const int cap = sa->pd.wg_cap > 0 ? sa->pd.wg_cap : PF_MAX_WG;
    const int n_main = dir3_main_wgs(sg.n, cap);
    k_pcg_dir3<1><<<sa->n_rim_wg + n_main + 1, PF_THREADS, 0, st>>>(init ? 1 : 0, n4, seg4, tail_off - o, ntail, z + o, p + o, S + o, u + o, D2 + o, cm + o, lm, step_partials, n_step, d2_partials, prev, next,
                                                                       host_flags, seq, *sa);
    return n_main;
}

// all systems of `lv` in one launch; returns the number of D^2 p^2 partials per system
int launch_pcg_dir3_lad(hipStream_t st, bool init, int nsys, Seg2 sg, size_t tail_off, int ntail, const float* z0, float* p0, const float* S, float* u0, const float* tD2_0, const float* cm, const LmState* lm,
                        const double* step_partials0, int n_step, double* d2_partials0, PcgState* st2_0, int prev_parity, int* host_flags0, int seq, const LadVec& lv,
                        int n_slice_partials, const double* red4_0) {
    const int n4 = sg.n >> 2, seg4 = (int)((sg.off1 - sg.off0) >> 2);
    const size_t o = sg.off0;
    const int blocks = dir3_main_wgs(sg.n, PF_MAX_WG);
    if (red4_0)         // sharded: the camera tail in a workgroup of its own (its D^2 p^2 partial follows the slice's at index `blocks`)
        k_pcg_dir3_lad<true><<<dim3(blocks + 1, nsys), PF_THREADS, 0, st>>>(init ? 1 : 0, n4, seg4, tail_off - o, ntail, z0 + o, p0 + o, S + o, u0 + o, tD2_0, cm + o, lm, step_partials0, n_step, d2_partials0,
                                                                            st2_0, prev_parity, host_flags0, seq, lv, n_slice_partials, red4_0);
    else
        k_pcg_dir3_lad<false><<<dim3(blocks, nsys), PF_THREADS, 0, st>>>(init ? 1 : 0, n4, seg4, tail_off - o, ntail, z0 + o, p0 + o, S + o, u0 + o, tD2_0, cm + o, lm, step_partials0, n_step, d2_partials0,
                                                                         st2_0, prev_parity, host_flags0, seq, lv, 0, nullptr);
    return blocks;
}

// residual-reset passes of a sharded run: the rim of the operator input u = S x (k_pcg_dir3's rim workgroups move z, not u).  Once per ten passes.
__global__ void __launch_bounds__(PF_THREADS) k_rim_u(ShardArgs sa, const PcgState* __restrict__ state) {
    if (state->done) return;
    const RimLists& rl = sa.rim; const unsigned e32 = p2p_pass_epoch(sa.seq, P2P_X_RESET_RIM);
    for (int j = blockIdx.x * PF_THREADS + threadIdx.x; j < rl.n_send; j += gridDim.x * PF_THREADS) {
        const int e = rl.send_idx[j], k = rl.send_peer[j];
        p2p_put_rim(sa.pd, 0, e32, k, j - rl.send_off[k], sa.ub[e], sa.ub[(size_t)sa.chunk + e]);
    }
    for (int j = blockIdx.x * PF_THREADS + threadIdx.x; j < rl.n_recv; j += gridDim.x * PF_THREADS) {
        const int e = rl.recv_idx[j], k = rl.recv_peer[j];
        float us, ua; p2p_get_rim(sa.pd, 0, e32, k, j - rl.recv_off[k], us, ua);
        sa.ub[e] = us; sa.ub[(size_t)sa.chunk + e] = ua;
    }
}
void launch_rim_u(hipStream_t st, const ShardArgs& sa, const PcgState* state) { if (sa.rim.n_send > 0 || sa.rim.n_recv > 0) k_rim_u<<<sa.n_rim_wg > 0 ? sa.n_rim_wg : 1, PF_THREADS, 0, st>>>(sa, state); }

enum { S3_INIT = 0, S3_NORMAL = 1, S3_XONLY = 2, S3_RESET = 3 };


// what one thread of k_pcg_step3 reads for its 4 entries (sdf + albedo segment): every load of a batch is issued before the first use.
// (Requesting the first batch BEFORE the prologue's reductions was tried: the registers it holds across them cost an occupancy step — 144 VGPRs —
// and the kernel got slower, 91 against 80 ms of vector kernels per 10 iterations; profiles/r03_ab_variants.json.)
// where system j of a ladder batch keeps its copy of the per-system arrays, relative to system 0's (all zero for the single-system launch).  Applied at the use
// sites: a modified COPY of the argument struct went to scratch memory (976 bytes per lane, every pointer a scratch load: the vector kernels of a batch ran at half
// the speed of the serial ones, profiles/r05_ladder_vector_scratch.json).
struct S3Off { size_t v4, vo, qh2, part, cam, mblk, tail; int st, sys; };
struct S3In { float4 as, aa; int4 o; int o4; float4 x[2], p[2], b[2], cm[2], r[2]; };
template <int MODE> static __device__ inline void s3_load(const Step3Args& a, const S3Off& o, int q, S3In& v) {
    const float4* const qacc = a.qacc + o.v4; const float4* const x = a.x + o.v4; const float4* const p = a.p + o.v4; const float4* const r = a.r + o.v4;
    if (MODE == S3_NORMAL || MODE == S3_RESET) {
        v.as = qacc[q]; v.aa = qacc[q + a.chunk4];
        const int e = a.e0 + 4 * q;
        v.o = *reinterpret_cast<const int4*>(a.ext_off + e);                 // (e is a multiple of 4: slices start at multiples of 1024)
        v.o4 = a.ext_off[e + 4];
    }
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const int i = q + seg * a.chunk4;
        v.cm[seg] = a.cm[i];
        if (MODE == S3_INIT) v.r[seg] = r[i];
        else {
            v.x[seg] = x[i];
            if (MODE != S3_RESET) v.p[seg] = p[i];
            if (MODE != S3_XONLY) { v.b[seg] = a.b[i]; if (MODE == S3_NORMAL) v.r[seg] = r[i]; }
        }
    }
}

// SH (sharded run over the mailboxes): the second exchange of a pass happens in here — this rank's p.q (rows it owns + D^2 p^2 of its slice) is summed over the
// ranks in the prologue of EVERY workgroup (workgroup 0 stores it into all mailboxes), and the camera workgroups sum their columns of the operator's camera
// block over the ranks before they update the (replicated) camera tail.  No launch of its own, no vector leaves a rank.
// SH = 2 (round 6): the sharded run whose exchanges are launches of the transport — [camera block | p.q] of this system arrive summed over the ranks in a.redop
// (k_lad_reduce_op + one all-reduce in front of this kernel): p.q of the slices at [6K + 9], the camera tail's D^2 p^2 (replicated, counted once) added locally.
template <int MODE, int SH>
static __device__ __forceinline__ void pcg_step3_body(const Step3Args& a, const S3Off& o) {
    __shared__ double sm[4 * 8];
    __shared__ double smx[SH == 1 ? P2P_MAX_RANKS : 1];
    __shared__ double camv[64];
    __shared__ double camred[8][64];
    __shared__ float rs[64];
    PcgState* const cur = a.cur + o.st;
    float4* const X = a.x + o.v4; float4* const R = a.r + o.v4; float4* const Z = a.z + o.v4;
    const double* const pq_partials = a.pq_partials + o.part; const double* const d2_partials = a.d2_partials + o.part; double* const step_partials = a.step_partials + o.part;
    const float2* const qh = a.qh + o.qh2; const float* const cam_partials = a.cam_partials + o.cam; const float* const Mblk = a.Mblk + o.mblk;
    const float* const tp = a.tp + o.vo; float* const tx = a.tx + o.vo; float* const tr = a.tr + o.vo; float* const tz = a.tz + o.vo; const float* const tD2 = a.tD2 + o.tail;
    const bool slice_wg = (int)blockIdx.x < a.n_slice_wg;
    const int q0 = blockIdx.x * PF_THREADS + threadIdx.x, qstride = a.n_slice_wg * PF_THREADS;
    S3In in;
    // ONE read of the flag per workgroup: workgroup 0 may store done = 2 (breakdown) below while later workgroups of the same launch start, and waves of
    // one workgroup that saw different values would leave the prologue reduction with unwritten slots.  Whatever a workgroup reads, it acts on as a whole:
    // done != 0 -> nothing is touched; done == 0 -> it finds the same breakdown itself (every workgroup adds the same partials in the same order) and returns.
    __shared__ int done_s;
    if (threadIdx.x == 0) done_s = cur->done;
    __syncthreads();
    if (done_s) return;
    const float inv_radius = o.sys >= 0 ? a.lm->lad_inv_radius[o.sys] : a.lm->inv_radius;      // (uniform: a scalar load)
    float alpha = 0.0f;
    if (MODE == S3_NORMAL || MODE == S3_XONLY) {                     // p.q = sum over rows of t (J u) + sum D^2 p^2  ->  alpha = rho / p.q
        double t1[1];
        if (SH == 2) t1[0] = a.redop[(size_t)(o.sys < 0 ? 0 : o.sys) * a.redop_stride + 6 * a.K + 9] + d2_partials[a.n_d2];
        else reduce_partials_all<1>(pq_partials, a.n_pq, t1, sm, d2_partials, a.n_d2);      // both lists in ONE reduction (one barrier pair instead of two)
        if (SH == 1) {
            const unsigned e32 = p2p_pass_epoch(a.sh.seq, P2P_X_STEP);
            if (blockIdx.x == 0 && threadIdx.x == 0) p2p_put_double_all(a.sh.pd, 1, e32, 0, t1[0]);
            p2p_sum_all<1>(a.sh.pd, 1, e32, t1, smx);
            t1[0] += d2_partials[a.n_d2];                                                // the camera tail's D^2 p^2: replicated, counted once
        }
        const double pq = t1[0];
        const double al = cur->rho / pq;
        const bool bad = !(pq > 0.0) || isinf(pq) || isinf(al);
        if (blockIdx.x == 0 && threadIdx.x == 0) { cur->pq = pq; if (bad) cur->done = 2; else cur->alpha = al; }      // (no kernel reads pq / alpha from the state; done = 2 only makes everyone return)
        if (bad) return;
        alpha = (float)al;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (slice_wg) {
        for (int q = q0; q < a.nq; q += qstride) {
            s3_load<MODE>(a, o, q, in);
            float accv[2][4];
            if (MODE == S3_NORMAL || MODE == S3_RESET) {
                const float4 as = in.as, aa = in.aa;
                accv[0][0] = as.x; accv[0][1] = as.y; accv[0][2] = as.z; accv[0][3] = as.w; accv[1][0] = aa.x; accv[1][1] = aa.y; accv[1][2] = aa.z; accv[1][3] = aa.w;
                // halo fold: the (entry, halo slot) pairs of the 4 entries are ONE contiguous run [o.x, o4) of the sorted pair list.  Batches of 8: all slot
                // indices are requested together, then all halo sums — two memory round trips per batch instead of two per PAIR (the per-pair loop was a
                // chain of ~9 dependent loads per thread; this kernel is latency-bound, not bandwidth-bound).  Sums are added in ascending pair order, as before.
                const int ob[5] = {in.o.x, in.o.y, in.o.z, in.o.w, in.o4};
                for (int j0 = ob[0]; j0 < ob[4]; j0 += 8) {
                    int pos[8]; float2 hv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) pos[u] = (j0 + u < ob[4]) ? a.ext_pos[j0 + u] : -1;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = pos[u] >= 0 ? qh[pos[u]] : make_float2(0.0f, 0.0f);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = j0 + u;
                        if (j < ob[4]) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) { const bool mine = j >= ob[k] && j < ob[k + 1]; accv[0][k] += mine ? hv[u].x : 0.0f; accv[1][k] += mine ? hv[u].y : 0.0f; }
                        }
                    }
                }
            }
#pragma unroll
            for (int seg = 0; seg < 2; ++seg) {
                const int i = q + seg * a.chunk4;
                float xv[4], rv[4], svv[4], dv[4], mv[4];
                { const float4 cv = in.cm[seg]; lm_from_colnorm(cv.x, inv_radius, svv[0], dv[0], mv[0]); lm_from_colnorm(cv.y, inv_radius, svv[1], dv[1], mv[1]);
                  lm_from_colnorm(cv.z, inv_radius, svv[2], dv[2], mv[2]); lm_from_colnorm(cv.w, inv_radius, svv[3], dv[3], mv[3]); }
                if (MODE == S3_INIT) { const float4 t = in.r[seg]; rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w; }
                else {
                    const float4 xo = in.x[seg];
                    xv[0] = xo.x; xv[1] = xo.y; xv[2] = xo.z; xv[3] = xo.w;
                    float pv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (MODE != S3_RESET) {
                        const float4 pp = in.p[seg]; pv[0] = pp.x; pv[1] = pp.y; pv[2] = pp.z; pv[3] = pp.w;
#pragma unroll
                        for (int k = 0; k < 4; ++k) xv[k] += alpha * pv[k];
                        X[i] = make_float4(xv[0], xv[1], xv[2], xv[3]);
                    }
                    if (MODE == S3_XONLY) continue;
                    const float4 bb = in.b[seg];
                    const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
                    float qq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) qq[k] = svv[k] * accv[seg][k] + dv[k] * (MODE == S3_NORMAL ? pv[k] : xv[k]);      // q = S acc + D^2 v
                    if (MODE == S3_NORMAL) { const float4 ro = in.r[seg]; rv[0] = ro.x - alpha * qq[0]; rv[1] = ro.y - alpha * qq[1]; rv[2] = ro.z - alpha * qq[2]; rv[3] = ro.w - alpha * qq[3]; }
                    else { rv[0] = bv[0] - qq[0]; rv[1] = bv[1] - qq[1]; rv[2] = bv[2] - qq[2]; rv[3] = bv[3] - qq[3]; }           // RESET: r = b - A x
                    R[i] = make_float4(rv[0], rv[1], rv[2], rv[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double xd = xv[k]; s1 += xd * ((double)bv[k] + (double)rv[k]); s2 += xd * (double)rv[k]; s3 += (double)dv[k] * xd * xd; }
                }
                const float zv[4] = {mv[0] * rv[0], mv[1] * rv[1], mv[2] * rv[2], mv[3] * rv[3]};
                Z[i] = make_float4(zv[0], zv[1], zv[2], zv[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) s0 += (double)rv[k] * (double)zv[k];
            }
        }
    } else {
        // ---- camera tail: PF_POSES_PER_WG poses per workgroup, the last tail workgroup takes intrinsics (4) + distortion (5) ----
        const int K = a.K, tw = (int)blockIdx.x - a.n_slice_wg, n_pose_wg = (K + PF_POSES_PER_WG - 1) / PF_POSES_PER_WG;
        int col0, ncol;
        if (tw < n_pose_wg) { col0 = 6 * PF_POSES_PER_WG * tw; const int c1 = min(6 * K, col0 + 6 * PF_POSES_PER_WG); ncol = c1 - col0; }
        else { col0 = 6 * K; ncol = 9; }
        const int t = threadIdx.x;
        if (MODE == S3_NORMAL || MODE == S3_RESET) {                 // column sums of the operator's camera partials (one float row per workgroup), fixed order
            const int g = t / 64, c = t % 64;                        // 8 row groups x 64 columns
            double v = 0.0;
            if (SH != 2 && c < ncol) for (int w = g; w < a.n_cam; w += 8) v += (double)cam_partials[(size_t)w * a.cam_stride + col0 + c];
            camred[g][c] = v;
            __syncthreads();
            if (t < ncol) {
                double s = 0.0; for (int gg = 0; gg < 8; ++gg) s += camred[gg][t];
                if (SH == 2) s = a.redop[(size_t)(o.sys < 0 ? 0 : o.sys) * a.redop_stride + col0 + t];      // summed over the workgroups (k_lad_reduce_op) and the ranks (all-reduce)
                if (SH == 1) {       // this rank's column sums (rows it owns) -> all ranks; the total in rank order
                    const unsigned e32 = p2p_pass_epoch(a.sh.seq, MODE == S3_RESET ? P2P_X_RESET_STEP : P2P_X_STEP);
                    p2p_put_double_all(a.sh.pd, 1, e32, 1 + col0 + t, s);
                    s = 0.0; for (int j = 0; j < a.sh.pd.L.world; ++j) s += p2p_get_double(a.sh.pd, 1, e32, j, 1 + col0 + t);
                }
                camv[t] = s;
            }
            __syncthreads();
        }
        if (t < ncol) {
            const int i = col0 + t;
            const bool fixed = i < 6 * K ? a.fix_poses : (i < 6 * K + 4 ? a.fix_intr : a.fix_dist);
            float ri;
            if (MODE == S3_INIT) ri = tr[i];
            else {
                float xi = tx[i];
                if (MODE != S3_RESET) { xi += alpha * tp[i]; tx[i] = xi; }
                if (MODE == S3_XONLY) ri = 0.0f;
                else {
                    const float vv = MODE == S3_NORMAL ? tp[i] : xi;
                    const float qi = a.tS[i] * (fixed ? 0.0f : (float)camv[t]) + tD2[i] * vv;
                    ri = MODE == S3_NORMAL ? tr[i] - alpha * qi : a.tb[i] - qi;
                    tr[i] = ri;
                    const double xd = xi; s1 += xd * ((double)a.tb[i] + (double)ri); s2 += xd * (double)ri; s3 += (double)tD2[i] * xd * xd;
                }
            }
            rs[t] = ri;
        }
        if (MODE == S3_XONLY) return;
        __syncthreads();                                             // the block preconditioner mixes the entries of a parameter block
        if (t < ncol) {
            const int i = col0 + t;
            int base, n, row; const float* M;
            if (i < 6 * K) { const int f = i / 6; base = 6 * f; n = 6; row = i - base; M = Mblk + 36 * f; }
            else if (i < 6 * K + 4) { base = 6 * K; n = 4; row = i - base; M = Mblk + 36 * K; }
            else { base = 6 * K + 4; n = 5; row = i - base; M = Mblk + 36 * K + 16; }
            float s = 0.0f;
            for (int j = 0; j < n; ++j) s += M[row * n + j] * rs[base - col0 + j];
            tz[i] = s; s0 += (double)rs[t] * (double)s;
        }
    }
    if (MODE == S3_XONLY) return;
    block_partial_d(s0, step_partials, 4, 0); block_partial_d(s1, step_partials, 4, 1); block_partial_d(s2, step_partials, 4, 2); block_partial_d(s3, step_partials, 4, 3);
}
template <int MODE, int SH>
__global__ void __launch_bounds__(PF_THREADS) k_pcg_step3(Step3Args a) { const S3Off o{0, 0, 0, 0, 0, 0, 0, 0, -1}; pcg_step3_body<MODE, SH>(a, o); }
// ladder batch (single rank): blockIdx.y picks the system (see k_pcg_dir3_lad); `a` holds system 0's pointers
template <int MODE, bool COLL>
__global__ void __launch_bounds__(PF_THREADS) k_pcg_step3_lad(Step3Args a, LadVec lv) {
    const int j = lv.sysid[blockIdx.y];
    const S3Off o{((size_t)j * lv.vec) >> 2, (size_t)j * lv.vec, ((size_t)j * lv.qh) >> 1, (size_t)j * lv.part, (size_t)j * lv.cam, (size_t)j * lv.mblk, (size_t)j * lv.tail, 2 * j, j};
    pcg_step3_body<MODE, COLL ? 2 : 0>(a, o);
}

int pcg_step3_slice_wgs(int n_entries, int cap) { if (cap <= 0 || cap > PF_MAX_WG) cap = PF_MAX_WG; int b = (n_entries / 4 + PF_THREADS - 1) / PF_THREADS; return b < 1 ? 1 : (b > cap ? cap : b); }
int pcg_step3_tail_wgs(int K) { return (K + PF_POSES_PER_WG - 1) / PF_POSES_PER_WG + 1; }

// returns the number of [4]-partials written (0 in XONLY mode)
int launch_pcg_step3(hipStream_t st, int mode, Step3Args a) {
    const int blocks = a.n_slice_wg + pcg_step3_tail_wgs(a.K);
#define I3D_S3(M) do { if (a.sharded) k_pcg_step3<M, 1><<<blocks, PF_THREADS, 0, st>>>(a); else k_pcg_step3<M, 0><<<blocks, PF_THREADS, 0, st>>>(a); } while (0)
    switch (mode) {
        case S3_INIT:   I3D_S3(S3_INIT); break;
        case S3_NORMAL: I3D_S3(S3_NORMAL); break;
        case S3_XONLY:  I3D_S3(S3_XONLY); break;
        default:        I3D_S3(S3_RESET); break;
    }
#undef I3D_S3
    return mode == S3_XONLY ? 0 : blocks;
}

// all systems of `lv` in one launch (a.cur = system 0's state of this pass's parity); returns the number of [4]-partials per system (0 in XONLY mode)
int launch_pcg_step3_lad(hipStream_t st, int mode, int nsys, Step3Args a, const LadVec& lv) {
    const int blocks = a.n_slice_wg + pcg_step3_tail_wgs(a.K);
    const dim3 grid(blocks, nsys);
#define I3D_S3L(M) do { if (a.redop) k_pcg_step3_lad<M, true><<<grid, PF_THREADS, 0, st>>>(a, lv); else k_pcg_step3_lad<M, false><<<grid, PF_THREADS, 0, st>>>(a, lv); } while (0)
    switch (mode) {
        case S3_INIT:   I3D_S3L(S3_INIT); break;
        case S3_NORMAL: I3D_S3L(S3_NORMAL); break;
        case S3_XONLY:  I3D_S3L(S3_XONLY); break;
        default:        I3D_S3L(S3_RESET); break;
    }
#undef I3D_S3L
    return mode == S3_XONLY ? 0 : blocks;
}

// Sharded ladder pass (round 6), the exchanges as launches of the transport: what a rank contributes to the two all-reduces of a pass, for every live system at once.
//   k_lad_reduce_step : red4[j][0..3]   = this rank's four slice sums of k_pcg_step3 (its first n_slice partial rows; the camera tail's rows stay local: replicated)
//   k_lad_reduce_op   : redop[j][0..NS) = column sums of this rank's camera partial rows of the operator pass (the rows it owns), redop[j][NS] = its p.q (rows it owns + D^2 p^2 of its slice)
// Fixed summation orders (the shapes of reduce_partials_all / of k_pcg_step3's camera workgroups): a rank's contribution is bit-reproducible; the all-reduce adds the ranks'.
__global__ void __launch_bounds__(PF_THREADS) k_lad_reduce_step(const double* __restrict__ step_partials0, int n_slice, double* __restrict__ red4_0, LadVec lv) {
    __shared__ double sm[4 * 8];
    const int j = lv.sysid[blockIdx.x];
    double tot[4];
    reduce_partials_all<4>(step_partials0 + (size_t)j * lv.part, n_slice, tot, sm);
    if (threadIdx.x < 4) red4_0[4 * j + threadIdx.x] = threadIdx.x == 0 ? tot[0] : threadIdx.x == 1 ? tot[1] : threadIdx.x == 2 ? tot[2] : tot[3];
}
void launch_lad_reduce_step(hipStream_t st, int nsys, const double* step_partials0, int n_slice, double* red4_0, const LadVec& lv) {
    k_lad_reduce_step<<<nsys, PF_THREADS, 0, st>>>(step_partials0, n_slice, red4_0, lv);
}
__global__ void __launch_bounds__(PF_THREADS) k_lad_reduce_op(const float* __restrict__ cam_partials0, int n_cam, int cam_stride, int NS, const double* __restrict__ pq_partials0, int n_pq,
                                                              const double* __restrict__ d2_partials0, int n_d2, double* __restrict__ redop0, int redop_stride, LadVec lv) {
    __shared__ double sm[8];
    __shared__ double camred[8][64];
    const int j = lv.sysid[blockIdx.y];
    double* const out = redop0 + (size_t)j * redop_stride;
    const int ncw = (NS + 63) / 64;
    if ((int)blockIdx.x < ncw) {                                     // 64 columns of the camera block: 8 row groups, then the groups in order
        const int t = threadIdx.x, g = t / 64, c = t % 64, col = (int)blockIdx.x * 64 + c;
        const float* cp = cam_partials0 + (size_t)j * lv.cam;
        double v = 0.0;
        if (col < NS) for (int w = g; w < n_cam; w += 8) v += (double)cp[(size_t)w * cam_stride + col];
        camred[g][c] = v;
        __syncthreads();
        if (t < 64 && col < NS) { double s = 0.0; for (int gg = 0; gg < 8; ++gg) s += camred[gg][t]; out[col] = s; }
    } else {
        double t1[1];
        reduce_partials_all<1>(pq_partials0 + (size_t)j * lv.part, n_pq, t1, sm, d2_partials0 + (size_t)j * lv.part, n_d2);
        if (threadIdx.x == 0) out[NS] = t1[0];
    }
}
void launch_lad_reduce_op(hipStream_t st, int nsys, const float* cam_partials0, int n_cam, int cam_stride, int NS, const double* pq_partials0, int n_pq, const double* d2_partials0, int n_d2,
                          double* redop0, int redop_stride, const LadVec& lv) {
    k_lad_reduce_op<<<dim3((NS + 63) / 64 + 1, nsys), PF_THREADS, 0, st>>>(cam_partials0, n_cam, cam_stride, NS, pq_partials0, n_pq, d2_partials0, n_d2, redop0, redop_stride, lv);
}

// ladder batch: both states of every system; a system whose radius has run out (k_lm_begin_lad) starts finished
__global__ void k_pcg_init_lad(PcgState* st2_all, int B, int fixed_iterations, int max_iterations, const LmState* lm) {
    const int j = threadIdx.x;
    if (j >= B) return;
    const bool over = lm->done != 0 || lm->lad_radius[j] < 1e-32;
    for (int b = 0; b < 2; ++b) {
        PcgState* st = st2_all + 2 * j + b;
        for (int k = 0; k < 4; ++k) st->acc[k] = 0.0;
        st->rho = 0.0; st->last_rho = 1.0; st->pq = 0.0; st->alpha = 0.0; st->beta = 0.0; st->xbr = 0.0; st->xr = 0.0; st->d2xx = 0.0;
        st->Q0 = 0.0; st->Q1 = 0.0; st->it = 0; st->done = (fixed_iterations == 0 || over) ? 1 : 0; st->fixed_iterations = fixed_iterations; st->max_iterations = max_iterations;
    }
}
void launch_pcg_init_lad(hipStream_t st, PcgState* st2_all, int B, int fixed_iterations, int max_iterations, const LmState* lm) { k_pcg_init_lad<<<1, LADDER_MAX, 0, st>>>(st2_all, B, fixed_iterations, max_iterations, lm); }

// plan side of the fold: ext_off[e] = index of the first (entry, halo slot) pair whose entry is >= e, e in [0, A]: one lower-bound search per entry over the
// sorted keys (the keys of a plan that overflowed are not entries: they compare like any other integer and nothing is written outside [0, A])
__global__ void k_ext_offsets(int n, const int* __restrict__ ext_e, int A, int* __restrict__ ext_off) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > A) return;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ext_e[mid] < e) lo = mid + 1; else hi = mid; }
    ext_off[e] = lo;
}
void launch_ext_offsets(hipStream_t st, int n, const int* ext_e, int A, int* ext_off) { k_ext_offsets<<<(A + 1 + 255) / 256, 256, 0, st>>>(n, ext_e, A, ext_off); }

}  // namespace i3d
