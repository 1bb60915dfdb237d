// K7 — the trust-region loop of NLSSolver::solve (nls_solver.cpp:296-337: Ceres 2.1.0 TrustRegionMinimizer + LevenbergMarquardtStrategy, un-vendored;
// semantics per SURVEY.md Appendix B.2) WITHOUT the host in it.  Round 3 synchronised the stream twice per LM attempt (terminal PCG state + candidate
// camera back to the host, rotation matrices of the candidate poses built there and uploaded, candidate cost back, rho test on the host): ~14 drained
// pipelines per Gauss-Newton iteration, 10-15 % of its wall clock.  Now:
//   k_lm_init     cost / gradient / free-parameter tests at the start (TrustRegionMinimizer::Init), state -> LmState
//   k_lm_begin    per attempt: radius underflow test, 1/radius for the vector kernels, LM diagonal of the camera tail, block-Jacobi inverses of the damped
//                 pose (6x6) / intrinsics (4x4) / distortion (5x5) blocks — Cholesky in fp64, one thread per block (was: K Cholesky inversions on the host)
//   k_cand_frames per attempt: per-keyframe constants of the CANDIDATE poses (device/frame_math.hpp: the formulas of the host's build_frame_consts)
//   k_lm_decide   per attempt: step validity, parameter / function tolerance, rho test, radius update, accept flag; one LmRecord to mapped host memory
// Every kernel of an attempt starts with `if (lm->done) return`, so the host may queue the next attempt before it knows how the last one ended.
#include "kernels.hpp"
#include "frame_math.hpp"

namespace i3d {

static __device__ inline void lm_publish(LmRecord* rec, int seq) {
    if (rec) __hip_atomic_store(&rec->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_lm_init(LmState* lm, const double* __restrict__ cost, const double* __restrict__ ngrad, const double* __restrict__ nfree, double radius0, LmRecord* rec, int seq) {
    LmState s;
    s.cost = *cost; s.radius = radius0; s.decrease_factor = 2.0; s.ngrad = *ngrad; s.nfree = *nfree;
    s.inv_radius = (float)(1.0 / radius0); s.pad0 = 0;
    s.done = 0; s.termination = 0; s.accepted = 0; s.invalid = 0; s.attempts = 0; s.successful = 0;
    // no free parameter, or gradient_tolerance: max-norm of the gradient over the free parameters <= 1e-10 (trust_region_minimizer.cc, Ceres 2.1.0), i.e. no entry above it
    if (s.nfree == 0.0 || s.ngrad == 0.0) { s.done = 1; s.termination = 1; }
    *lm = s;
    if (rec) {
        rec->final_ = s.done; rec->accepted = 0; rec->pcg_it = 0; rec->termination = s.termination; rec->kind = 0;
        rec->cost = s.cost; rec->cand_cost = 0.0; rec->model_change = 0.0; rec->rel = 0.0; rec->radius_after = s.radius; rec->ngrad = s.ngrad; rec->nfree = s.nfree;
        lm_publish(rec, seq);
    }
}
__global__ void k_set_double(double* dst, double v) { *dst = v; }
void launch_set_double(hipStream_t st, double* dst, double v) { k_set_double<<<1, 1, 0, st>>>(dst, v); }
void launch_lm_init(hipStream_t st, LmState* lm, const double* cost, const double* ngrad, const double* nfree, double radius0, LmRecord* rec, int seq) {
    k_lm_init<<<1, 1, 0, st>>>(lm, cost, ngrad, nfree, radius0, rec, seq);
}

// Cholesky inverse of an SPD n x n block, n <= 6 (Ceres: BlockRandomAccessDiagonalMatrix::Invert)
static __device__ inline bool spd_invert_dev(int n, const double* m, double* inv) {
    double L[36];
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = m[i * n + j];
        for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
        if (i == j) { if (!(s > 0.0)) return false; L[i * n + i] = sqrt(s); } else L[i * n + j] = s / L[j * n + j];
    }
    for (int col = 0; col < n; ++col) {
        double y[6], x[6];
        for (int i = 0; i < n; ++i) { double s = (i == col) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
        for (int i = 0; i < n; ++i) inv[i * n + col] = x[i];
    }
    return true;
}
// block-Jacobi inverse of one camera block of  S H S + D^2  (fp64): cdiag = its squared column norms, tri = the upper triangle of H
static __device__ inline void cam_block_inverse(int n, const double* __restrict__ cdiag, const double* __restrict__ tri, bool fixed, double radius, float* __restrict__ out) {
    if (fixed) { for (int i = 0; i < n * n; ++i) out[i] = 0.0f; return; }
    double S[6], M[36], inv[36];
    for (int i = 0; i < n; ++i) S[i] = 1.0 / (1.0 + sqrt(cdiag[i]));
    int o = 0;
    for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { const double v = S[i] * S[j] * tri[o++]; M[i * n + j] = v; M[j * n + i] = v; }
    for (int i = 0; i < n; ++i) { const double cs = cdiag[i] * S[i] * S[i]; M[i * n + i] += fmin(fmax(cs, 1e-6), 1e32) / radius; }
    if (!spd_invert_dev(n, M, inv)) { for (int i = 0; i < n * n; ++i) inv[i] = 0.0; for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0 / M[i * n + i]; }
    for (int i = 0; i < n * n; ++i) out[i] = (float)inv[i];
}

__global__ void __launch_bounds__(64) k_lm_begin(LmState* lm, int K, int fix_poses, int fix_intr, int fix_dist, const double* __restrict__ cdiag /* [6K+9] */, const double* __restrict__ tri /* [21K+25] */,
                                                 float* __restrict__ Mblk /* [36K+41] */, const float* __restrict__ tc, const float* __restrict__ tS, float* __restrict__ tD2, float* __restrict__ tMinv /* camera tail of the vectors */,
                                                 LmRecord* rec, int seq) {
    if (lm->done) return;
    const double radius = lm->radius;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (radius < 1e-32) {           // LevenbergMarquardtStrategy: the radius cannot shrink further -> TrustRegionMinimizer stops (reported as convergence)
        if (gid == 0) {
            lm->termination = 1; lm->done = 1;
            if (rec) { rec->final_ = 1; rec->accepted = 0; rec->pcg_it = 0; rec->termination = 1; rec->kind = 2; rec->cost = lm->cost; rec->cand_cost = 0.0; rec->model_change = 0.0; rec->rel = 0.0;
                       rec->radius_after = radius; rec->ngrad = lm->ngrad; rec->nfree = lm->nfree; lm_publish(rec, seq); }
        }
        return;
    }
    const float inv_radius = (float)(1.0 / radius);
    if (gid == 0) lm->inv_radius = inv_radius;
    if (gid < K) cam_block_inverse(6, cdiag + 6 * gid, tri + 21 * gid, fix_poses != 0, radius, Mblk + 36 * (size_t)gid);
    else if (gid == K) cam_block_inverse(4, cdiag + 6 * K, tri + 21 * K, fix_intr != 0, radius, Mblk + 36 * (size_t)K);
    else if (gid == K + 1) cam_block_inverse(5, cdiag + 6 * K + 4, tri + 21 * K + 10, fix_dist != 0, radius, Mblk + 36 * (size_t)K + 16);
    const int NS = 6 * K + 9;
    for (int i = gid; i < NS; i += gridDim.x * blockDim.x) { float d2, mi; lm_diag(tc[i], tS[i], inv_radius, d2, mi); tD2[i] = d2; tMinv[i] = mi; }
}
void launch_lm_begin(hipStream_t st, LmState* lm, int K, int fix_poses, int fix_intr, int fix_dist, const double* cdiag, const double* tri, float* Mblk,
                     const float* tc, const float* tS, float* tD2, float* tMinv, LmRecord* rec, int seq) {
    k_lm_begin<<<(K + 2 + 63) / 64, 64, 0, st>>>(lm, K, fix_poses, fix_intr, fix_dist, cdiag, tri, Mblk, tc, tS, tD2, tMinv, rec, seq);
}

// The same for a LADDER batch of B consecutive attempts (common.hpp: LADDER_MAX): system j (blockIdx.y) gets the radius the trust region will have after j
// rejections from its current state — radius / decrease_factor, decrease_factor * 2, replayed in the strategy's own arithmetic (k_lm_decide), so system j is solved
// with bit for bit the radius the serial loop would reach — its 1/radius for the vector kernels, its damped block-Jacobi inverses and the LM diagonal of its camera
// tail.  A system whose radius has run out (< 1e-32) gets inv_radius 0 and no work: its PCG starts finished (k_pcg_init_lad), the decision chain ends the solve
// where the serial loop would (k_lm_decide's look-ahead).  Clears the out-of-step state of the previous batch.
__global__ void __launch_bounds__(64) k_lm_begin_lad(LmState* lm, int B, int K, int fix_poses, int fix_intr, int fix_dist, const double* __restrict__ cdiag, const double* __restrict__ tri,
                                                     float* __restrict__ Mblk, size_t mblk_stride, const float* __restrict__ tc, const float* __restrict__ tS, float* __restrict__ tD2, size_t tail_stride,
                                                     LmRecord* rec, int seq) {
    if (lm->done == 1) return;
    const int j = blockIdx.y, gid = blockIdx.x * blockDim.x + threadIdx.x;
    double radius = lm->radius, dec = lm->decrease_factor;
    if (j == 0 && radius < 1e-32) {      // the first attempt of the batch: exactly k_lm_begin's test
        if (gid == 0) {
            lm->termination = 1; lm->done = 1; lm->lad_n = 0;
            if (rec) { rec->final_ = 1; rec->accepted = 0; rec->pcg_it = 0; rec->termination = 1; rec->kind = 2; rec->cost = lm->cost; rec->cand_cost = 0.0; rec->model_change = 0.0; rec->rel = 0.0;
                       rec->radius_after = radius; rec->ngrad = lm->ngrad; rec->nfree = lm->nfree; lm_publish(rec, seq); }
        }
        return;
    }
    for (int s = 0; s < j; ++s) { radius = radius / dec; dec *= 2.0; }
    const bool dead = radius < 1e-32;
    const float inv_radius = dead ? 0.0f : (float)(1.0 / radius);
    if (gid == 0) { lm->lad_radius[j] = radius; lm->lad_inv_radius[j] = inv_radius; if (j == 0) { lm->lad_n = B; lm->inv_radius = inv_radius; lm->done = 0; } }
    if (dead) return;
    float* const Mb = Mblk + (size_t)j * mblk_stride;
    if (gid < K) cam_block_inverse(6, cdiag + 6 * gid, tri + 21 * gid, fix_poses != 0, radius, Mb + 36 * (size_t)gid);
    else if (gid == K) cam_block_inverse(4, cdiag + 6 * K, tri + 21 * K, fix_intr != 0, radius, Mb + 36 * (size_t)K);
    else if (gid == K + 1) cam_block_inverse(5, cdiag + 6 * K + 4, tri + 21 * K + 10, fix_dist != 0, radius, Mb + 36 * (size_t)K + 16);
    const int NS = 6 * K + 9;
    for (int i = gid; i < NS; i += gridDim.x * blockDim.x) { float d2, mi; lm_diag(tc[i], tS[i], inv_radius, d2, mi); tD2[(size_t)j * tail_stride + i] = d2; }
}
void launch_lm_begin_lad(hipStream_t st, LmState* lm, int B, int K, int fix_poses, int fix_intr, int fix_dist, const double* cdiag, const double* tri, float* Mblk, size_t mblk_stride,
                         const float* tc, const float* tS, float* tD2, size_t tail_stride, LmRecord* rec, int seq) {
    k_lm_begin_lad<<<dim3((K + 2 + 63) / 64, B), 64, 0, st>>>(lm, B, K, fix_poses, fix_intr, fix_dist, cdiag, tri, Mblk, mblk_stride, tc, tS, tD2, tail_stride, rec, seq);
}

// LM diagonal + 1x1 block-Jacobi inverses of the whole vector at the radius of the attempt in flight (sharded / untiled solve: the three-launch pass recomputes
// them from the column norms instead)
__global__ void k_lm_diag_dev(int n, const float* __restrict__ c, const float* __restrict__ S, const LmState* __restrict__ lm, float* __restrict__ D2, float* __restrict__ Minv) {
    if (lm->done) return;
    const float ir = lm->inv_radius;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { float d2, mi; lm_diag(c[i], S[i], ir, d2, mi); D2[i] = d2; Minv[i] = mi; }
}
void launch_lm_diag_dev(hipStream_t st, int n, const float* c, const float* S, const LmState* lm, float* D2, float* Minv) {
    if (n <= 0) return;
    int b = (n + 255) / 256; b = b > 2048 ? 2048 : b;
    k_lm_diag_dev<<<b, 256, 0, st>>>(n, c, S, lm, D2, Minv);
}

// per-keyframe constants of the candidate poses xc[0 .. 6K): the image pointers / sizes come from the assembled point's array
__global__ void __launch_bounds__(64) k_cand_frames(int K, const double* __restrict__ xc, const FrameConst* __restrict__ base, FrameConst* __restrict__ out, const LmState* __restrict__ lm) {
    if (lm->done) return;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= K) return;
    FrameConst fc = base[f];
    double p[6];
    for (int i = 0; i < 6; ++i) p[i] = xc[6 * f + i];
    fm::frame_from_pose(p, fc);
    out[f] = fc;
}
void launch_cand_frames(hipStream_t st, int K, const double* xc, const FrameConst* base, FrameConst* out, const LmState* lm) {
    if (K > 0) k_cand_frames<<<(K + 63) / 64, 64, 0, st>>>(K, xc, base, out, lm);
}

// One attempt decided (TrustRegionMinimizer::Minimize after the linear solve; the reference stops after the first successful step, nls_solver.cpp:279-293).
//   ps      terminal state of the PCG solve (x.(b+r), sum D^2 x^2 -> model_cost_change = -(J s)^T (r + J s / 2), s = -x)
//   norms2  |delta|^2, |x|^2 over the free parameters (k_candidate);  cand_cost: cost at the candidate (k_build<false>, all-reduced when sharded)
//   lad_next >= 0: this attempt is system lad_next - 1 of a ladder batch.  A rejection must leave the radius system lad_next was solved with; anything else (an invalid
//           step halves the radius) puts the batch out of step: done = 2, the NEXT attempt's record says "not decided" (kind 3) and the host solves it again alone.
//           The radius-underflow test k_lm_begin makes at the start of an attempt is made here for the next system of the batch, into the same record.
//   debug_invalid: tests only — treat this attempt's step as invalid (model_cost_change <= 0)
__global__ void k_lm_decide(LmState* lm, const PcgState* __restrict__ ps, const double* __restrict__ norms2, const double* __restrict__ cand_cost_p, int attempt /* 0-based */, int lm_steps,
                            LmRecord* rec, int seq, int lad_next, int debug_invalid) {
    if (lm->done) return;
    const int pcg_it = ps->done == 2 ? ps->it + 1 : ps->it;          // Ceres counts the iteration it broke in
    const double xbr = ps->xbr, d2xx = ps->d2xx;
    const double model_change = 0.5 * xbr + 0.5 * d2xx;
    const bool finite = !(isnan(xbr) || isinf(xbr) || isnan(d2xx) || isinf(d2xx));
    double radius = lm->radius, decrease = lm->decrease_factor, cost = lm->cost;
    const double cand = *cand_cost_p;
    int final_ = 0, accepted = 0, termination = lm->termination, invalid = lm->invalid, successful = lm->successful;
    double rel = 0.0;
    if (!finite || !(model_change > 0.0) || debug_invalid) {         // invalid step; TrustRegionMinimizer::HandleInvalidStep fails on the 5th in a row (++n >= max_num_consecutive_invalid_steps = 5)
        if (++invalid >= 5) { termination = 3; final_ = 1; }
        else radius *= 0.5;
    } else {
        invalid = 0;
        const double step_norm = sqrt(norms2[0]), x_norm = sqrt(norms2[1]);
        const double cost_change = cost - cand;
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; final_ = 1; }                      // parameter_tolerance
        else if (fabs(cost_change) <= 1e-6 * cost) { termination = 1; final_ = 1; }                    // function_tolerance
        else {
            rel = cost_change / model_change;
            if (rel > 1e-3) {                                        // min_relative_decrease
                accepted = 1; cost = cand;
                radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3.0)));
                successful += 1; termination = 2; final_ = 1;        // SuccessfulStepCallback: stop after the first successful step
            } else { radius = radius / decrease; decrease *= 2.0; }
        }
    }
    if (!final_ && attempt + 1 >= lm_steps) final_ = 1;              // max_num_iterations: termination stays 0
    lm->radius = radius; lm->decrease_factor = decrease; lm->cost = cost; lm->invalid = invalid; lm->successful = successful; lm->termination = termination;
    lm->attempts = lm->attempts + 1; lm->accepted = accepted; lm->done = final_;
    if (rec) {
        rec->final_ = final_; rec->accepted = accepted; rec->pcg_it = pcg_it; rec->termination = termination; rec->kind = 1;
        rec->cost = cost; rec->cand_cost = cand; rec->model_change = model_change; rec->rel = rel; rec->radius_after = radius; rec->ngrad = lm->ngrad; rec->nfree = lm->nfree;
        lm_publish(rec, seq);
    }
    if (!final_ && lad_next >= 0 && lad_next < lm->lad_n) {          // the next system of the batch
        LmRecord* const nrec = rec ? rec + 1 : nullptr;
        if (radius != lm->lad_radius[lad_next]) {
            lm->done = 2;
            if (nrec) { nrec->final_ = 0; nrec->accepted = 0; nrec->pcg_it = 0; nrec->termination = termination; nrec->kind = 3; nrec->cost = cost; nrec->cand_cost = 0.0; nrec->model_change = 0.0;
                        nrec->rel = 0.0; nrec->radius_after = radius; nrec->ngrad = lm->ngrad; nrec->nfree = lm->nfree; lm_publish(nrec, seq + 1); }
        } else if (radius < 1e-32) {                                 // LevenbergMarquardtStrategy: the radius cannot shrink further (k_lm_begin's test, for the attempt that would start now)
            lm->termination = 1; lm->done = 1;
            if (nrec) { nrec->final_ = 1; nrec->accepted = 0; nrec->pcg_it = 0; nrec->termination = 1; nrec->kind = 2; nrec->cost = cost; nrec->cand_cost = 0.0; nrec->model_change = 0.0;
                        nrec->rel = 0.0; nrec->radius_after = radius; nrec->ngrad = lm->ngrad; nrec->nfree = lm->nfree; lm_publish(nrec, seq + 1); }
        }
    }
}
void launch_lm_decide(hipStream_t st, LmState* lm, const PcgState* ps, const double* norms2, const double* cand_cost, int attempt, int lm_steps, LmRecord* rec, int seq, int lad_next, int debug_invalid) {
    k_lm_decide<<<1, 1, 0, st>>>(lm, ps, norms2, cand_cost, attempt, lm_steps, rec, seq, lad_next, debug_invalid);
}

}  // namespace i3d
