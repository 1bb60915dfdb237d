// Host control of one Optimizer::optimize call on the device:
//   assemble()  = Optimizer::addVoxelResiduals over all voxels + NLSSolver::buildProblem (optimizer.cpp:119-165, nls_solver.cpp:190-276)
//   lm_solve()  = NLSSolver::solve (nls_solver.cpp:296-367): Ceres 2.1.0 TrustRegionMinimizer + LevenbergMarquardtStrategy + CGNR with
//                 block-Jacobi preconditioning [Ceres is un-vendored; semantics per SURVEY.md Appendix B], stopping after the first
//                 successful step (SuccessfulStepCallback, nls_solver.cpp:279-293).
// Vectors are fp32 on the device; every reduction (dots, cost, weight sums, camera blocks) accumulates in fp64; the unknowns keep an fp64 master copy.
#include "context.hpp"
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <limits>

namespace i3d {

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static OptParams make_params(const i3d_context* c, const i3d_optimizer_config& cfg, int iteration, const double* intr, const double* dist) {
    OptParams p; std::memset(&p, 0, sizeof(p));
    p.thres_shell = cfg.thres_shell; p.lambda_a = cfg.lambda_a; p.K = c->K; p.level = cfg.rgbd_level;
    p.pyr_scale = 1.0 / std::pow(2.0, cfg.rgbd_level);                      // cost.h:146-150
    p.occlusion = cfg.occlusion_distance;
    p.use_er = (cfg.lambda_r0 > 0.0 && cfg.lambda_r1 > 0.0); p.use_es = (cfg.lambda_s0 > 0.0 && cfg.lambda_s1 > 0.0); p.use_ea = cfg.lambda_a > 0.0;
    for (int i = 0; i < 4; ++i) { p.intr[i] = intr[i]; p.cam_f[i] = (float)(intr[i] * p.pyr_scale); }   // optimizer.cpp:125, camera.cpp:95-98
    bool dz = true;
    for (int i = 0; i < 5; ++i) { p.dist[i] = dist[i]; p.dist_f[i] = (float)dist[i]; if (std::fabs(p.dist_f[i]) > 1e-5f) dz = false; }
    p.dist_zero = dz ? 1 : 0;
    p.w = c->fw[cfg.rgbd_level]; p.h = c->fh[cfg.rgbd_level];
    p.fix_poses = cfg.fix_poses; p.fix_intr = cfg.fix_intrinsics; p.fix_dist = cfg.fix_distortion;
    (void)iteration;
    return p;
}

static double varying_lambda(int it, int n, double l0, double l1) {        // cost.h:130-143
    if (n <= 1) return l0;
    return l0 + ((l1 - l0) / (double)(n - 1)) * (double)it;
}

static int alloc_rows(i3d_context* c, int slots) {
    const size_t Acap = (size_t)c->N;
    c->Acap = (int)Acap; c->slots = slots;
    CTX_HIP(c, c->obs_frame.alloc(Acap * slots)); CTX_HIP(c, c->obs_w.alloc(Acap * slots)); CTX_HIP(c, c->res.alloc(Acap * slots));
    CTX_HIP(c, c->roww.alloc(Acap * slots)); CTX_HIP(c, c->rowfree.alloc(Acap * slots)); CTX_HIP(c, c->J.alloc(Acap * slots * P_TOTAL));
    CTX_HIP(c, c->regflags.alloc(Acap)); CTX_HIP(c, c->ea_free.alloc(Acap)); CTX_HIP(c, c->ea_w.alloc(Acap * 6));
    CTX_HIP(c, c->C.alloc(Acap * P_VOX)); CTX_HIP(c, c->treg.alloc(Acap * 8));
    const size_t NP = 2 * (size_t)c->N + 6 * (size_t)c->K + 9;
    for (DevBuf<float>* v : {&c->v_mask, &c->v_c, &c->v_S, &c->v_D2, &c->v_Minv, &c->v_b, &c->v_x, &c->v_r, &c->v_p, &c->v_z, &c->v_q, &c->v_u, &c->v_acc, &c->v_tmp})
        CTX_HIP(c, v->alloc(NP));
    CTX_HIP(c, c->Minv_blocks.alloc((size_t)36 * c->K + 16 + 25));
    CTX_HIP(c, c->d_shared.alloc((size_t)6 * c->K + 9)); CTX_HIP(c, c->d_blocks.alloc((size_t)21 * c->K + 25));
    CTX_HIP(c, c->d_scal.alloc(16)); CTX_HIP(c, c->d_xshared.alloc((size_t)6 * c->K + 9)); CTX_HIP(c, c->d_xcshared.alloc((size_t)6 * c->K + 9));
    return ensure_pinned(c, 64 + (size_t)27 * c->K + 64);
}

// read `n` doubles from device memory (after everything queued on the stream)
static int read_doubles(i3d_context* c, const double* dptr, size_t n, double* out) {
    CTX_HIP(c, hipMemcpyAsync(c->h_pinned, dptr, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(out, c->h_pinned, n * sizeof(double));
    return I3D_OK;
}

int assemble(i3d_context* c, const i3d_optimizer_config& cfg, int iteration, OptParams& p, i3d_iteration_stats* st) {
    if (!c->have_grid || !c->have_frames || !c->have_camera) return ctx_fail(c, I3D_ERR_STATE, "optimize: grid, keyframes and camera must be set");
    if (cfg.rgbd_level < 0 || cfg.rgbd_level >= c->levels) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "optimize: rgbd_level outside the uploaded pyramid");
    int slots = (cfg.num_observations <= 0 || cfg.num_observations >= c->K) ? c->K : cfg.num_observations;
    if (slots > MAX_SLOTS) return ctx_fail(c, I3D_ERR_CAPACITY, "optimize: more than 8 observations per voxel requested");
    if (c->slots != slots || c->Acap != c->N || !c->J.p) { int rc = alloc_rows(c, slots); if (rc) return rc; }
    hipStream_t s = c->stream;
    p = make_params(c, cfg, iteration, c->intr, c->dist);
    std::vector<FrameConst> fc; build_frame_consts(c, cfg.rgbd_level, c->poses.data(), fc);
    CTX_HIP(c, hipMemcpyAsync(c->d_frames.p, fc.data(), sizeof(FrameConst) * fc.size(), hipMemcpyHostToDevice, s));
    GridView g = c->grid_view();
    { TimedScope t(c, I3D_K_CLASSIFY);
      launch_classify(s, g, p, c->aflag.p);
      CTX_HIP(c, rocprim::exclusive_scan(c->scan_tmp.p, c->scan_tmp_bytes, c->aflag.p, c->ascan.p, 0, (size_t)c->N, rocprim::plus<int>(), s));
      launch_compact(s, c->N, c->aflag.p, c->ascan.p, c->aidx.p, c->alist.p); }
    int tail[2];
    CTX_HIP(c, hipMemcpyAsync(&tail[0], c->ascan.p + (c->N - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    CTX_HIP(c, hipMemcpyAsync(&tail[1], c->aflag.p + (c->N - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    CTX_HIP(c, hipStreamSynchronize(s));
    c->A = tail[0] + tail[1];
    RowView r = c->row_view();
    { TimedScope t(c, I3D_K_OBSERVE); launch_observe(s, g, r, p, c->d_frames.p); }
    { TimedScope t(c, I3D_K_BUILD); launch_build(s, g, r, p, c->d_frames.p, true, nullptr); }
    CTX_HIP(c, hipMemsetAsync(c->d_scal.p, 0, sizeof(double) * 16, s));
    { TimedScope t(c, I3D_K_VECTOR); launch_weight_sums(s, r, c->d_scal.p); }
    double sums[8]; { int rc = read_doubles(c, c->d_scal.p, 8, sums); if (rc) return rc; }
    sums[5] = sums[1]; sums[6] = sums[2];
    const double lambda[4] = {cfg.lambda_g, varying_lambda(iteration, cfg.iterations, cfg.lambda_r0, cfg.lambda_r1),
                              varying_lambda(iteration, cfg.iterations, cfg.lambda_s0, cfg.lambda_s1), cfg.lambda_a};
    for (int t = 0; t < 4; ++t) p.type_w[t] = sums[t] != 0.0 ? (lambda[t] / sums[t]) * 1000.0 : 0.0;     // nls_solver.cpp:379-394
    if (st) for (int t = 0; t < 4; ++t) { st->rows[t] = (int64_t)(sums[4 + t] + 0.5); st->weight_sum[t] = sums[t]; st->type_weight[t] = p.type_w[t]; st->valid_voxels = c->A; }
    c->last_sizes[0] = c->A; for (int t = 0; t < 4; ++t) c->last_sizes[1 + t] = (long long)(sums[4 + t] + 0.5);
    c->last_params = p; c->assembled = true;
    CTX_HIP(c, hipGetLastError());
    return I3D_OK;
}

// ---- normal-equation pieces ----------------------------------------------------------------------------------
static int run_pass(i3d_context* c, PassMode mode, const OptParams& p, const float* u, float* out /*[NP]*/) {
    hipStream_t s = c->stream; GridView g = c->grid_view(); RowView r = c->row_view();
    PassBuffers b{c->C.p, c->treg.p, c->d_shared.p, c->d_blocks.p};
    CTX_HIP(c, hipMemsetAsync(c->d_shared.p, 0, sizeof(double) * (6 * (size_t)c->K + 9), s));
    if (mode == PASS_COLNORM) CTX_HIP(c, hipMemsetAsync(c->d_blocks.p, 0, sizeof(double) * (21 * (size_t)c->K + 25), s));
    { TimedScope t(c, I3D_K_EG_PASS); launch_eg_pass(s, mode, g, r, p, u, b); }
    { TimedScope t(c, I3D_K_GATHER); launch_gather(s, mode, g, r, b, out); }
    { TimedScope t(c, I3D_K_VECTOR); launch_shared_finalize(s, c->K, p, c->d_shared.p, out + 2 * (size_t)c->N); }
    return I3D_OK;
}

// q = S J^T W J S v + D2 v
static int apply_A(i3d_context* c, const OptParams& p, const float* v, float* q) {
    const int NP = 2 * c->N + 6 * c->K + 9;
    { TimedScope t(c, I3D_K_VECTOR); launch_mul(c->stream, NP, c->v_S.p, v, c->v_u.p); }
    int rc = run_pass(c, PASS_JTJP, p, c->v_u.p, c->v_acc.p); if (rc) return rc;
    { TimedScope t(c, I3D_K_VECTOR); launch_apply_op_tail(c->stream, NP, c->v_S.p, c->v_acc.p, c->v_D2.p, v, q); }
    return I3D_OK;
}

static int dot(i3d_context* c, int n, const float* a, const float* b, double* out) {
    CTX_HIP(c, hipMemsetAsync(c->d_scal.p, 0, sizeof(double) * 4, c->stream));
    { TimedScope t(c, I3D_K_VECTOR); launch_dot(c->stream, n, a, b, c->d_scal.p); }
    return read_doubles(c, c->d_scal.p, 1, out);
}

static bool spd_invert(int n, const double* m, double* inv) {            // Cholesky (Ceres: BlockRandomAccessDiagonalMatrix::Invert)
    double L[36];
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = m[i * n + j];
        for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
        if (i == j) { if (!(s > 0.0)) return false; L[i * n + i] = std::sqrt(s); } else L[i * n + j] = s / L[j * n + j];
    }
    for (int col = 0; col < n; ++col) {
        double y[6], x[6];
        for (int i = 0; i < n; ++i) { double s = (i == col) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
        for (int i = 0; i < n; ++i) inv[i * n + col] = x[i];
    }
    return true;
}

struct SharedBlocks { std::vector<double> c, H; };     // diag (6K+9) and upper triangles (21K+25) of J^T W J on the camera unknowns

// block-Jacobi inverse of the pose (6x6), intrinsics (4x4) and distortion (5x5) blocks of  S H S + D^2
static int upload_shared_precond(i3d_context* c, const OptParams& p, const SharedBlocks& sb, double radius) {
    const int K = c->K;
    std::vector<float> Minv((size_t)36 * K + 41, 0.0f);
    auto do_block = [&](int n, const double* cdiag, const double* tri, bool fixed, float* out) {
        if (fixed) return;
        double S[6], M[36], inv[36];
        for (int i = 0; i < n; ++i) S[i] = 1.0 / (1.0 + std::sqrt(cdiag[i]));
        int o = 0;
        for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { const double v = S[i] * S[j] * tri[o++]; M[i * n + j] = v; M[j * n + i] = v; }
        for (int i = 0; i < n; ++i) { const double cs = cdiag[i] * S[i] * S[i]; M[i * n + i] += std::min(std::max(cs, 1e-6), 1e32) / radius; }
        if (!spd_invert(n, M, inv)) { for (int i = 0; i < n * n; ++i) inv[i] = 0.0; for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0 / M[i * n + i]; }
        for (int i = 0; i < n * n; ++i) out[i] = (float)inv[i];
    };
    for (int f = 0; f < K; ++f) do_block(6, &sb.c[6 * f], &sb.H[21 * f], p.fix_poses, &Minv[36 * (size_t)f]);
    do_block(4, &sb.c[6 * K], &sb.H[21 * K], p.fix_intr, &Minv[36 * (size_t)K]);
    do_block(5, &sb.c[6 * K + 4], &sb.H[21 * K + 10], p.fix_dist, &Minv[36 * (size_t)K + 16]);
    CTX_HIP(c, hipMemcpyAsync(c->Minv_blocks.p, Minv.data(), sizeof(float) * Minv.size(), hipMemcpyHostToDevice, c->stream));
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    return I3D_OK;
}

static int precondition(i3d_context* c, const float* r, float* z) {
    const int N2 = 2 * c->N;
    TimedScope t(c, I3D_K_VECTOR);
    launch_mul(c->stream, N2, c->v_Minv.p, r, z);
    launch_precond_shared(c->stream, c->K, c->Minv_blocks.p, r + N2, z + N2);
    return I3D_OK;
}

static int eval_cost(i3d_context* c, const OptParams& p, bool candidate, const FrameConst* frames, double* cost) {
    GridView g = c->grid_view();
    if (candidate) { g.x_sdf = c->xc_sdf.p; g.x_alb = c->xc_alb.p; }
    CTX_HIP(c, hipMemsetAsync(c->d_scal.p + 8, 0, sizeof(double), c->stream));
    { TimedScope t(c, I3D_K_COST); launch_build(c->stream, g, c->row_view(), p, frames, false, c->d_scal.p + 8); }
    return read_doubles(c, c->d_scal.p + 8, 1, cost);
}

// NLSSolver::solve on the assembled rows.  Updates the device unknowns and the host camera when a step is accepted.
static int lm_solve(i3d_context* c, const i3d_optimizer_config& cfg, OptParams& p, i3d_iteration_stats* st) {
    hipStream_t s = c->stream;
    const int N = c->N, K = c->K, NP = 2 * N + 6 * K + 9, NS = 6 * K + 9;
    GridView g = c->grid_view();
    { TimedScope t(c, I3D_K_VECTOR); launch_freemask(s, g, p, c->v_mask.p); }
    // column norms -> Jacobi scaling (computed once, TrustRegionMinimizer::Init)
    int rc = run_pass(c, PASS_COLNORM, p, nullptr, c->v_c.p); if (rc) return rc;
    SharedBlocks sb; sb.c.resize(NS); sb.H.resize((size_t)21 * K + 25);
    rc = read_doubles(c, c->d_shared.p, NS, sb.c.data()); if (rc) return rc;
    {   // d_blocks can exceed the pinned scratch for large K: read in one go through a sized pinned buffer
        rc = ensure_pinned(c, (size_t)21 * K + 25 + 64); if (rc) return rc;
        rc = read_doubles(c, c->d_blocks.p, (size_t)21 * K + 25, sb.H.data()); if (rc) return rc;
    }
    { TimedScope t(c, I3D_K_VECTOR); launch_scale_from_colnorm(s, NP, c->v_c.p, c->v_mask.p, c->v_S.p); }
    // gradient b = S J^T W r and initial cost
    rc = run_pass(c, PASS_GRAD, p, nullptr, c->v_acc.p); if (rc) return rc;
    { TimedScope t(c, I3D_K_VECTOR); launch_mul(s, NP, c->v_S.p, c->v_acc.p, c->v_b.p); }
    double cost = 0.0; rc = eval_cost(c, p, false, c->d_frames.p, &cost); if (rc) return rc;
    double gmax2 = 0.0; rc = dot(c, NP, c->v_acc.p, c->v_acc.p, &gmax2); if (rc) return rc;
    double nfree = 0.0; rc = dot(c, NP, c->v_mask.p, c->v_mask.p, &nfree); if (rc) return rc;
    if (st) { st->cost_initial = cost; st->cost_final = cost; st->free_parameters = (int64_t)(nfree + 0.5); }
    c->last_sizes[5] = (long long)(nfree + 0.5);
    if (c->A == 0 || nfree == 0.0) { if (st) st->termination = 1; return I3D_OK; }
    if (gmax2 == 0.0) { if (st) st->termination = 1; return I3D_OK; }            // gradient_tolerance (max-norm <= 1e-10)

    std::vector<double> xshared(NS), xcshared(NS);
    for (int i = 0; i < 6 * K; ++i) xshared[i] = c->poses[i];
    for (int i = 0; i < 4; ++i) xshared[6 * K + i] = c->intr[i];
    for (int i = 0; i < 5; ++i) xshared[6 * K + 4 + i] = c->dist[i];
    CTX_HIP(c, hipMemcpyAsync(c->d_xshared.p, xshared.data(), sizeof(double) * NS, hipMemcpyHostToDevice, s));

    double radius = 1e4, decrease_factor = 2.0;
    int invalid = 0, attempts = 0;
    if (st) { st->termination = 0; st->final_radius = radius; }
    for (int iter = 1; iter <= cfg.lm_steps; ++iter) {
        if (radius < 1e-32) { if (st) st->termination = 1; break; }
        if (st) st->lm_iterations = iter;
        { TimedScope t(c, I3D_K_VECTOR); launch_lm_diag(s, NP, c->v_c.p, c->v_S.p, (float)(1.0 / radius), c->v_D2.p, c->v_Minv.p); }
        rc = upload_shared_precond(c, p, sb, radius); if (rc) return rc;
        // ---- CGNR (ConjugateGradientsSolver) on (S J^T W J S + D^2) x = b, x0 = 0 ----
        { TimedScope t(c, I3D_K_VECTOR); launch_fill(s, NP, c->v_x.p, 0.0f); }
        CTX_HIP(c, hipMemcpyAsync(c->v_r.p, c->v_b.p, sizeof(float) * (size_t)NP, hipMemcpyDeviceToDevice, s));
        double rho = 1.0, Q0 = 0.0; int it = 1;
        for (;; ++it) {
            rc = precondition(c, c->v_r.p, c->v_z.p); if (rc) return rc;
            const double last_rho = rho;
            rc = dot(c, NP, c->v_r.p, c->v_z.p, &rho); if (rc) return rc;
            if (rho == 0.0 || !std::isfinite(rho)) break;
            if (it == 1) CTX_HIP(c, hipMemcpyAsync(c->v_p.p, c->v_z.p, sizeof(float) * (size_t)NP, hipMemcpyDeviceToDevice, s));
            else { const double beta = rho / last_rho; if (beta == 0.0 || !std::isfinite(beta)) break;
                   TimedScope t(c, I3D_K_VECTOR); launch_xpay(s, NP, c->v_z.p, (float)beta, c->v_p.p); }
            rc = apply_A(c, p, c->v_p.p, c->v_q.p); if (rc) return rc;
            double pq = 0.0; rc = dot(c, NP, c->v_p.p, c->v_q.p, &pq); if (rc) return rc;
            if (pq <= 0.0 || std::isinf(pq)) break;
            const double alpha = rho / pq;
            if (std::isinf(alpha)) break;
            { TimedScope t(c, I3D_K_VECTOR); launch_axpy(s, NP, (float)alpha, c->v_p.p, c->v_x.p); }
            if (it % 10 == 0) { rc = apply_A(c, p, c->v_x.p, c->v_tmp.p); if (rc) return rc; TimedScope t(c, I3D_K_VECTOR); launch_sub(s, NP, c->v_b.p, c->v_tmp.p, c->v_r.p); }
            else { TimedScope t(c, I3D_K_VECTOR); launch_axpy(s, NP, (float)(-alpha), c->v_q.p, c->v_r.p); }
            double q3[3];
            CTX_HIP(c, hipMemsetAsync(c->d_scal.p, 0, sizeof(double) * 4, s));
            { TimedScope t(c, I3D_K_VECTOR); launch_dot3(s, NP, c->v_x.p, c->v_b.p, c->v_r.p, c->v_D2.p, c->d_scal.p); }
            rc = read_doubles(c, c->d_scal.p, 3, q3); if (rc) return rc;
            const double Q1 = -q3[0];
            if (cfg.pcg_fixed_iterations >= 0) { if (it >= cfg.pcg_fixed_iterations) break; Q0 = Q1; continue; }
            const double zeta = it * (Q1 - Q0) / Q1;
            if (zeta < 0.1) break;                               // eta = 0.1, min_num_iterations = 0
            Q0 = Q1;
            if (it >= 500) break;                                // max_linear_solver_iterations
        }
        if (st && attempts < 50) st->pcg_iterations[attempts] = it;
        // model_cost_change = -(J s)^T (r + J s / 2) with s = -x  ==  x.b/2 + x.r_cg/2 + sum D^2 x^2 / 2
        double q3[3];
        CTX_HIP(c, hipMemsetAsync(c->d_scal.p, 0, sizeof(double) * 4, s));
        { TimedScope t(c, I3D_K_VECTOR); launch_dot3(s, NP, c->v_x.p, c->v_b.p, c->v_r.p, c->v_D2.p, c->d_scal.p); }
        rc = read_doubles(c, c->d_scal.p, 3, q3); if (rc) return rc;
        const double model_change = 0.5 * q3[0] + 0.5 * q3[2];
        const bool finite = std::isfinite(q3[0]) && std::isfinite(q3[2]);
        if (!finite || !(model_change > 0.0)) {                  // invalid step (max_num_consecutive_invalid_steps = 5)
            if (st && attempts < 50) st->step_accepted[attempts] = 0;
            ++attempts;
            if (++invalid > 5) { if (st) st->termination = 3; break; }
            radius *= 0.5; if (st) st->final_radius = radius; continue;
        }
        invalid = 0;
        // candidate point
        CTX_HIP(c, hipMemsetAsync(c->d_scal.p + 4, 0, sizeof(double) * 2, s));
        { TimedScope t(c, I3D_K_VECTOR); launch_candidate(s, g, K, -1.0f, c->v_x.p, c->v_S.p, c->d_xshared.p, c->xc_sdf.p, c->xc_alb.p, c->d_xcshared.p, c->d_scal.p + 4, c->v_mask.p); }
        double norms[2]; rc = read_doubles(c, c->d_scal.p + 4, 2, norms); if (rc) return rc;
        rc = read_doubles(c, c->d_xcshared.p, NS, xcshared.data()); if (rc) return rc;
        OptParams pc = p;
        for (int i = 0; i < 4; ++i) pc.intr[i] = xcshared[6 * K + i];
        for (int i = 0; i < 5; ++i) pc.dist[i] = xcshared[6 * K + 4 + i];
        std::vector<FrameConst> fcc; build_frame_consts(c, cfg.rgbd_level, xcshared.data(), fcc);
        CTX_HIP(c, hipMemcpyAsync(c->d_frames_cand.p, fcc.data(), sizeof(FrameConst) * fcc.size(), hipMemcpyHostToDevice, s));
        double cand_cost = 0.0; rc = eval_cost(c, pc, true, c->d_frames_cand.p, &cand_cost); if (rc) return rc;
        const double step_norm = std::sqrt(norms[0]), x_norm = std::sqrt(norms[1]);
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { if (st) { if (attempts < 50) st->step_accepted[attempts] = 0; st->termination = 1; } ++attempts; break; }
        const double cost_change = cost - cand_cost;
        if (std::fabs(cost_change) <= 1e-6 * cost) { if (st) { if (attempts < 50) st->step_accepted[attempts] = 0; st->termination = 1; } ++attempts; break; }
        const double rel = cost_change / model_change;
        if (cfg.verbose) std::printf("  [i3d LM] it %d cost %.9e cand %.9e model %.3e rho %.4f radius %.3e cg %d\n", iter, cost, cand_cost, model_change, rel, radius, it);
        if (rel > 1e-3) {                                        // min_relative_decrease
            { TimedScope t(c, I3D_K_VECTOR); launch_accept(s, g, c->xc_sdf.p, c->xc_alb.p); }
            for (int i = 0; i < 6 * K; ++i) c->poses[i] = xcshared[i];
            for (int i = 0; i < 4; ++i) c->intr[i] = xcshared[6 * K + i];
            for (int i = 0; i < 5; ++i) c->dist[i] = xcshared[6 * K + 4 + i];
            cost = cand_cost;
            radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
            if (st) { st->cost_final = cost; st->successful_steps += 1; if (attempts < 50) st->step_accepted[attempts] = 1; st->final_radius = radius; st->termination = 2; }
            ++attempts;
            break;                                               // SuccessfulStepCallback: stop after the first successful step
        }
        if (st && attempts < 50) st->step_accepted[attempts] = 0;
        ++attempts;
        radius = radius / decrease_factor; decrease_factor *= 2.0;
        if (st) st->final_radius = radius;
    }
    if (st) st->num_attempts = std::min(attempts, 50);
    CTX_HIP(c, hipStreamSynchronize(s));
    return I3D_OK;
}

int optimize(i3d_context* c, const i3d_optimizer_config& cfg, i3d_iteration_stats* stats) {
    if (!c->have_grid || cfg.iterations < 1) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "optimize: no grid or iterations < 1");   // optimizer.cpp:113-114
    CTX_HIP(c, hipSetDevice(c->device));
    for (int itr = 0; itr < cfg.iterations; ++itr) {
        i3d_iteration_stats local; std::memset(&local, 0, sizeof(local));
        i3d_iteration_stats* st = stats ? &stats[itr] : &local;
        std::memset(st, 0, sizeof(*st));
        OptParams p;
        const double t0 = now_s();
        int rc = assemble(c, cfg, itr, p, st); if (rc) return rc;
        const double t1 = now_s();
        st->time_add = t1 - t0;
        if (c->A > 0) { rc = lm_solve(c, cfg, p, st); if (rc) return rc; }
        const double t2 = now_s();
        st->time_solve = t2 - t1; st->time_build = 0.0;
        if (cfg.verbose) std::printf("[i3d] itr %d rows %lld/%lld/%lld/%lld valid %lld cost %.9e -> %.9e (add %.3f ms, solve %.3f ms)\n", itr,
                                     (long long)st->rows[0], (long long)st->rows[1], (long long)st->rows[2], (long long)st->rows[3], (long long)st->valid_voxels,
                                     st->cost_initial, st->cost_final, st->time_add * 1e3, st->time_solve * 1e3);
        timing_flush(c);
    }
    c->assembled = false;
    return I3D_OK;
}

// ---- parity probes ------------------------------------------------------------------------------------------------
static int to_visit_order(i3d_context* c, const float* dev_vec, double* out) {
    const int N = c->N, K = c->K, NP = 2 * N + 6 * K + 9;
    std::vector<float> h(NP); std::vector<int> rank(N);
    CTX_HIP(c, hipMemcpy(h.data(), dev_vec, sizeof(float) * (size_t)NP, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    for (int s = 0; s < N; ++s) { out[rank[s]] = h[s]; out[N + rank[s]] = h[N + s]; }
    for (int i = 2 * N; i < NP; ++i) out[i] = h[i];
    return I3D_OK;
}

int normal_eq_debug(i3d_context* c, double* gradient, double* jtj_diag, double* cost) {
    if (!c->assembled) return ctx_fail(c, I3D_ERR_STATE, "debug: call i3d_debug_assemble first");
    CTX_HIP(c, hipSetDevice(c->device));
    OptParams p = c->last_params;
    launch_freemask(c->stream, c->grid_view(), p, c->v_mask.p);
    int rc;
    if (jtj_diag) { rc = run_pass(c, PASS_COLNORM, p, nullptr, c->v_c.p); if (rc) return rc; CTX_HIP(c, hipStreamSynchronize(c->stream)); rc = to_visit_order(c, c->v_c.p, jtj_diag); if (rc) return rc; }
    if (gradient) { rc = run_pass(c, PASS_GRAD, p, nullptr, c->v_acc.p); if (rc) return rc; CTX_HIP(c, hipStreamSynchronize(c->stream)); rc = to_visit_order(c, c->v_acc.p, gradient); if (rc) return rc; }
    if (cost) { rc = eval_cost(c, p, false, c->d_frames.p, cost); if (rc) return rc; }
    return I3D_OK;
}

int jtj_apply_debug(i3d_context* c, const double* x, double* y) {
    if (!c->assembled) return ctx_fail(c, I3D_ERR_STATE, "debug: call i3d_debug_assemble first");
    CTX_HIP(c, hipSetDevice(c->device));
    const int N = c->N, K = c->K, NP = 2 * N + 6 * K + 9;
    OptParams p = c->last_params;
    std::vector<int> rank(N); std::vector<float> h(NP), m(NP);
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    launch_freemask(c->stream, c->grid_view(), p, c->v_mask.p);
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    CTX_HIP(c, hipMemcpy(m.data(), c->v_mask.p, sizeof(float) * (size_t)NP, hipMemcpyDeviceToHost));
    for (int s = 0; s < N; ++s) { h[s] = (float)x[rank[s]] * m[s]; h[N + s] = (float)x[N + rank[s]] * m[N + s]; }
    for (int i = 2 * N; i < NP; ++i) h[i] = (float)x[i] * m[i];
    CTX_HIP(c, hipMemcpy(c->v_u.p, h.data(), sizeof(float) * (size_t)NP, hipMemcpyHostToDevice));
    int rc = run_pass(c, PASS_JTJP, p, c->v_u.p, c->v_acc.p); if (rc) return rc;
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    return to_visit_order(c, c->v_acc.p, y);
}

}  // namespace i3d
