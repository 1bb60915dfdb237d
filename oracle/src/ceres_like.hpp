// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Restatement of the part of Ceres Solver 2.1.0 the reference drives
// (nls_solver.cpp:296-337, lighting_svsh.cpp:325-341): trust-region
// Levenberg-Marquardt (TrustRegionMinimizer + LevenbergMarquardtStrategy) with
// linear_solver_type = CGNR and the default JACOBI (block-Jacobi) preconditioner.
// Ceres is an un-vendored dependency pinned only by /root/reference/README.md:75;
// its sources are NOT under /root/reference, so this follows the published 2.1.0
// algorithm.  PARITY UNPINNED: the reference ships no test
// or golden vector for it.
//
// Defaults that matter (Solver::Options, 2.1.0): initial_trust_region_radius 1e4,
// max 1e16, min 1e-32, min_relative_decrease 1e-3, min/max_lm_diagonal 1e-6/1e32,
// jacobi_scaling, eta 0.1, max_linear_solver_iterations 500, min 0,
// residual_reset_period 10, function_tolerance 1e-6, gradient_tolerance 1e-10,
// parameter_tolerance 1e-8, max_num_consecutive_invalid_steps 5.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <limits>
#include <vector>

namespace orc {

struct CRS {                       // rows x cols, fixed structure, values refreshed by evaluate()
    int rows = 0, cols = 0;
    std::vector<int> ptr, col;
    std::vector<double> val;
    void mul(const double* x, double* y) const {          // y = A x
        for (int r = 0; r < rows; ++r) { double s = 0.0; for (int k = ptr[r]; k < ptr[r + 1]; ++k) s += val[k] * x[col[k]]; y[r] = s; } }
    void mul_t_add(const double* y, double* x) const {     // x += A^T y
        for (int r = 0; r < rows; ++r) { const double yr = y[r]; for (int k = ptr[r]; k < ptr[r + 1]; ++k) x[col[k]] += val[k] * yr; } }
    void col_sq_norm(double* out) const {
        for (int c = 0; c < cols; ++c) out[c] = 0.0;
        for (size_t k = 0; k < val.size(); ++k) out[col[k]] += val[k] * val[k]; }
    void scale_cols(const double* s) { for (size_t k = 0; k < val.size(); ++k) val[k] *= s[col[k]]; }
};

struct LMOptions {
    int max_num_iterations = 50;
    bool stop_after_first_successful_step = false;    // nls_solver.cpp:279-293
    int cg_fixed_iterations = -1;                      // parity pinning; -1 = Ceres' Q-test
    double initial_radius = 1e4, max_radius = 1e16, min_radius = 1e-32;
    double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    double eta = 1e-1; int max_cg_iterations = 500, min_cg_iterations = 0, residual_reset_period = 10;
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    int max_consecutive_invalid_steps = 5;
    bool verbose = false;
};

struct LMSummary {
    double initial_cost = 0, final_cost = 0;
    int iterations = 0, successful_steps = 0;
    double final_radius = 0;
    std::vector<int> cg_iterations;          // one per LM attempt
    std::vector<int> step_accepted;          // 1/0 per LM attempt
    int termination = 0;                     // 0 no-conv, 1 convergence, 2 user success (first successful step), 3 failure
};

// evaluate(x, &cost, residuals, J) : residuals/J already include the sqrt(loss weight) scaling. J may be null.
typedef std::function<bool(const double* x, double* cost, std::vector<double>* residuals, CRS* J)> EvalFn;

// dense SPD inverse via Cholesky (block sizes 1..9) — BlockRandomAccessDiagonalMatrix::Invert
inline bool spd_invert(int n, const double* m, double* inv) {
    double L[81];
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = m[i * n + j];
        for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
        if (i == j) { if (!(s > 0.0)) return false; L[i * n + i] = std::sqrt(s); }
        else L[i * n + j] = s / L[j * n + j];
    }
    for (int c = 0; c < n; ++c) {            // solve L L^T x = e_c
        double y[9], x[9];
        for (int i = 0; i < n; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
        for (int i = 0; i < n; ++i) inv[i * n + c] = x[i];
    }
    return true;
}

struct BlockJacobi {       // block_jacobi_preconditioner.cc
    std::vector<int> start, size; std::vector<int> off; std::vector<double> inv;
    void update(const CRS& A, const std::vector<int>& col_block, const double* D) {
        std::vector<double> m(inv.size(), 0.0);
        for (int r = 0; r < A.rows; ++r)
            for (int k = A.ptr[r]; k < A.ptr[r + 1]; ++k) {
                const int b = col_block[A.col[k]]; const int n = size[b]; const int i = A.col[k] - start[b];
                if (n == 1) { m[off[b]] += A.val[k] * A.val[k]; continue; }
                for (int k2 = A.ptr[r]; k2 < A.ptr[r + 1]; ++k2) {
                    if (col_block[A.col[k2]] != b) continue;
                    m[off[b] + i * n + (A.col[k2] - start[b])] += A.val[k] * A.val[k2];
                }
            }
        for (size_t b = 0; b < start.size(); ++b) {
            const int n = size[b];
            for (int i = 0; i < n; ++i) m[off[b] + i * n + i] += D[start[b] + i] * D[start[b] + i];
            if (n == 1) inv[off[b]] = 1.0 / m[off[b]];
            else spd_invert(n, &m[off[b]], &inv[off[b]]);
        }
    }
    void apply(const double* r, double* z) const {
        for (size_t b = 0; b < start.size(); ++b) {
            const int n = size[b];
            for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s += inv[off[b] + i * n + j] * r[start[b] + j]; z[start[b] + i] = s; }
        }
    }
};

// conjugate_gradients_solver.cc on (A^T A + D^2) x = A^T b, x0 = 0
inline int cgnr_solve(const CRS& A, const double* b, const double* D, const BlockJacobi& M,
                      const LMOptions& opt, double* x) {
    const int n = A.cols, m = A.rows;
    std::vector<double> rhs(n, 0.0), r(n), p(n, 0.0), z(n), tmp(n), t(m);
    A.mul_t_add(b, rhs.data());
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    double nb = 0.0; for (int i = 0; i < n; ++i) nb += rhs[i] * rhs[i];
    if (std::sqrt(nb) == 0.0) return 0;
    auto apply = [&](const double* v, double* out) {
        A.mul(v, t.data()); for (int i = 0; i < n; ++i) out[i] = 0.0; A.mul_t_add(t.data(), out);
        for (int i = 0; i < n; ++i) out[i] += D[i] * D[i] * v[i]; };
    apply(x, tmp.data());
    for (int i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i];
    double rho = 1.0, Q0 = 0.0;
    for (int i = 0; i < n; ++i) Q0 += x[i] * (rhs[i] + r[i]);
    Q0 = -Q0;
    int it = 1;
    for (;; ++it) {
        M.apply(r.data(), z.data());
        const double last_rho = rho;
        rho = 0.0; for (int i = 0; i < n; ++i) rho += r[i] * z[i];
        if (rho == 0.0 || !std::isfinite(rho)) break;
        if (it == 1) p = z;
        else { const double beta = rho / last_rho; if (beta == 0.0 || !std::isfinite(beta)) break; for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i]; }
        std::vector<double>& q = z;
        apply(p.data(), q.data());
        double pq = 0.0; for (int i = 0; i < n; ++i) pq += p[i] * q[i];
        if (pq <= 0.0 || std::isinf(pq)) break;
        const double alpha = rho / pq;
        if (std::isinf(alpha)) break;
        for (int i = 0; i < n; ++i) x[i] += alpha * p[i];
        if (it % opt.residual_reset_period == 0) { apply(x, tmp.data()); for (int i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i]; }
        else for (int i = 0; i < n; ++i) r[i] -= alpha * q[i];
        double Q1 = 0.0; for (int i = 0; i < n; ++i) Q1 += x[i] * (rhs[i] + r[i]);
        Q1 = -Q1;
        if (opt.cg_fixed_iterations >= 0) { if (it >= opt.cg_fixed_iterations) break; Q0 = Q1; continue; }
        const double zeta = it * (Q1 - Q0) / Q1;
        if (zeta < opt.eta && it >= opt.min_cg_iterations) break;
        Q0 = Q1;
        if (it >= opt.max_cg_iterations) break;
    }
    return it;
}

// col_block[c] = parameter block of column c; blocks are contiguous column ranges.
inline LMSummary lm_minimize(const EvalFn& evaluate, CRS& J, const std::vector<int>& block_start,
                             const std::vector<int>& block_size, std::vector<double>& x, const LMOptions& opt) {
    LMSummary sum;
    const int n = (int)x.size();
    std::vector<int> col_block(n);
    BlockJacobi M; M.start = block_start; M.size = block_size; M.off.resize(block_start.size());
    { int o = 0; for (size_t b = 0; b < block_start.size(); ++b) { M.off[b] = o; o += block_size[b] * block_size[b];
        for (int i = 0; i < block_size[b]; ++i) col_block[block_start[b] + i] = (int)b; } M.inv.assign(o, 0.0); }

    std::vector<double> res, cand_res;
    double cost = 0.0;
    if (!evaluate(x.data(), &cost, &res, &J)) { sum.termination = 3; return sum; }
    sum.initial_cost = cost; sum.final_cost = cost;
    const int m = J.rows;
    std::vector<double> grad(n, 0.0);
    J.mul_t_add(res.data(), grad.data());
    std::vector<double> scale(n);
    J.col_sq_norm(scale.data());
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(scale[i]));
    J.scale_cols(scale.data());
    double gmax = 0.0; for (int i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(grad[i]));
    double radius = opt.initial_radius, decrease_factor = 2.0; bool reuse_diag = false;
    sum.final_radius = radius;
    if (gmax <= opt.gradient_tolerance) { sum.termination = 1; return sum; }
    double xnorm = 0.0; for (int i = 0; i < n; ++i) xnorm += x[i] * x[i]; xnorm = std::sqrt(xnorm);

    std::vector<double> diag(n), D(n), step(n), delta(n), xc(n), Jstep(m);
    int invalid = 0;
    for (int iter = 1; iter <= opt.max_num_iterations; ++iter) {
        if (radius < opt.min_radius) { sum.termination = 1; break; }
        sum.iterations = iter;
        if (!reuse_diag) { J.col_sq_norm(diag.data()); for (int i = 0; i < n; ++i) diag[i] = std::min(std::max(diag[i], opt.min_lm_diagonal), opt.max_lm_diagonal); }
        for (int i = 0; i < n; ++i) D[i] = std::sqrt(diag[i] / radius);
        M.update(J, col_block, D.data());
        const int cg_it = cgnr_solve(J, res.data(), D.data(), M, opt, step.data());
        sum.cg_iterations.push_back(cg_it);
        bool finite = true; for (int i = 0; i < n; ++i) { step[i] = -step[i]; if (!std::isfinite(step[i])) finite = false; }
        reuse_diag = true;
        double model_change = 0.0;
        if (finite) {
            J.mul(step.data(), Jstep.data());
            for (int r = 0; r < m; ++r) model_change += Jstep[r] * (res[r] + Jstep[r] / 2.0);
            model_change = -model_change;
        }
        if (!finite || !(model_change > 0.0)) {            // invalid step
            sum.step_accepted.push_back(0);
            if (++invalid >= opt.max_consecutive_invalid_steps) { sum.termination = 3; break; }      // TrustRegionMinimizer::HandleInvalidStep: fails ON the 5th consecutive invalid step
            radius *= 0.5; reuse_diag = false; continue;
        }
        invalid = 0;
        double step_norm = 0.0;
        for (int i = 0; i < n; ++i) { delta[i] = step[i] * scale[i]; xc[i] = x[i] + delta[i]; step_norm += (xc[i] - x[i]) * (xc[i] - x[i]); }
        step_norm = std::sqrt(step_norm);
        double cand_cost = 0.0;
        if (!evaluate(xc.data(), &cand_cost, &cand_res, nullptr)) cand_cost = std::numeric_limits<double>::max();
        if (step_norm <= opt.parameter_tolerance * (xnorm + opt.parameter_tolerance)) { sum.step_accepted.push_back(0); sum.termination = 1; break; }
        const double cost_change = cost - cand_cost;
        if (std::fabs(cost_change) <= opt.function_tolerance * cost) { sum.step_accepted.push_back(0); sum.termination = 1; break; }
        const double rel = cost_change / model_change;
        if (opt.verbose) std::printf("  [oracle LM] it %d cost %.9e cand %.9e model %.3e rho %.4f radius %.3e cg %d\n", iter, cost, cand_cost, model_change, rel, radius, cg_it);
        if (rel > opt.min_relative_decrease) {
            x = xc; xnorm = 0.0; for (int i = 0; i < n; ++i) xnorm += x[i] * x[i]; xnorm = std::sqrt(xnorm);
            evaluate(x.data(), &cost, &res, &J);     // Ceres re-evaluates r,J here (thrown away by the reference's callback)
            J.scale_cols(scale.data());
            sum.final_cost = cost; ++sum.successful_steps; sum.step_accepted.push_back(1);
            radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
            radius = std::min(opt.max_radius, radius); decrease_factor = 2.0; reuse_diag = false;
            sum.final_radius = radius;
            if (opt.stop_after_first_successful_step) { sum.termination = 2; break; }
            std::fill(grad.begin(), grad.end(), 0.0);
            // gradient with the unscaled Jacobian: g_j = (Js^T r)_j / scale_j
            J.mul_t_add(res.data(), grad.data());
            gmax = 0.0; for (int i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(grad[i] / scale[i]));
            if (gmax <= opt.gradient_tolerance) { sum.termination = 1; break; }
        } else {
            sum.step_accepted.push_back(0);
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = true;
            sum.final_radius = radius;
        }
    }
    return sum;
}

}  // namespace orc
