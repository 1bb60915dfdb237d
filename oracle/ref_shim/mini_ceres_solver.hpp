// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Second stand-in layer for Ceres Solver 2.1.0 [un-vendored dependency of /root/reference, pinned only by its README.md:75; absent
// from this image]: the classes the reference's NLSSolver / Optimizer / LightingSVSH *drive* — CostFunction,
// AutoDiffCostFunction, DynamicAutoDiffCostFunction, ScaledLoss, Problem, Solver::Options / Summary, IterationCallback and
// ceres::Solve (trust-region Levenberg-Marquardt + CGNR + block-Jacobi).  With it the reference's own
// optimizer.cpp / nls_solver.cpp / lighting_svsh.cpp run unmodified inside oracle/_ref.
//
// Written from the published 2.1.0 algorithm and deliberately NOT sharing code with oracle/src/ceres_like.hpp: this one works on
// per-residual-block dense Jacobian blocks of a generic Problem, the oracle's on one CRS matrix of its specialised rows — two
// implementations of the same semantics that the tests hold against each other.  Ceres itself stays unpinned (no tarball here).
// Nothing of this is reference code.
#pragma once
#include <cstdio>
#include <functional>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace ceres {

enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum LoggingType { SILENT, PER_MINIMIZER_ITERATION };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };

class CostFunction {
public:
    CostFunction() : num_residuals_(0) {}
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
    const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }
protected:
    std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }
private:
    std::vector<int32_t> parameter_block_sizes_;
    int num_residuals_;
};

// autodiff_cost_function.h: static block sizes, functor(const T* p0, ..., T* residuals); row-major jacobians[i][r * Ni + c]
template <typename Functor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public CostFunction {
public:
    explicit AutoDiffCostFunction(Functor* f) : functor_(f) { set_num_residuals(kNumResiduals); for (int n : {Ns...}) mutable_parameter_block_sizes()->push_back(n); }
    const Functor& functor() const { return *functor_; }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        if (!jacobians) return call(parameters, residuals, std::make_index_sequence<sizeof...(Ns)>());
        constexpr int kTotal = total();
        typedef Jet<double, kTotal> J;
        const int sizes[] = {Ns...};
        std::vector<J> x((size_t)kTotal); const J* ptr[sizeof...(Ns)];
        int o = 0;
        for (size_t b = 0; b < sizeof...(Ns); ++b) { ptr[b] = x.data() + o; for (int i = 0; i < sizes[b]; ++i) x[o + i] = J(parameters[b][i], o + i); o += sizes[b]; }
        J out[kNumResiduals];
        if (!call(ptr, out, std::make_index_sequence<sizeof...(Ns)>())) return false;
        for (int r = 0; r < kNumResiduals; ++r) residuals[r] = out[r].a;
        o = 0;
        for (size_t b = 0; b < sizeof...(Ns); ++b) {
            if (jacobians[b]) for (int r = 0; r < kNumResiduals; ++r) for (int i = 0; i < sizes[b]; ++i) jacobians[b][r * sizes[b] + i] = out[r].v[o + i];
            o += sizes[b];
        }
        return true;
    }
private:
    static constexpr int total() { int s = 0; for (int n : {Ns...}) s += n; return s; }
    template <typename T, size_t... I> bool call(T const* const* p, T* out, std::index_sequence<I...>) const { return (*functor_)(p[I]..., out); }
    std::unique_ptr<Functor> functor_;
};

// dynamic_autodiff_cost_function.h: functor(T const* const* params, T* residuals), derivatives in passes of Stride
template <typename Functor, int Stride = 4>
class DynamicAutoDiffCostFunction : public CostFunction {
public:
    explicit DynamicAutoDiffCostFunction(Functor* f) : functor_(f) {}
    void AddParameterBlock(int size) { mutable_parameter_block_sizes()->push_back(size); }
    void SetNumResiduals(int n) { set_num_residuals(n); }
    const Functor& functor() const { return *functor_; }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        if (!jacobians) return (*functor_)(parameters, residuals);
        typedef Jet<double, Stride> J;
        const std::vector<int32_t>& sizes = parameter_block_sizes();
        const int nb = (int)sizes.size(), nr = num_residuals();
        int total = 0; std::vector<int> start((size_t)nb);
        for (int b = 0; b < nb; ++b) { start[b] = total; total += sizes[b]; }
        std::vector<J> x((size_t)total), out((size_t)nr); std::vector<const J*> ptr((size_t)nb);
        for (int b = 0; b < nb; ++b) ptr[b] = x.data() + start[b];
        bool have_value = false;
        for (int pass = 0; pass * Stride < total; ++pass) {
            int k = 0;
            for (int b = 0; b < nb; ++b) for (int i = 0; i < sizes[b]; ++i, ++k) {
                x[k] = J(parameters[b][i]);
                const int s = k - pass * Stride;
                if (s >= 0 && s < Stride && jacobians[b]) x[k].v[s] = 1.0;
            }
            if (!(*functor_)(ptr.data(), out.data())) return false;
            if (!have_value) { for (int r = 0; r < nr; ++r) residuals[r] = out[r].a; have_value = true; }
            k = 0;
            for (int b = 0; b < nb; ++b) for (int i = 0; i < sizes[b]; ++i, ++k) {
                const int s = k - pass * Stride;
                if (s >= 0 && s < Stride && jacobians[b]) for (int r = 0; r < nr; ++r) jacobians[b][r * sizes[b] + i] = out[r].v[s];
            }
        }
        if (!have_value) return (*functor_)(parameters, residuals);
        return true;
    }
private:
    std::unique_ptr<Functor> functor_;
};

class LossFunction { public: virtual ~LossFunction() {} virtual void Evaluate(double sq_norm, double out[3]) const = 0; };
class ScaledLoss : public LossFunction {            // loss_function.h: rho(s) = a * f(s); f == nullptr is the identity
public:
    ScaledLoss(const LossFunction* rho, double a, Ownership) : rho_(rho), a_(a) {}
    void Evaluate(double s, double out[3]) const override {
        if (!rho_) { out[0] = a_ * s; out[1] = a_; out[2] = 0.0; return; }
        rho_->Evaluate(s, out); out[0] *= a_; out[1] *= a_; out[2] *= a_;
    }
    double scale() const { return a_; }
private:
    const LossFunction* rho_; double a_;
};

struct ResidualBlock { CostFunction* cost; LossFunction* loss; std::vector<double*> params; };

class Problem {
public:
    struct ParamInfo { int size; bool constant; int order; };
    Problem() {}
    ~Problem() {
        std::set<CostFunction*> cs; std::set<LossFunction*> ls;
        for (auto& b : blocks_) { cs.insert(b.cost); if (b.loss) ls.insert(b.loss); }
        for (auto* c : cs) delete c;
        for (auto* l : ls) delete l;
    }
    void AddResidualBlock(CostFunction* cost, LossFunction* loss, const std::vector<double*>& params) {
        blocks_.push_back(ResidualBlock{cost, loss, params});
        const std::vector<int32_t>& sizes = cost->parameter_block_sizes();
        for (size_t i = 0; i < params.size(); ++i)
            if (!params_.count(params[i])) { params_[params[i]] = ParamInfo{sizes[i], false, (int)order_.size()}; order_.push_back(params[i]); }
    }
    bool HasParameterBlock(const double* p) const { return params_.count(const_cast<double*>(p)) != 0; }
    void SetParameterBlockConstant(const double* p) { params_.at(const_cast<double*>(p)).constant = true; }
    int NumResiduals() const { int n = 0; for (auto& b : blocks_) n += b.cost->num_residuals(); return n; }
    int NumParameters() const { int n = 0; for (auto& kv : params_) n += kv.second.size; return n; }
    int NumResidualBlocks() const { return (int)blocks_.size(); }
    int NumParameterBlocks() const { return (int)order_.size(); }
    const std::vector<ResidualBlock>& blocks() const { return blocks_; }
    const std::vector<double*>& parameter_order() const { return order_; }
    const ParamInfo& info(double* p) const { return params_.at(p); }
private:
    std::vector<ResidualBlock> blocks_;
    std::unordered_map<double*, ParamInfo> params_;
    std::vector<double*> order_;
    Problem(const Problem&); Problem& operator=(const Problem&);
};

struct IterationSummary {
    int iteration = 0; bool step_is_valid = false, step_is_nonmonotonic = false, step_is_successful = false;
    double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, relative_decrease = 0, trust_region_radius = 0, eta = 0;
    int linear_solver_iterations = 0;
};
class IterationCallback { public: virtual ~IterationCallback() {} virtual CallbackReturnType operator()(const IterationSummary& summary) = 0; };

class Solver {
public:
    struct Options {
        int max_num_iterations = 50;
        bool minimizer_progress_to_stdout = false;
        LoggingType logging_type = PER_MINIMIZER_ITERATION;
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
        double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
        int max_num_consecutive_invalid_steps = 5;
        double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
        double eta = 1e-1; int min_linear_solver_iterations = 0, max_linear_solver_iterations = 500;
        bool jacobi_scaling = true;
        bool use_nonmonotonic_steps = false;
        int num_threads = 1;
        std::vector<IterationCallback*> callbacks;
    };
    struct Summary {
        TerminationType termination_type = FAILURE;
        double initial_cost = -1, final_cost = -1, fixed_cost = -1;
        std::vector<IterationSummary> iterations;
        int num_parameters_reduced = 0, num_residuals_reduced = 0, num_successful_steps = 0;
        std::string message;
        std::string FullReport() const { char b[256]; std::snprintf(b, sizeof b, "mini-ceres: cost %.9e -> %.9e, %d iterations, termination %d", initial_cost, final_cost, (int)iterations.size(), (int)termination_type); return b; }
        std::string BriefReport() const { return FullReport(); }
        bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE || termination_type == USER_SUCCESS; }
    };
};

// test hooks (ours): a snapshot callback that sees every Problem handed to Solve, an option to return before minimising, and a pinned
// CG iteration count (the same parity knob as the oracle's cg_fixed_iterations)
struct SolveHooks {
    std::function<void(const Problem&)> on_problem;
    bool skip_minimize = false;
    int cg_fixed_iterations = -1;
};
inline SolveHooks& hooks() { static thread_local SolveHooks h; return h; }
inline std::function<void(const Solver::Summary&)>& hooks_summary() { static thread_local std::function<void(const Solver::Summary&)> f; return f; }

namespace mini_internal {

struct Program {                                    // reduced program: constant blocks removed, rows without a free block dropped
    struct Row { const ResidualBlock* rb; int nres; std::vector<int> col; std::vector<int> size; std::vector<int> slot; int row0; std::vector<double> J; /* nres x width */ int width; };
    std::vector<double*> pblock; std::vector<int> pstart, psize; int n = 0, m = 0;
    std::vector<Row> rows; double fixed_cost = 0.0;
    std::vector<double> x;                          // current state of the free parameters

    void build(const Problem& P) {
        std::unordered_map<double*, int> id;
        for (double* p : P.parameter_order()) { const auto& pi = P.info(p); if (pi.constant) continue; id[p] = (int)pblock.size(); pblock.push_back(p); pstart.push_back(n); psize.push_back(pi.size); n += pi.size; }
        x.resize((size_t)n);
        for (size_t b = 0; b < pblock.size(); ++b) for (int i = 0; i < psize[b]; ++i) x[pstart[b] + i] = pblock[b][i];
        for (const ResidualBlock& rb : P.blocks()) {
            Row r; r.rb = &rb; r.nres = rb.cost->num_residuals(); r.width = 0; r.row0 = m;
            for (size_t i = 0; i < rb.params.size(); ++i) { auto it = id.find(rb.params[i]); if (it == id.end()) continue;
                r.slot.push_back((int)i); r.col.push_back(pstart[it->second]); r.size.push_back(psize[it->second]); r.width += psize[it->second]; }
            if (r.slot.empty()) {                   // contributes to the fixed cost only
                std::vector<double> res((size_t)r.nres);
                if (rb.cost->Evaluate(rb.params.data(), res.data(), nullptr)) { double s = 0; for (double v : res) s += v * v; double rho[3] = {s, 1, 0}; if (rb.loss) rb.loss->Evaluate(s, rho); fixed_cost += 0.5 * rho[0]; }
                continue;
            }
            r.J.assign((size_t)r.nres * r.width, 0.0); m += r.nres; rows.push_back(std::move(r));
        }
    }
    // residuals / Jacobian at xs (free parameters), loss-corrected (corrector.cc with rho'' <= 0: both scaled by sqrt(rho'))
    bool evaluate(const std::vector<double>& xs, double* cost, std::vector<double>* res, bool with_jacobian) {
        for (size_t b = 0; b < pblock.size(); ++b) for (int i = 0; i < psize[b]; ++i) pblock[b][i] = xs[pstart[b] + i];
        res->assign((size_t)m, 0.0);
        std::vector<double> costs(rows.size(), 0.0); bool ok = true;
#pragma omp parallel for schedule(dynamic, 64)
        for (long ri = 0; ri < (long)rows.size(); ++ri) {
            Row& r = rows[ri]; const ResidualBlock& rb = *r.rb; const size_t np = rb.params.size();
            std::vector<double*> jac(np, nullptr); std::vector<std::vector<double>> store;
            if (with_jacobian) { store.resize(r.slot.size()); for (size_t k = 0; k < r.slot.size(); ++k) { store[k].assign((size_t)r.nres * r.size[k], 0.0); jac[r.slot[k]] = store[k].data(); } }
            double* rr = res->data() + r.row0;
            if (!rb.cost->Evaluate(rb.params.data(), rr, with_jacobian ? jac.data() : nullptr)) { ok = false; continue; }
            double s = 0; for (int i = 0; i < r.nres; ++i) s += rr[i] * rr[i];
            double rho[3] = {s, 1.0, 0.0}; if (rb.loss) rb.loss->Evaluate(s, rho);
            costs[ri] = 0.5 * rho[0];
            const double sq = std::sqrt(rho[1]);
            for (int i = 0; i < r.nres; ++i) rr[i] *= sq;
            if (with_jacobian) { int o = 0; for (size_t k = 0; k < r.slot.size(); ++k) { for (int i = 0; i < r.nres; ++i) for (int c = 0; c < r.size[k]; ++c) r.J[(size_t)i * r.width + o + c] = sq * store[k][(size_t)i * r.size[k] + c]; o += r.size[k]; } }
        }
        double c = 0; for (double v : costs) c += v;              // serial sum: independent of the thread count
        *cost = c; return ok;
    }
    void restore() { for (size_t b = 0; b < pblock.size(); ++b) for (int i = 0; i < psize[b]; ++i) pblock[b][i] = x[pstart[b] + i]; }
    template <class F> void for_entries(F f) const {              // f(row index, column, value)
        for (const Row& r : rows) for (int i = 0; i < r.nres; ++i) { int o = 0; for (size_t k = 0; k < r.col.size(); ++k) { for (int c = 0; c < r.size[k]; ++c) f(r.row0 + i, r.col[k] + c, r.J[(size_t)i * r.width + o + c]); o += r.size[k]; } }
    }
    void scale_columns(const std::vector<double>& s) { for (Row& r : rows) for (int i = 0; i < r.nres; ++i) { int o = 0; for (size_t k = 0; k < r.col.size(); ++k) { for (int c = 0; c < r.size[k]; ++c) r.J[(size_t)i * r.width + o + c] *= s[r.col[k] + c]; o += r.size[k]; } } }
    void right_multiply(const double* v, double* y) const { for (int i = 0; i < m; ++i) y[i] = 0; for_entries([&](int r, int c, double a) { y[r] += a * v[c]; }); }   // y = J v
    void left_multiply(const double* y, double* v) const { for_entries([&](int r, int c, double a) { v[c] += a * y[r]; }); }                                       // v += J^T y
    void squared_column_norm(double* out) const { for (int i = 0; i < n; ++i) out[i] = 0; for_entries([&](int, int c, double a) { out[c] += a * a; }); }
};

inline bool invert_spd(int n, const std::vector<double>& a, std::vector<double>& inv) {     // LLT solve against the identity
    std::vector<double> L((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double s = a[(size_t)j * n + j]; for (int k = 0; k < j; ++k) s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(s > 0.0)) return false;
        L[(size_t)j * n + j] = std::sqrt(s);
        for (int i = j + 1; i < n; ++i) { double t = a[(size_t)i * n + j]; for (int k = 0; k < j; ++k) t -= L[(size_t)i * n + k] * L[(size_t)j * n + k]; L[(size_t)i * n + j] = t / L[(size_t)j * n + j]; }
    }
    inv.assign((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c) {
        std::vector<double> y((size_t)n), z((size_t)n);
        for (int i = 0; i < n; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * y[k]; y[i] = s / L[(size_t)i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * z[k]; z[i] = s / L[(size_t)i * n + i]; }
        for (int i = 0; i < n; ++i) inv[(size_t)i * n + c] = z[i];
    }
    return true;
}

// block_jacobi_preconditioner.cc: M = blockdiag(J^T J) + D^2 over the parameter blocks, inverted block by block
struct BlockJacobi {
    const Program* P; std::vector<std::vector<double>> inv;
    void update(const Program& prog, const double* D) {
        P = &prog; const size_t nb = prog.pblock.size();
        std::vector<std::vector<double>> M(nb);
        std::unordered_map<int, int> block_of_col;
        for (size_t b = 0; b < nb; ++b) { M[b].assign((size_t)prog.psize[b] * prog.psize[b], 0.0); block_of_col[prog.pstart[b]] = (int)b; }
        for (const Program::Row& r : prog.rows) for (int i = 0; i < r.nres; ++i) { int o = 0;
            for (size_t k = 0; k < r.col.size(); ++k) { const int b = block_of_col[r.col[k]], s = r.size[k];
                for (int c1 = 0; c1 < s; ++c1) for (int c2 = 0; c2 < s; ++c2) M[b][(size_t)c1 * s + c2] += r.J[(size_t)i * r.width + o + c1] * r.J[(size_t)i * r.width + o + c2];
                o += s; } }
        inv.resize(nb);
        for (size_t b = 0; b < nb; ++b) { const int s = prog.psize[b]; for (int i = 0; i < s; ++i) M[b][(size_t)i * s + i] += D[prog.pstart[b] + i] * D[prog.pstart[b] + i];
            if (s == 1) inv[b].assign(1, 1.0 / M[b][0]); else invert_spd(s, M[b], inv[b]); }
    }
    void apply(const double* r, double* z) const {
        for (size_t b = 0; b < inv.size(); ++b) { const int s = P->psize[b], o = P->pstart[b];
            for (int i = 0; i < s; ++i) { double t = 0; for (int j = 0; j < s; ++j) t += inv[b][(size_t)i * s + j] * r[o + j]; z[o + i] = t; } }
    }
};

inline bool zero_or_inf(double v) { return v == 0.0 || std::isinf(v); }

// cgnr_solver.cc + conjugate_gradients_solver.cc: (J^T J + D^2) x = J^T b, x0 = 0, q_tolerance = eta, r_tolerance disabled
inline int cgnr(const Program& P, const std::vector<double>& b, const std::vector<double>& D, const BlockJacobi& M, const Solver::Options& o, int fixed_iterations, std::vector<double>& x) {
    const int n = P.n, m = P.m;
    std::vector<double> rhs((size_t)n, 0.0), r((size_t)n), p((size_t)n, 0.0), z((size_t)n), tmp((size_t)n), t((size_t)m);
    P.left_multiply(b.data(), rhs.data());
    x.assign((size_t)n, 0.0);
    double nb = 0; for (double v : rhs) nb += v * v;
    if (std::sqrt(nb) == 0.0) return 0;
    auto lhs = [&](const double* v, double* y) { P.right_multiply(v, t.data()); for (int i = 0; i < n; ++i) y[i] = 0.0; P.left_multiply(t.data(), y); for (int i = 0; i < n; ++i) y[i] += D[i] * D[i] * v[i]; };
    lhs(x.data(), tmp.data());
    for (int i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i];
    double rho = 1.0, Q0 = 0.0; for (int i = 0; i < n; ++i) Q0 += x[i] * (rhs[i] + r[i]); Q0 = -1.0 * Q0;
    int it = 1;
    for (;; ++it) {
        M.apply(r.data(), z.data());
        const double last_rho = rho; rho = 0; for (int i = 0; i < n; ++i) rho += r[i] * z[i];
        if (zero_or_inf(rho)) break;
        if (it == 1) p = z; else { const double beta = rho / last_rho; if (zero_or_inf(beta)) break; for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i]; }
        std::vector<double>& q = z; lhs(p.data(), q.data());
        double pq = 0; for (int i = 0; i < n; ++i) pq += p[i] * q[i];
        if (pq <= 0 || std::isinf(pq)) break;
        const double alpha = rho / pq; if (std::isinf(alpha)) break;
        for (int i = 0; i < n; ++i) x[i] = x[i] + alpha * p[i];
        if (it % 10 == 0) { lhs(x.data(), tmp.data()); for (int i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i]; } else for (int i = 0; i < n; ++i) r[i] = r[i] - alpha * q[i];
        double Q1 = 0; for (int i = 0; i < n; ++i) Q1 += x[i] * (rhs[i] + r[i]); Q1 = -1.0 * Q1;
        if (fixed_iterations >= 0) { if (it >= fixed_iterations) break; Q0 = Q1; continue; }
        const double zeta = it * (Q1 - Q0) / Q1;
        if (zeta < o.eta && it >= o.min_linear_solver_iterations) break;
        Q0 = Q1;
        if (it >= o.max_linear_solver_iterations) break;
    }
    return it;
}

}  // namespace mini_internal

// trust_region_minimizer.cc + levenberg_marquardt_strategy.cc (2.1.0), monotonic steps, no bounds, no inner iterations
inline void SolveImpl(const Solver::Options& opt, Problem* problem, Solver::Summary* sum);
inline void Solve(const Solver::Options& opt, Problem* problem, Solver::Summary* sum) {
    if (hooks().on_problem) hooks().on_problem(*problem);
    SolveImpl(opt, problem, sum);
    if (hooks_summary()) hooks_summary()(*sum);
}
inline void SolveImpl(const Solver::Options& opt, Problem* problem, Solver::Summary* sum) {
    using namespace mini_internal;
    *sum = Solver::Summary();
    if (hooks().skip_minimize) { sum->termination_type = USER_SUCCESS; sum->initial_cost = sum->final_cost = 0; sum->message = "skipped"; return; }
    Program P; P.build(*problem);
    sum->fixed_cost = P.fixed_cost; sum->num_parameters_reduced = P.n; sum->num_residuals_reduced = P.m;
    if (P.n == 0 || P.m == 0) { sum->termination_type = CONVERGENCE; sum->initial_cost = sum->final_cost = P.fixed_cost; sum->message = "no free parameters"; return; }
    const int n = P.n, m = P.m;
    std::vector<double> res, cand_res, grad((size_t)n, 0.0), scale((size_t)n, 1.0);
    double cost = 0;
    if (!P.evaluate(P.x, &cost, &res, true)) { P.restore(); sum->termination_type = FAILURE; sum->message = "initial evaluation failed"; return; }
    P.left_multiply(res.data(), grad.data());                      // gradient of the unscaled problem
    if (opt.jacobi_scaling) { P.squared_column_norm(scale.data()); for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(scale[i])); P.scale_columns(scale); }
    sum->initial_cost = cost + P.fixed_cost; sum->final_cost = sum->initial_cost;
    double radius = opt.initial_trust_region_radius, decrease_factor = 2.0; bool reuse_diagonal = false;
    double gmax = 0; for (double g : grad) gmax = std::max(gmax, std::fabs(g));
    double xnorm = 0; for (double v : P.x) xnorm += v * v; xnorm = std::sqrt(xnorm);
    IterationSummary is; is.iteration = 0; is.cost = cost + P.fixed_cost; is.gradient_max_norm = gmax; is.trust_region_radius = radius; is.step_is_valid = false;
    sum->iterations.push_back(is);
    if (gmax <= opt.gradient_tolerance) { sum->termination_type = CONVERGENCE; sum->message = "gradient tolerance"; P.restore(); return; }
    auto run_callbacks = [&](const IterationSummary& s) -> int { for (IterationCallback* cb : opt.callbacks) { const CallbackReturnType r = (*cb)(s); if (r == SOLVER_TERMINATE_SUCCESSFULLY) return 1; if (r == SOLVER_ABORT) return 2; } return 0; };
    { const int r = run_callbacks(is); if (r == 1) { sum->termination_type = USER_SUCCESS; P.restore(); return; } if (r == 2) { sum->termination_type = USER_FAILURE; P.restore(); return; } }

    std::vector<double> diag((size_t)n), D((size_t)n), step((size_t)n), delta((size_t)n), xc((size_t)n), Jstep((size_t)m);
    BlockJacobi M; int invalid = 0, iteration = 0;
    sum->termination_type = NO_CONVERGENCE;
    while (true) {
        if (iteration >= opt.max_num_iterations) { sum->termination_type = NO_CONVERGENCE; sum->message = "max iterations"; break; }
        if (gmax <= opt.gradient_tolerance) { sum->termination_type = CONVERGENCE; sum->message = "gradient tolerance"; break; }
        if (radius < opt.min_trust_region_radius) { sum->termination_type = CONVERGENCE; sum->message = "min trust region radius"; break; }
        ++iteration;
        is = IterationSummary(); is.iteration = iteration; is.cost = cost + P.fixed_cost;
        if (!reuse_diagonal) { P.squared_column_norm(diag.data()); for (int i = 0; i < n; ++i) diag[i] = std::min(std::max(diag[i], opt.min_lm_diagonal), opt.max_lm_diagonal); }
        for (int i = 0; i < n; ++i) D[i] = std::sqrt(diag[i] / radius);
        M.update(P, D.data());
        is.linear_solver_iterations = cgnr(P, res, D, M, opt, hooks().cg_fixed_iterations, step);
        bool finite = true; for (int i = 0; i < n; ++i) { step[i] *= -1.0; if (!std::isfinite(step[i])) finite = false; }
        reuse_diagonal = true;
        double model_cost_change = 0;
        if (finite) { P.right_multiply(step.data(), Jstep.data()); for (int r = 0; r < m; ++r) model_cost_change += Jstep[r] * (res[r] + Jstep[r] / 2.0); model_cost_change = -model_cost_change; }
        is.step_is_valid = finite && model_cost_change > 0.0;
        if (!is.step_is_valid) {
            ++invalid; is.trust_region_radius = radius;
            if (invalid >= opt.max_num_consecutive_invalid_steps) { sum->iterations.push_back(is); sum->termination_type = FAILURE; sum->message = "too many invalid steps"; break; }
            radius *= 0.5; reuse_diagonal = false; is.trust_region_radius = radius; sum->iterations.push_back(is);
            { const int r = run_callbacks(is); if (r == 1) { sum->termination_type = USER_SUCCESS; break; } if (r == 2) { sum->termination_type = USER_FAILURE; break; } }
            continue;
        }
        invalid = 0;
        double step_norm = 0;
        for (int i = 0; i < n; ++i) { delta[i] = step[i] * scale[i]; xc[i] = P.x[i] + delta[i]; const double d = P.x[i] - xc[i]; step_norm += d * d; }
        step_norm = std::sqrt(step_norm); is.step_norm = step_norm;
        double cand_cost = 0;
        if (!P.evaluate(xc, &cand_cost, &cand_res, false)) cand_cost = std::numeric_limits<double>::max();
        if (step_norm <= opt.parameter_tolerance * (xnorm + opt.parameter_tolerance)) { sum->termination_type = CONVERGENCE; sum->message = "parameter tolerance"; is.trust_region_radius = radius; sum->iterations.push_back(is); break; }
        const double cost_change = cost - cand_cost;
        if (std::fabs(cost_change) <= opt.function_tolerance * cost) { sum->termination_type = CONVERGENCE; sum->message = "function tolerance"; is.trust_region_radius = radius; sum->iterations.push_back(is); break; }
        const double relative_decrease = cost_change / model_cost_change; is.relative_decrease = relative_decrease;
        if (relative_decrease > opt.min_relative_decrease) {
            P.x = xc; xnorm = 0; for (double v : P.x) xnorm += v * v; xnorm = std::sqrt(xnorm);
            P.evaluate(P.x, &cost, &res, true);
            std::fill(grad.begin(), grad.end(), 0.0); P.left_multiply(res.data(), grad.data());
            P.scale_columns(scale);
            gmax = 0; for (double g : grad) gmax = std::max(gmax, std::fabs(g));
            is.step_is_successful = true; is.cost = cost + P.fixed_cost; is.cost_change = cost_change; is.gradient_max_norm = gmax;
            radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3)); radius = std::min(opt.max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = false; ++sum->num_successful_steps; sum->final_cost = cost + P.fixed_cost;
        } else {
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
        }
        is.trust_region_radius = radius; sum->iterations.push_back(is);
        { const int r = run_callbacks(is); if (r == 1) { sum->termination_type = USER_SUCCESS; break; } if (r == 2) { sum->termination_type = USER_FAILURE; break; } }
    }
    P.restore();                                                   // user parameters hold the best (= last accepted) point
}

}  // namespace ceres
