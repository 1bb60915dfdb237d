// LightingSVSH::estimate + computeVoxelShCoeffs (lighting/lighting_svsh.cpp:93-110,166-346) on the device.
//
//   device: subvolume keys of every stored voxel -> sort/unique (Subvolumes::generate, subvolumes.cpp:211-237);
//           eligible voxels sorted by subvolume; 10x10 Gram block per subvolume on the fp64 matrix cores (sh_kernels.hip);
//           per-voxel trilinear interpolation of the solved coefficients.
//   host  : the 9S-unknown linear least-squares problem is tiny, so the reference's solver sequence (Ceres 2.1.0 LM + CGNR,
//           <= 50 iterations, default tolerances; lighting_svsh.cpp:325-341) runs here in fp64 on the normal equations:
//           every quantity Ceres derives from the Jacobian of this LINEAR problem (cost, J^T r, J^T J p, column norms, 9x9
//           diagonal blocks) is a function of the Gram blocks and the neighbour pairs.
#include "context.hpp"
#include "../device/sh_kernels.hpp"
#include <rocprim/rocprim.hpp>
#include <functional>
#include <limits>

namespace i3d {

namespace {

struct ShSystem {
    int S = 0;
    std::vector<double> G;                       // [S][100]: rows/cols 0..8 = sum w phi phi^T, col 9 = sum w phi I, [9][9] = sum w I^2
    std::vector<std::pair<int, int>> pairs;      // directed (i, neighbour): every undirected pair appears twice (lighting_svsh.cpp:258-289)
    double data_w = 1.0, reg_w = 0.0;
    std::vector<int> blk_of, sub_of;             // subvolume -> block or -1; block -> subvolume
    int n() const { return 9 * (int)sub_of.size(); }
    const double* g(int s) const { return &G[(size_t)s * 100]; }

    double cost(const std::vector<double>& x) const {
        double c = 0.0;
        for (size_t b = 0; b < sub_of.size(); ++b) {
            const double* Gs = g(sub_of[b]); const double* l = &x[9 * b];
            double q = Gs[99];
            for (int i = 0; i < 9; ++i) { double hl = 0.0; for (int j = 0; j < 9; ++j) hl += Gs[i * 10 + j] * l[j]; q += l[i] * hl - 2.0 * Gs[i * 10 + 9] * l[i]; }
            c += 0.5 * data_w * q;
        }
        for (auto& p : pairs) { const double* a = &x[9 * blk_of[p.first]]; const double* b = &x[9 * blk_of[p.second]];
            double d2 = 0.0; for (int j = 0; j < 9; ++j) { const double d = a[j] - b[j]; d2 += d * d; } c += 0.5 * reg_w * d2; }
        // rows of subvolumes outside the reduced program contribute a constant Ceres never sees; none exist here (every data row touches a block)
        return c;
    }
    void jtj(const std::vector<double>& x, std::vector<double>& y) const {
        y.assign(x.size(), 0.0);
        for (size_t b = 0; b < sub_of.size(); ++b) {
            const double* Gs = g(sub_of[b]);
            for (int i = 0; i < 9; ++i) { double s = 0.0; for (int j = 0; j < 9; ++j) s += Gs[i * 10 + j] * x[9 * b + j]; y[9 * b + i] = data_w * s; }
        }
        for (auto& p : pairs) { const int a = 9 * blk_of[p.first], b = 9 * blk_of[p.second];
            for (int j = 0; j < 9; ++j) { const double d = reg_w * (x[a + j] - x[b + j]); y[a + j] += d; y[b + j] -= d; } }
    }
    void grad(const std::vector<double>& x, std::vector<double>& gr) const {
        jtj(x, gr);
        for (size_t b = 0; b < sub_of.size(); ++b) for (int i = 0; i < 9; ++i) gr[9 * b + i] -= data_w * g(sub_of[b])[i * 10 + 9];
    }
    void block(int b, double* M /*81*/) const {
        const double* Gs = g(sub_of[b]);
        int deg = 0; for (auto& p : pairs) { if (blk_of[p.first] == b) ++deg; if (blk_of[p.second] == b) ++deg; }
        for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) M[i * 9 + j] = data_w * Gs[i * 10 + j] + (i == j ? reg_w * deg : 0.0);
    }
};

bool spd_invert9(const double* m, double* inv) {
    const int n = 9; double L[81];
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = m[i * n + j];
        for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
        if (i == j) { if (!(s > 0.0)) return false; L[i * n + i] = std::sqrt(s); } else L[i * n + j] = s / L[j * n + j];
    }
    for (int c = 0; c < n; ++c) {
        double y[9], x[9];
        for (int i = 0; i < n; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
        for (int i = 0; i < n; ++i) inv[i * n + c] = x[i];
    }
    return true;
}

// Ceres 2.1.0 TrustRegionMinimizer + LM strategy + CGNR/block-Jacobi on a linear least-squares problem given by its normal equations
void solve_lm(const ShSystem& sys, std::vector<double>& x, i3d_sh_stats* st) {
    const int n = sys.n(), nb = (int)sys.sub_of.size();
    x.assign(n, 0.0);
    double cost = sys.cost(x);
    if (st) { st->cost_initial = cost; st->cost_final = cost; st->lm_iterations = 0; st->termination = 0; }
    if (n == 0) { if (st) st->termination = 1; return; }
    std::vector<double> g(n), scale(n), diag0(n), blocks((size_t)nb * 81);
    sys.grad(x, g);
    for (int b = 0; b < nb; ++b) { sys.block(b, &blocks[(size_t)b * 81]); for (int i = 0; i < 9; ++i) diag0[9 * b + i] = blocks[(size_t)b * 81 + i * 10]; }
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(diag0[i]));
    double gmax = 0.0; for (int i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(g[i]));
    if (gmax <= 1e-10) { if (st) st->termination = 1; return; }
    double xnorm = 0.0;
    double radius = 1e4, decrease = 2.0; int invalid = 0;
    std::vector<double> D2(n), Minv((size_t)nb * 81), b(n), y(n), r(n), p(n), z(n), q(n), tmp(n), xc(n), step(n), sv(n);
    auto applyA = [&](const std::vector<double>& v, std::vector<double>& out) {       // S JtJ S v + D2 v
        for (int i = 0; i < n; ++i) sv[i] = scale[i] * v[i];
        sys.jtj(sv, out);
        for (int i = 0; i < n; ++i) out[i] = scale[i] * out[i] + D2[i] * v[i];
    };
    for (int iter = 1; iter <= 50; ++iter) {
        if (radius < 1e-32) { if (st) st->termination = 1; break; }
        if (st) st->lm_iterations = iter;
        for (int i = 0; i < n; ++i) D2[i] = std::min(std::max(diag0[i] * scale[i] * scale[i], 1e-6), 1e32) / radius;
        for (int bi = 0; bi < nb; ++bi) {
            double M[81];
            for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) M[i * 9 + j] = scale[9 * bi + i] * scale[9 * bi + j] * blocks[(size_t)bi * 81 + i * 9 + j] + (i == j ? D2[9 * bi + i] : 0.0);
            if (!spd_invert9(M, &Minv[(size_t)bi * 81])) { for (int k = 0; k < 81; ++k) Minv[(size_t)bi * 81 + k] = 0.0; for (int i = 0; i < 9; ++i) Minv[(size_t)bi * 81 + i * 10] = 1.0 / M[i * 10]; }
        }
        for (int i = 0; i < n; ++i) b[i] = scale[i] * g[i];
        // CGNR
        std::fill(y.begin(), y.end(), 0.0); r = b;
        double rho = 1.0, Q0 = 0.0;
        double nb2 = 0.0; for (int i = 0; i < n; ++i) nb2 += b[i] * b[i];
        if (nb2 > 0.0) for (int it = 1;; ++it) {
            for (int bi = 0; bi < nb; ++bi) for (int i = 0; i < 9; ++i) { double s = 0.0; for (int j = 0; j < 9; ++j) s += Minv[(size_t)bi * 81 + i * 9 + j] * r[9 * bi + j]; z[9 * bi + i] = s; }
            const double last = rho; rho = 0.0; for (int i = 0; i < n; ++i) rho += r[i] * z[i];
            if (rho == 0.0 || !std::isfinite(rho)) break;
            if (it == 1) p = z; else { const double beta = rho / last; if (beta == 0.0 || !std::isfinite(beta)) break; for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i]; }
            applyA(p, q);
            double pq = 0.0; for (int i = 0; i < n; ++i) pq += p[i] * q[i];
            if (pq <= 0.0 || std::isinf(pq)) break;
            const double alpha = rho / pq; if (std::isinf(alpha)) break;
            for (int i = 0; i < n; ++i) y[i] += alpha * p[i];
            if (it % 10 == 0) { applyA(y, tmp); for (int i = 0; i < n; ++i) r[i] = b[i] - tmp[i]; } else for (int i = 0; i < n; ++i) r[i] -= alpha * q[i];
            double Q1 = 0.0; for (int i = 0; i < n; ++i) Q1 += y[i] * (b[i] + r[i]); Q1 = -Q1;
            const double zeta = it * (Q1 - Q0) / Q1;
            if (zeta < 0.1) break;
            Q0 = Q1;
            if (it >= 500) break;
        }
        bool finite = true; for (int i = 0; i < n; ++i) { step[i] = -y[i]; if (!std::isfinite(step[i])) finite = false; }
        double model = 0.0;
        if (finite) {       // -(Js s)^T (r + Js s/2) = -s.(S g) - s.(S JtJ S s)/2
            for (int i = 0; i < n; ++i) sv[i] = scale[i] * step[i];
            sys.jtj(sv, tmp);
            double a = 0.0, c2 = 0.0; for (int i = 0; i < n; ++i) { a += step[i] * b[i]; c2 += sv[i] * tmp[i]; }
            model = -a - 0.5 * c2;
        }
        if (!finite || !(model > 0.0)) { if (++invalid > 5) { if (st) st->termination = 3; break; } radius *= 0.5; continue; }
        invalid = 0;
        double sn = 0.0; for (int i = 0; i < n; ++i) { const double d = step[i] * scale[i]; xc[i] = x[i] + d; sn += d * d; }
        sn = std::sqrt(sn);
        const double cand = sys.cost(xc);
        if (sn <= 1e-8 * (xnorm + 1e-8)) { if (st) st->termination = 1; break; }
        const double change = cost - cand;
        if (std::fabs(change) <= 1e-6 * cost) { if (st) st->termination = 1; break; }
        const double rel = change / model;
        if (rel > 1e-3) {
            x = xc; xnorm = 0.0; for (int i = 0; i < n; ++i) xnorm += x[i] * x[i]; xnorm = std::sqrt(xnorm);
            cost = cand; sys.grad(x, g);
            if (st) st->cost_final = cost;
            radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3))); decrease = 2.0;
            gmax = 0.0; for (int i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(g[i]));
            if (gmax <= 1e-10) { if (st) st->termination = 1; break; }
        } else { radius = radius / decrease; decrease *= 2.0; }
    }
}

}  // namespace

int estimate_sh(i3d_context* c, float subvolume_size, double lambda_reg, double thres_shell, int* num_subvolumes, double* sh_out,
                int32_t* sub_index, int cap, i3d_sh_stats* stats) {
    if (!c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_estimate_sh: no grid");
    if (!(thres_shell > 0.0)) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_estimate_sh: thres_shell <= 0 (LightingSVSH::estimate returns false)");
    CTX_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int N = c->N;
    ShParams sp; sp.size = subvolume_size; sp.thres_shell = thres_shell; sp.single = subvolume_size <= 0.0f ? 1 : 0;
    GridView g = c->grid_view();
    TimedScope tscope(c, I3D_K_SH);

    DevBuf<unsigned long long> k0, k1, uniq; DevBuf<int> i0, svox, ssub, counts, nruns; DevBuf<unsigned char> tmp;
    CTX_HIP(c, k0.alloc(N)); CTX_HIP(c, k1.alloc(N)); CTX_HIP(c, uniq.alloc(N)); CTX_HIP(c, i0.alloc(N)); CTX_HIP(c, svox.alloc(N)); CTX_HIP(c, ssub.alloc(N));
    CTX_HIP(c, counts.alloc(N)); CTX_HIP(c, nruns.alloc(1));
    auto sort_keys = [&](bool pairs) -> int {
        size_t bytes = 0;
        if (pairs) CTX_HIP(c, rocprim::radix_sort_pairs(nullptr, bytes, k0.p, k1.p, i0.p, svox.p, (size_t)N, 0, 64, st));
        else CTX_HIP(c, rocprim::radix_sort_keys(nullptr, bytes, k0.p, k1.p, (size_t)N, 0, 64, st));
        CTX_HIP(c, tmp.alloc(bytes ? bytes : 1));
        if (pairs) CTX_HIP(c, rocprim::radix_sort_pairs(tmp.p, bytes, k0.p, k1.p, i0.p, svox.p, (size_t)N, 0, 64, st));
        else CTX_HIP(c, rocprim::radix_sort_keys(tmp.p, bytes, k0.p, k1.p, (size_t)N, 0, 64, st));
        return I3D_OK;
    };
    auto rle = [&](std::vector<unsigned long long>& keys, std::vector<int>& cnt) -> int {
        size_t bytes = 0;
        CTX_HIP(c, rocprim::run_length_encode(nullptr, bytes, k1.p, (unsigned int)N, uniq.p, counts.p, nruns.p, st));
        CTX_HIP(c, tmp.alloc(bytes ? bytes : 1));
        CTX_HIP(c, rocprim::run_length_encode(tmp.p, bytes, k1.p, (unsigned int)N, uniq.p, counts.p, nruns.p, st));
        int runs = 0;
        CTX_HIP(c, hipMemcpyAsync(&runs, nruns.p, sizeof(int), hipMemcpyDeviceToHost, st)); CTX_HIP(c, hipStreamSynchronize(st));
        keys.resize(runs); cnt.resize(runs);
        if (runs) { CTX_HIP(c, hipMemcpy(keys.data(), uniq.p, sizeof(unsigned long long) * runs, hipMemcpyDeviceToHost));
                    CTX_HIP(c, hipMemcpy(cnt.data(), counts.p, sizeof(int) * runs, hipMemcpyDeviceToHost)); }
        return I3D_OK;
    };

    // ---- subvolume set (every stored voxel allocates its subvolume) ----
    std::vector<unsigned long long> sub_keys; std::vector<int> dummy;
    if (sp.single) sub_keys.assign(1, ((1ull << 20)) | ((1ull << 20) << 21) | ((1ull << 20) << 42));
    else { launch_sh_all_keys(st, g, sp, k0.p); int rc = sort_keys(false); if (rc) return rc; rc = rle(sub_keys, dummy); if (rc) return rc; }
    const int S = (int)sub_keys.size();
    if (S == 0) return ctx_fail(c, I3D_ERR_STATE, "i3d_estimate_sh: no subvolumes");
    if (S > cap) return ctx_fail(c, I3D_ERR_CAPACITY, "i3d_estimate_sh: more subvolumes than the caller's capacity");
    DevBuf<unsigned long long> d_sub; CTX_HIP(c, d_sub.alloc(S));
    CTX_HIP(c, hipMemcpyAsync(d_sub.p, sub_keys.data(), sizeof(unsigned long long) * S, hipMemcpyHostToDevice, st));

    // ---- eligible voxels sorted by subvolume, Gram blocks on the matrix cores ----
    launch_sh_keys(st, g, sp, k0.p, i0.p);
    { int rc = sort_keys(true); if (rc) return rc; }
    std::vector<unsigned long long> el_keys; std::vector<int> el_cnt;
    { int rc = rle(el_keys, el_cnt); if (rc) return rc; }
    long long M = 0; for (size_t i = 0; i < el_keys.size(); ++i) if (el_keys[i] != ~0ull) M += el_cnt[i];
    DevBuf<double> d_gram, d_wsum; CTX_HIP(c, d_gram.alloc((size_t)S * 100)); CTX_HIP(c, d_wsum.alloc((size_t)S));      // per-subvolume weight sums (added up on the host in subvolume order)
    CTX_HIP(c, hipMemsetAsync(d_gram.p, 0, sizeof(double) * (size_t)S * 100, st)); CTX_HIP(c, hipMemsetAsync(d_wsum.p, 0, sizeof(double) * (size_t)S, st));
    launch_sh_assign(st, (int)M, k1.p, d_sub.p, S, ssub.p);
    // Sharded (lighting_svsh.cpp:196-253 is the data term): the subvolume-sorted list of eligible voxels is cut into `world` contiguous
    // slices, a rank accumulates the Gram blocks of ITS slice only, and one all-reduce (sum) of S x 100 doubles + the weight sum gives every rank the
    // same totals — the all-reduce hands out one result, so the lighting, and with it every replicated row, is bit-identical on all ranks.
    const int world = (c->comm && (c->comm->world > 1 || c->comm->force)) ? c->comm->world : 1, me = world > 1 ? c->comm->rank : 0;
    const long long m0 = (M * me) / world, m1 = (M * (me + 1)) / world;
    long long longest = 0; for (size_t i = 0; i < el_keys.size(); ++i) if (el_keys[i] != ~0ull) longest = std::max(longest, (long long)el_cnt[i]);
    const int nchunk = sh_gram_chunks(longest);          // (a slice holds at most the whole run of a subvolume)
    const int slab = sh_gram_slab_chunks(S, nchunk);     // chunks per launch: bounds the scratch (<= 128 MB) and the grid
    DevBuf<double> d_part, d_wpart; CTX_HIP(c, d_part.alloc((size_t)S * slab * 100)); CTX_HIP(c, d_wpart.alloc((size_t)S * slab));
    if (m1 > m0) CTX_HIP(c, launch_sh_gram(st, g, (int)m0, (int)m1, S, nchunk, svox.p, ssub.p, d_part.p, d_wpart.p, d_gram.p, d_wsum.p));
    if (c->comm && (c->comm->world > 1 || c->comm->force)) {
        if (c->comm->allreduce_sum(d_gram.p, (size_t)S * 100, st) || c->comm->allreduce_sum(d_wsum.p, (size_t)S, st)) return ctx_fail(c, I3D_ERR_COMM, "i3d_estimate_sh: all-reduce failed");
    }
    ShSystem sys; sys.S = S; sys.G.resize((size_t)S * 100);
    double wsum = 0.0; std::vector<double> wsub((size_t)S, 0.0);
    CTX_HIP(c, hipMemcpyAsync(sys.G.data(), d_gram.p, sizeof(double) * (size_t)S * 100, hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipMemcpyAsync(wsub.data(), d_wsum.p, sizeof(double) * (size_t)S, hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipStreamSynchronize(st));
    for (int q = 0; q < S; ++q) wsum += wsub[(size_t)q];

    // ---- neighbour pairs between subvolumes (6-ring, both directions) ----
    auto unpack = [](unsigned long long k, int& x, int& y, int& z) { x = (int)(k & 0x1fffff) - (1 << 20); y = (int)((k >> 21) & 0x1fffff) - (1 << 20); z = (int)((k >> 42) & 0x1fffff) - (1 << 20); };
    auto pack = [](int x, int y, int z) { const long long B = 1ll << 20; return ((unsigned long long)(x + B) & 0x1fffffull) | (((unsigned long long)(y + B) & 0x1fffffull) << 21) | (((unsigned long long)(z + B) & 0x1fffffull) << 42); };
    if (!sp.single) {       // with a single global volume the reference would hand Ceres duplicate parameter blocks (fatal); no regulariser
        for (int i = 0; i < S; ++i) {
            int x, y, z; unpack(sub_keys[i], x, y, z);
            const int off[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
            for (int d = 0; d < 6; ++d) {
                const unsigned long long nk = pack(x + off[d][0], y + off[d][1], z + off[d][2]);
                auto it = std::lower_bound(sub_keys.begin(), sub_keys.end(), nk);
                if (it != sub_keys.end() && *it == nk) sys.pairs.push_back({i, (int)(it - sub_keys.begin())});
            }
        }
    }
    sys.data_w = wsum > 0.0 ? 1.0 / wsum : 1.0;                                       // lighting_svsh.cpp:298-301
    sys.reg_w = sys.pairs.empty() ? 0.0 : lambda_reg / (double)sys.pairs.size();      // :314
    sys.blk_of.assign(S, -1);
    std::vector<char> touched(S, 0);
    for (size_t i = 0; i < el_keys.size(); ++i) if (el_keys[i] != ~0ull && el_cnt[i] > 0) {
        auto it = std::lower_bound(sub_keys.begin(), sub_keys.end(), el_keys[i]); if (it != sub_keys.end() && *it == el_keys[i]) touched[it - sub_keys.begin()] = 1; }
    for (auto& p : sys.pairs) { touched[p.first] = 1; touched[p.second] = 1; }
    for (int s = 0; s < S; ++s) if (touched[s]) { sys.blk_of[s] = (int)sys.sub_of.size(); sys.sub_of.push_back(s); }
    i3d_sh_stats local; std::memset(&local, 0, sizeof(local));
    i3d_sh_stats* sst = stats ? stats : &local;
    std::memset(sst, 0, sizeof(*sst));
    sst->data_rows = M; sst->reg_rows = (int64_t)sys.pairs.size(); sst->subvolumes = S;
    std::vector<double> x; solve_lm(sys, x, sst);
    std::vector<double> sh((size_t)S * 9, 0.0);
    for (size_t b = 0; b < sys.sub_of.size(); ++b) for (int j = 0; j < 9; ++j) sh[(size_t)sys.sub_of[b] * 9 + j] = x[9 * b + j];

    // ---- per-voxel coefficients (stay resident for i3d_optimize) ----
    DevBuf<double> d_sh; CTX_HIP(c, d_sh.alloc((size_t)S * 9));
    CTX_HIP(c, hipMemcpyAsync(d_sh.p, sh.data(), sizeof(double) * (size_t)S * 9, hipMemcpyHostToDevice, st));
    launch_sh_interpolate(st, g, sp, d_sub.p, S, d_sh.p, c->sh.p);
    CTX_HIP(c, hipStreamSynchronize(st));
    CTX_HIP(c, hipGetLastError());
    c->have_sh = true;
    c->sv_keys = sub_keys; c->sv_sh = sh; c->sv_size = subvolume_size; c->have_subvolumes = true;
    if (num_subvolumes) *num_subvolumes = S;
    if (sh_out) std::memcpy(sh_out, sh.data(), sizeof(double) * (size_t)S * 9);
    if (sub_index) for (int i = 0; i < S; ++i) { int x2, y2, z2; unpack(sub_keys[i], x2, y2, z2); sub_index[3 * i] = x2; sub_index[3 * i + 1] = y2; sub_index[3 * i + 2] = z2; }
    return I3D_OK;
}

}  // namespace i3d

extern "C" int i3d_estimate_sh(i3d_context* c, float subvolume_size, double lambda_reg, double thres_shell, int32_t* num_subvolumes,
                               double* sh, int32_t* sub_index, int32_t cap, i3d_sh_stats* stats) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    return i3d::estimate_sh(c, subvolume_size, lambda_reg, thres_shell, num_subvolumes, sh, sub_index, cap, stats);
}
