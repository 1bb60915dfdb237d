// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// CPU restatement of the spatially-varying SH lighting solve.
//   lighting/subvolumes.cpp:64-95,143-161,164-304   compute / generate / indexToSubvolume / interpolate / pointToIndex
//   lighting/lighting_svsh.cpp:83-110,113-163,166-346  interpolate / computeVoxelShCoeffs / costs / estimate
//   math.cpp:74-96  average
// Solved with the same Ceres-2.1.0-equivalent LM + CGNR (ceres_like.hpp), 50 iterations max, no callback.
// PARITY UNPINNED (no reference tests; reference not buildable here).
#pragma once
#include "optimizer.hpp"

namespace orc {

struct Subvolumes {
    float size = 0.0f, voxel_size = 0.0f;
    std::unordered_map<V3i, int, V3iHash> map;   // default-constructed map, as in the reference (no reserve)
    std::vector<V3i> indices; bool single = false;

    int point_to_index(float pt) const { return (int)std::floor(pt * (1.0f / size)); }         // subvolumes.cpp:263-289
    void compute(const Grid<VoxelSBR>& g) {
        voxel_size = g.voxel_size; map.clear(); indices.clear(); single = false;
        if (size <= 0.0f) { single = true; indices.push_back({0, 0, 0}); map[{0, 0, 0}] = 0; return; }
        for (auto it = g.data.begin(); it != g.data.end(); ++it) {                              // generate(), :211-237
            const V3f c = g.voxelToWorld(it->first);
            const V3i idx = {point_to_index(c.x), point_to_index(c.y), point_to_index(c.z)};
            if (map.find(idx) == map.end()) map[idx] = 0;
        }
        int cnt = 0;
        for (auto it = map.begin(); it != map.end(); ++it) { it->second = cnt; indices.push_back(it->first); ++cnt; }
    }
    int index_to_subvolume(const V3i& idx) const {                                              // :143-161
        if (single) return 0;
        auto f = map.find(idx); return f == map.end() ? -1 : f->second;
    }
    int point_to_subvolume(const V3f& p) const {
        if (single) return 0;
        return index_to_subvolume({point_to_index(p.x), point_to_index(p.y), point_to_index(p.z)});
    }
    size_t count() const { return indices.size(); }

    // interpolate(values, pt, linear=true) :164-205 + math::average; values = 9 doubles per subvolume
    void interpolate(const std::vector<double>& values, const V3f& pt, double out[9]) const {
        for (int j = 0; j < 9; ++j) out[j] = 0.0;
        if (size == 0.0f) {
            // The reference divides by size_ (= 0) here and produces NaN coefficients through inf arithmetic
            // and an undefined float->int cast.  Not reproduced: a single global volume returns its coefficients.
            for (int j = 0; j < 9; ++j) out[j] = values[j];
            return;
        }
        const float inv = 1.0f / size;
        const float pos[3] = {pt.x * inv - 0.5f, pt.y * inv - 0.5f, pt.z * inv - 0.5f};
        V3i coords[8]; float w[8];
        interpolation_weights(pos, coords, w);
        int sub[8];
        for (int i = 0; i < 8; ++i) { sub[i] = index_to_subvolume(coords[i]); if (sub[i] < 0) w[i] = 0.0f; }
        float sum_w = 0.0f;
        for (int i = 0; i < 8; ++i) {
            const float wi = w[i];
            if (wi == 0.0f) continue;
            // Eigen: (float * VectorXd) promotes the scalar to double
            if (sum_w == 0.0f) for (int j = 0; j < 9; ++j) out[j] = (double)wi * values[(size_t)sub[i] * 9 + j];
            else for (int j = 0; j < 9; ++j) out[j] += (double)wi * values[(size_t)sub[i] * 9 + j];
            sum_w += wi;
        }
        if (sum_w != 0.0f) { const double s = (double)(1.0f / sum_w); for (int j = 0; j < 9; ++j) out[j] = out[j] * s; }
    }
};

// shading.h:53-66 basis functions of a unit normal (double)
inline void sh_basis(double nx, double ny, double nz, double b[9]) {
    b[0] = 1.0; b[1] = ny; b[2] = nz; b[3] = nx; b[4] = nx * ny; b[5] = ny * nz; b[6] = (-(nx * nx)) - (ny * ny) + 2.0 * (nz * nz); b[7] = nx * nz; b[8] = (nx * nx) - (ny * ny);
}
// SHDataCost::operator() (lighting_svsh.cpp:127-141): albedo * sum_j l_j H_j(n) - luminance; albedo_basis[j] = albedo * H_j(n) is the row's Jacobian
inline double sh_data_raw(double albedo, const double* albedo_basis, const double* l, double lum) {
    double shd = 0.0;                                   // shading.h:98-112: shad += sh_coeffs[i]*sh_funcs[i]; shading = albedo*shad
    for (int j = 0; j < 9; ++j) shd += l[j] * (albedo_basis[j] / albedo);
    return albedo * shd - lum;
}

struct ShStats { int data_rows, reg_rows, subvolumes, lm_iterations, termination; double cost_initial, cost_final; };

struct Lighting {
    Subvolumes sub; double lambda_reg = 10.0, thres_shell = 0.0; bool weighted = true;
    std::vector<double> sh;      // 9 per subvolume

    // lighting_svsh.cpp:166-346
    bool estimate(const Grid<VoxelSBR>& g, ShStats* st, int cg_fixed_iterations = -1, bool verbose = false) {
        sh.clear();
        if (g.size() == 0 || thres_shell <= 0.0) return false;
        sub.compute(g);
        const int S = (int)sub.count();
        if (S == 0) return false;
        sh.assign((size_t)S * 9, 0.0);
        struct DRow { int s; double w, lum, albedo; float n[3]; };
        std::vector<DRow> drows;
        for (auto it = g.data.begin(); it != g.data.end(); ++it) {
            const V3i& p = it->first; const VoxelSBR& v = it->second;
            if (!g.valid(p)) continue;
            if (std::abs(v.sdf_refined) > thres_shell) continue;
            float n[3]; surface_normal(g, p, n);
            const float nn = std::sqrt(n[0] * n[0] + (n[1] * n[1] + n[2] * n[2]));
            if (is_zero3(n) || std::isnan(nn)) continue;
            if (v.albedo == 0.0 || std::isnan(v.albedo)) continue;
            const int s = sub.point_to_subvolume(g.voxelToWorld(p));
            if (s < 0) continue;
            DRow r; r.s = s; r.lum = (double)(intensity_u8(v.color) / 255.0f); r.albedo = v.albedo;
            r.n[0] = n[0]; r.n[1] = n[1]; r.n[2] = n[2];
            r.w = weighted ? sdf_to_weight(v.sdf_refined, (double)g.truncation) : 1.0;
            drows.push_back(r);
        }
        std::vector<std::pair<int, int>> pairs;       // directed (i, neighbour) — each undirected pair appears twice
        for (int i = 0; i < S; ++i) {
            V3i nb[6]; ring6(sub.indices[i], nb);
            // size <= 0 ("single volume"): the reference's exists() is true for every neighbour index and it would hand
            // Ceres a residual block with the same parameter block twice, which Ceres 2.1.0 rejects with LOG(FATAL)
            // (problem_impl.cc duplicate check) — undefined as a result; the restatement adds no regulariser there.
            if (sub.single) continue;
            for (int j = 0; j < 6; ++j) { const int k = sub.index_to_subvolume(nb[j]); if (k >= 0) pairs.push_back({i, k}); }
        }
        double sum_w = 0.0; for (auto& r : drows) sum_w += r.w;
        const double data_w = sum_w > 0.0 ? 1.0 / sum_w : 1.0;
        const double reg_w = pairs.empty() ? 0.0 : lambda_reg / (double)pairs.size();

        // reduced program: 9-blocks in order of first appearance
        std::vector<int> blk(S, -1), blk_sub; std::vector<int> bstart, bsize;
        auto touch = [&](int s) { if (blk[s] < 0) { blk[s] = (int)blk_sub.size(); blk_sub.push_back(s); bstart.push_back(9 * blk[s]); bsize.push_back(9); } };
        for (auto& r : drows) touch(r.s);
        for (auto& p : pairs) { touch(p.first); touch(p.second); }
        const int n = 9 * (int)blk_sub.size();
        const int m = (int)drows.size() + 9 * (int)pairs.size();
        if (st) { st->data_rows = (int)drows.size(); st->reg_rows = (int)pairs.size(); st->subvolumes = S; }
        if (n == 0 || m == 0) return true;
        CRS J; J.rows = m; J.cols = n; J.ptr.resize(m + 1); J.ptr[0] = 0;
        for (size_t r = 0; r < drows.size(); ++r) J.ptr[r + 1] = J.ptr[r] + 9;
        for (size_t q = 0; q < pairs.size(); ++q) for (int j = 0; j < 9; ++j) { const size_t r = drows.size() + 9 * q + j;
            J.ptr[r + 1] = J.ptr[r] + 2; }
        J.col.resize(J.ptr[m]); J.val.assign(J.ptr[m], 0.0);
        // constant Jacobian (linear problem)
        std::vector<double> basis((size_t)drows.size() * 9);
        for (size_t r = 0; r < drows.size(); ++r) {
            const double nx = (double)drows[r].n[0], ny = (double)drows[r].n[1], nz = (double)drows[r].n[2];
            double b[9]; sh_basis(nx, ny, nz, b);
            const double s = std::sqrt(data_w * drows[r].w);
            for (int j = 0; j < 9; ++j) { basis[r * 9 + j] = drows[r].albedo * b[j]; J.col[J.ptr[r] + j] = 9 * blk[drows[r].s] + j; J.val[J.ptr[r] + j] = s * drows[r].albedo * b[j]; }
        }
        for (size_t q = 0; q < pairs.size(); ++q) for (int j = 0; j < 9; ++j) {
            const size_t r = drows.size() + 9 * q + j; const double s = std::sqrt(reg_w);
            J.col[J.ptr[r]] = 9 * blk[pairs[q].first] + j; J.val[J.ptr[r]] = s; J.col[J.ptr[r] + 1] = 9 * blk[pairs[q].second] + j; J.val[J.ptr[r] + 1] = -s;
        }
        const std::vector<double> Jconst = J.val;
        std::vector<double> x(n, 0.0);
        EvalFn eval = [&](const double* xr, double* cost, std::vector<double>* res, CRS* Jout) -> bool {
            res->resize(m); double cs = 0.0;
            for (size_t r = 0; r < drows.size(); ++r) {
                const double raw = sh_data_raw(drows[r].albedo, &basis[r * 9], xr + 9 * blk[drows[r].s], drows[r].lum); const double w = data_w * drows[r].w;
                (*res)[r] = std::sqrt(w) * raw; cs += 0.5 * w * raw * raw;
            }
            for (size_t q = 0; q < pairs.size(); ++q) for (int j = 0; j < 9; ++j) {
                const double raw = xr[9 * blk[pairs[q].first] + j] - xr[9 * blk[pairs[q].second] + j];
                (*res)[drows.size() + 9 * q + j] = std::sqrt(reg_w) * raw; cs += 0.5 * reg_w * raw * raw;
            }
            if (Jout) Jout->val = Jconst;
            *cost = cs; return true;
        };
        LMOptions lo; lo.max_num_iterations = 50; lo.stop_after_first_successful_step = false; lo.cg_fixed_iterations = cg_fixed_iterations; lo.verbose = verbose;
        LMSummary s = lm_minimize(eval, J, bstart, bsize, x, lo);
        for (size_t b = 0; b < blk_sub.size(); ++b) for (int j = 0; j < 9; ++j) sh[(size_t)blk_sub[b] * 9 + j] = x[9 * b + j];
        if (st) { st->lm_iterations = s.iterations; st->termination = s.termination; st->cost_initial = s.initial_cost; st->cost_final = s.final_cost; }
        return s.termination != 3;
    }

    // lighting_svsh.cpp:93-110 — out: 9 doubles per voxel in visit order; untouched (NaN-marked) when skipped
    void voxel_sh(const Grid<VoxelSBR>& g, std::vector<double>& out, std::vector<uint8_t>* has) const {
        out.assign(g.size() * 9, 0.0); if (has) has->assign(g.size(), 0);
        size_t i = 0;
        for (auto it = g.data.begin(); it != g.data.end(); ++it, ++i) {
            if (!g.valid(it->first) || std::abs(it->second.sdf_refined) > thres_shell) continue;
            sub.interpolate(sh, g.voxelToWorld(it->first), &out[i * 9]);
            if (has) (*has)[i] = 1;
        }
    }
};

}  // namespace orc
