// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// CPU restatement of the reference's joint optimisation:
//   refinement/optimizer.cpp:109-361   Optimizer::optimize / addVoxelResiduals / buildProblem / fixVoxelParams
//   refinement/nls_solver.cpp:172-394  addResidual / buildProblem / normalizeCostTermWeights / solve
//   sdf/colorization.cpp:113-370       add / compute / collectObservations / computeObservation / weights / filter
//   sdf/operators.cpp:58-77,142-147    computeSurfaceNormal / sdfToWeight
//   refinement/{volumetric,surface_stab,albedo}_regularizer.{h,cpp}
//   math.cpp:43-47,151-163             robustKernel / poseVecAAToMat
// PARITY UNPINNED: the reference has no tests/golden vectors and cannot be built here
// (needs Ceres, Eigen, OpenCV, Boost — none present); see DESIGN.md.
#pragma once
#include <chrono>
#include <cstring>
#include <map>
#include "grid.hpp"
#include "imaging.hpp"
#include "residuals.hpp"
#include "ceres_like.hpp"

namespace orc {

struct OptConfig {
    int iterations = 10, lm_steps = 50;
    double lambda_g = 0.2, lambda_r0 = 20.0, lambda_r1 = 160.0, lambda_s0 = 10.0, lambda_s1 = 120.0, lambda_a = 0.1;
    int fix_poses = 0, fix_intrinsics = 0, fix_distortion = 0, fix_sdf = 0;
    int carry_trust_radius = 0;        // extension mirrored from the product (what nls_solver.cpp:322-323 intends; dead code in the reference)
    float occlusion_distance = 0.02f; int num_observations = 5;
    double thres_shell = 0.0; int grid_level = 0, rgbd_level = 0;
    int cg_fixed_iterations = -1;      // parity pinning, -1 = native Ceres stopping rule
    int verbose = 0;
};

struct CameraIO { double intr[4]; double dist[5]; std::vector<double> poses; /* 6*K: angle-axis, translation (world->cam) */ };

struct IterStats {
    int rows[4]; double weight_sum[4]; double type_weight[4];
    int valid_voxels; int num_params; int num_rows_reduced;
    double cost_initial, cost_final; int lm_iterations; int successful;
    int cg_iters[50]; int accepted[50]; int n_attempts; double final_radius; int termination;
};

struct Observation { uint8_t color[3] = {0, 0, 0}; float weight = 0.0f; int frame = -1;
    bool operator<(const Observation& o) const { return weight < o.weight; } };

// measurement aids of the CPU baseline (bench.py): threads of the residual collection (default 1 = the reference's behaviour) and the seconds the last
// optimize spent collecting residuals / building + normalising / solving (the reference's time_add / time_build / time_solve, nls_solver.cpp:66-67,101)
inline int& collect_threads_ref() { static int n = 1; return n; }
inline int collect_threads() { return collect_threads_ref(); }
inline double* phase_seconds() { static double t[3] = {0, 0, 0}; return t; }
inline double wall_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline double varying_lambda(int it, int n, double l0, double l1) {            // cost.h:130-143
    if (n <= 1) return l0;
    return l0 + ((l1 - l0) / (double)(n - 1)) * (double)it;
}
inline double sdf_to_weight(double sdf, double trunc) {                       // operators.cpp:142-147
    const double a = std::min(std::abs(sdf), trunc) / trunc;
    return std::min(std::max(1.0 - a, 0.01), 1.0);
}
inline float intensity_u8(const uint8_t c[3]) { return 0.299f * (float)c[0] + 0.587f * (float)c[1] + 0.114f * (float)c[2]; }  // color_util.cpp:41-52

// albedo_regularizer.cpp:59-70: chromaticity weight of an Ea row (the caller drops NaN / Inf)
inline double chroma_weight(const uint8_t ca[3], const uint8_t cb[3]) {
    const float s255 = 1.0f / 255.0f;
    const float lum = intensity_u8(ca), lum_nb = intensity_u8(cb);
    float a[3];
    for (int c = 0; c < 3; ++c) a[c] = ((float)ca[c] * s255) / lum - ((float)cb[c] * s255) / lum_nb;
    float chroma = std::sqrt(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]));          // Vec3f::norm(): halving reduction
    chroma = std::max(1.0f - chroma, 0.01f);
    return (double)chroma * (double)1.0f;
}

// Eigen reduces a FIXED-SIZE 3-vector sum by halving (Core/Redux.h, redux_novec_unroller<0,3>): a0 + (a1 + a2).  Every dot / norm / fixed 3x3 * 3
// product of the reference goes through it; run-time sized blocks (`topLeftCorner(3, 3) * p`) go through GEMV instead, which accumulates
// left to right.  (Eigen is absent from this image: the two orders are restated from its published source, unpinned.)
template <class T> inline T esum3(T a0, T a1, T a2) { return a0 + (a1 + a2); }

// operators.cpp:58-77 (float; Eigen normalize() divides by sqrt(squaredNorm))
inline void surface_normal(const Grid<VoxelSBR>& g, const V3i& p, float n[3]) {
    n[0] = n[1] = n[2] = 0.0f;
    const V3i px = {p.x + 1, p.y, p.z}, py = {p.x, p.y + 1, p.z}, pz = {p.x, p.y, p.z + 1};
    if (!g.valid(p) || !g.valid(px) || !g.valid(py) || !g.valid(pz)) return;
    const float s0 = (float)g.voxel(p).sdf_refined;
    n[0] = (float)g.voxel(px).sdf_refined - s0;
    n[1] = (float)g.voxel(py).sdf_refined - s0;
    n[2] = (float)g.voxel(pz).sdf_refined - s0;
    const float sq = esum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
    const float len = std::sqrt(sq);
    if (len != 0.0f) { n[0] /= len; n[1] /= len; n[2] /= len; }
}
inline bool is_zero3(const float n[3]) { return std::fabs(n[0]) <= 1e-5f && std::fabs(n[1]) <= 1e-5f && std::fabs(n[2]) <= 1e-5f; }

// math.cpp:151-163 — Eigen::AngleAxisd(norm, normalized).matrix() in double, then cast<float>() at the caller
inline void pose_aa_to_mat(const double p[6], double R[9], double t[3]) {
    const double n2 = esum3(p[0] * p[0], p[1] * p[1], p[2] * p[2]);
    const double angle = std::sqrt(n2);
    double ax[3] = {p[0], p[1], p[2]};
    if (n2 > 0.0) { ax[0] /= angle; ax[1] /= angle; ax[2] /= angle; }
    const double s = std::sin(angle), c = std::cos(angle);
    const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const double ca[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
    double tmp;
    tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
    t[0] = p[3]; t[1] = p[4]; t[2] = p[5];
}

struct Colorizer {     // SDFColorization restricted to what the hot path uses
    CameraF cam; float max_occlusion_distance = 0.05f; size_t max_num_observations = 5;

    // colorization.cpp:215-251
    Observation compute_observation(const Grid<VoxelSBR>& g, const V3i& p, const float n[3],
                                    const float R[9], const float t[3], const Image& im) const {
        Observation obs;
        const V3f c = g.voxelToWorld(p); const float sdf = (float)g.voxel(p).sdf_refined;
        const float pt[3] = {c.x - n[0] * sdf, c.y - n[1] * sdf, c.z - n[2] * sdf};
        float q[3];
        for (int i = 0; i < 3; ++i) q[i] = ((R[3 * i] * pt[0] + R[3 * i + 1] * pt[1]) + R[3 * i + 2] * pt[2]) + t[i];
        float p2f[2]; int p2i[2];
        bool ok = cam.project(q, p2f, p2i);
        if (ok) ok = visible(q, im, p2i[0], p2i[1]);
        if (ok) {
            float nc[3];
            for (int i = 0; i < 3; ++i) nc[i] = (R[3 * i] * n[0] + R[3 * i + 1] * n[1]) + R[3 * i + 2] * n[2];
            const float w = weight(im, nc, p2i[0], p2i[1], q);
            if (w > 0.0f) {
                if (im.bgr) { obs.color[0] = bilinear_u8(im.bgr, im.w, im.h, 3, p2f[0], p2f[1], 2);
                              obs.color[1] = bilinear_u8(im.bgr, im.w, im.h, 3, p2f[0], p2f[1], 1);
                              obs.color[2] = bilinear_u8(im.bgr, im.w, im.h, 3, p2f[0], p2f[1], 0); }
                obs.weight = w;
            }
        }
        return obs;
    }
    bool visible(const float q[3], const Image& im, int x, int y) const {     // colorization.cpp:254-270
        if (max_occlusion_distance <= 0.0f) return true;
        const float d = im.depth[(size_t)y * im.w + x];
        if (d > 0.0f) { const float sd = d - q[2]; if (std::abs(sd) <= max_occlusion_distance) return true; }
        return false;
    }
    float weight(const Image& im, const float n[3], int x, int y, const float v[3]) const {   // colorization.cpp:274-315
        const float d = im.depth[(size_t)y * im.w + x];
        if (d <= 0.0f) return 0.0f;
        float wn = 0.0f;
        if (!is_zero3(n)) {
            const float vsq = esum3(v[0] * v[0], v[1] * v[1], v[2] * v[2]);
            float vn[3] = {v[0], v[1], v[2]};
            if (vsq > 0.0f) { const float l = std::sqrt(vsq); vn[0] /= l; vn[1] /= l; vn[2] /= l; }
            wn = 1.0f - std::abs(esum3(vn[0] * n[0], vn[1] * n[1], vn[2] * n[2]));
            wn = std::max(std::min(wn, 1.0f), 0.0f);
            const float div = 1.0f + 2.0f * wn;                      // math.cpp:43-47, thres = 2
            wn = std::max(1.0f / (div * div * div), 0.001f);
        }
        const float dmin = 0.01f, dmax = 5.0f;
        const float dw = std::max(std::min(dmax, d), dmin);
        const float dn = (dw - dmin) / (dmax - dmin);
        float wd = std::max(1.0f - dn, 1.0f);                        // identically 1 (hazard 7)
        wd = std::max(std::min(wd, 5.0f), 0.001f);
        return wn * wd;
    }
    static void mean_color(const std::vector<Observation>& obs, float c[3]) {   // colorization.cpp:318-354 (computeColor, non-empty list)
        c[0] = c[1] = c[2] = 0.0f; float ws = 0.0f; const float sc = 1.0f / 255.0f;
        for (auto& o : obs) { for (int k = 0; k < 3; ++k) c[k] += (float)o.color[k] * (o.weight * sc); ws = ws + o.weight; }
        if (ws > 0.0f) for (int k = 0; k < 3; ++k) c[k] = c[k] * (255.0f / ws);
    }
    static void filter(std::vector<Observation>& obs, size_t n) {    // colorization.cpp:357-370
        const size_t num = obs.size();
        if (n == 0 || n >= num) return;
        std::sort(obs.begin(), obs.end());
        const size_t start = num - n;
        for (size_t i = 0; i < num; ++i) if (i < start) obs[i].weight = 0.0f;
    }
    // colorization.cpp:192-212
    void collect(const Grid<VoxelSBR>& g, const std::vector<double>& poses, const Frames& fr, const V3i& p,
                 const float n[3], int lvl, std::vector<Observation>& out) const {
        const int K = (int)poses.size() / 6;
        out.assign(K, Observation());
        for (int f = 0; f < K; ++f) {
            double Rd[9], td[3]; pose_aa_to_mat(&poses[6 * f], Rd, td);
            float R[9], t[3]; for (int i = 0; i < 9; ++i) R[i] = (float)Rd[i]; for (int i = 0; i < 3; ++i) t[i] = (float)td[i];
            Observation o = compute_observation(g, p, n, R, t, fr.at(f, lvl));
            o.frame = f; out[f] = o;
        }
        filter(out, max_num_observations);
    }
};

// One residual row of the assembled problem
struct Row {
    int type; int v; int f; int dir; double weight; double sdf0;
    int ncols; int cols[P_TOTAL]; ShadingRowConst k;
};

struct Problem {
    Grid<VoxelSBR>* grid = nullptr;
    std::vector<V3i> keys; std::vector<VoxelSBR*> vox;     // visit order
    std::unordered_map<V3i, int, V3iHash> index;
    int N = 0, K = 0;
    std::vector<Row> rows[4];
    std::vector<uint8_t> active, ringok, fix_sdf, fix_alb;
    double weight_sum[4] = {0, 0, 0, 0}, type_weight[4] = {0, 0, 0, 0};
    int valid_voxels = 0;

    void bind(Grid<VoxelSBR>* g, int K_) {
        grid = g; K = K_; N = (int)g->size(); keys.clear(); vox.clear(); index.clear(); index.reserve(N * 2);
        for (auto it = g->data.begin(); it != g->data.end(); ++it) { index[it->first] = (int)keys.size(); keys.push_back(it->first); vox.push_back(&it->second); }
    }
    int id_sdf(const V3i& p) const { return index.find(p)->second; }
    int pose_id(int f, int j) const { return 2 * N + 6 * f + j; }
    int intr_id(int j) const { return 2 * N + 6 * K + j; }
    int dist_id(int j) const { return 2 * N + 6 * K + 4 + j; }
    int num_global() const { return 2 * N + 6 * K + 9; }
};

// Evaluate an Eg row in double (creation-time validity test, shading_cost.cpp:136-145) or with duals.
inline void gather_row_params(const Problem& P, const Row& r, const std::vector<double>& xg, double prm[P_TOTAL]) {
    for (int i = 0; i < r.ncols; ++i) prm[i] = xg[r.cols[i]];
}

// optimizer.cpp:176-282 (one voxel), in two halves.  xg = flat global parameter vector mirroring the grid / camera state.
// First half (:176-241): is the voxel taken, and its Eg rows — reads the grid, the frames and the camera, writes nothing shared: the half that
// collect_rows may run on several threads (orc_set_collect_threads; the reference runs it on one).
inline bool voxel_eg_rows(const Problem& P, const OptConfig& cfg, const Colorizer& col, const Frames& fr,
                          const CameraIO& cam, const std::vector<double>& voxel_sh, int vi, const std::vector<double>& xg, std::vector<Row>& eg_out) {
    const Grid<VoxelSBR>& g = *P.grid; const V3i p = P.keys[vi];
    eg_out.clear();
    if (!g.valid(p)) return false;
    const VoxelSBR& v = *P.vox[vi];
    if (std::abs(v.sdf_refined) > cfg.thres_shell) return false;
    float n[3]; surface_normal(g, p, n);
    if (is_zero3(n)) return false;
    const double weight_sdf = sdf_to_weight(v.sdf_refined, (double)g.truncation);
    std::vector<Observation> obs;
    col.collect(g, cam.poses, fr, p, n, cfg.rgbd_level, obs);

    std::vector<Row> eg;
    for (size_t i = 0; i < obs.size(); ++i) {
        if (obs[i].weight <= 0.0f) continue;
        const int f = obs[i].frame;
        // ShadingCost::create (shading_cost.cpp:59-150)
        if (!g.exists({p.x + 2, p.y, p.z}) || !g.exists({p.x, p.y + 2, p.z}) || !g.exists({p.x, p.y, p.z + 2}) ||
            !g.exists({p.x, p.y + 1, p.z + 1}) || !g.exists({p.x + 1, p.y + 1, p.z}) || !g.exists({p.x + 1, p.y, p.z + 1})) continue;
        float n2[3]; surface_normal(g, p, n2);
        if (is_zero3(n2)) continue;
        Row r; r.type = 0; r.v = vi; r.f = f; r.dir = -1; r.sdf0 = 0; r.ncols = P_TOTAL;
        for (int s = 0; s < 10; ++s) r.cols[P_SDF + s] = P.id_sdf({p.x + SDF_OFF[s][0], p.y + SDF_OFF[s][1], p.z + SDF_OFF[s][2]});
        for (int s = 0; s < 4; ++s) r.cols[P_ALB + s] = P.N + P.id_sdf({p.x + ALB_OFF[s][0], p.y + ALB_OFF[s][1], p.z + ALB_OFF[s][2]});
        for (int j = 0; j < 6; ++j) r.cols[P_POSE + j] = P.pose_id(f, j);
        for (int j = 0; j < 4; ++j) r.cols[P_INTR + j] = P.intr_id(j);
        for (int j = 0; j < 5; ++j) r.cols[P_DIST + j] = P.dist_id(j);
        r.k.vx = p.x; r.k.vy = p.y; r.k.vz = p.z;
        for (int j = 0; j < 9; ++j) r.k.sh[j] = voxel_sh[(size_t)vi * 9 + j];
        r.k.pyr_scale = 1.0 / std::pow(2.0, cfg.rgbd_level);
        r.k.voxel_size = (double)g.voxel_size;
        const Image& im = fr.at(f, cfg.rgbd_level); r.k.w = im.w; r.k.h = im.h; r.k.lum = im.lum;
        double prm[P_TOTAL]; gather_row_params(P, r, xg, prm);
        const double res = shading_residual<double>(r.k, prm);
        if (res == 0.0) continue;                                // NV_INVALID_RESIDUAL
        r.weight = (double)obs[i].weight;
        eg.push_back(r);
    }
    for (auto& r : eg) { r.weight *= weight_sdf; if (r.weight != 0.0) eg_out.push_back(r); }
    return true;
}
// Second half (:243-282), in visit order: the rows into the problem, the regularisers, the visit-order dependent Ea edge rule.
inline void add_voxel_rows(Problem& P, const OptConfig& cfg, int vi, const std::vector<Row>& eg, std::unordered_set<V3i, V3iHash>& voxels_added) {
    Grid<VoxelSBR>& g = *P.grid; const V3i p = P.keys[vi]; const VoxelSBR& v = *P.vox[vi];
    P.active[vi] = 1;
    for (const auto& r : eg) P.rows[0].push_back(r);

    V3i nb[6]; ring6(p, nb);
    const bool ring_ok = ring_valid(g, p);
    if (cfg.lambda_r0 > 0.0 && cfg.lambda_r1 > 0.0 && ring_ok) {      // volumetric_regularizer.cpp:52-78
        Row r; r.type = 1; r.v = vi; r.f = -1; r.dir = -1; r.weight = 1.0; r.sdf0 = 0; r.ncols = 7;
        r.cols[0] = vi; for (int i = 0; i < 6; ++i) r.cols[1 + i] = P.id_sdf(nb[i]);
        P.rows[1].push_back(r);
    }
    if (cfg.lambda_s0 > 0.0 && cfg.lambda_s1 > 0.0) {                  // surface_stab_regularizer.cpp:51-61
        Row r; r.type = 2; r.v = vi; r.f = -1; r.dir = -1; r.weight = 1.0; r.sdf0 = v.sdf; r.ncols = 1; r.cols[0] = vi;
        P.rows[2].push_back(r);
    }
    if (cfg.lambda_a > 0.0 && ring_ok) {                                // optimizer.cpp:259-276
        for (int d = 0; d < 6; ++d) {
            if (voxels_added.find(nb[d]) != voxels_added.end()) continue;
            // albedo_regularizer.cpp:50-84 (both voxels are valid here)
            const VoxelSBR& vn = g.voxel(nb[d]);
            const double w = chroma_weight(v.color, vn.color);
            if (std::isnan(w) || std::isinf(w)) continue;
            if (w == 0.0) continue;
            Row r; r.type = 3; r.v = vi; r.f = -1; r.dir = d; r.weight = w; r.sdf0 = 0; r.ncols = 2;
            r.cols[0] = P.N + vi; r.cols[1] = P.N + P.id_sdf(nb[d]);
            P.rows[3].push_back(r);
        }
    }
    voxels_added.insert(p);
}

// optimizer.cpp:312-361
inline void compute_fixed_flags(Problem& P, const OptConfig& cfg) {
    Grid<VoxelSBR>& g = *P.grid;
    for (int i = 0; i < P.N; ++i) {
        bool fs = false, fa = false;
        if (!g.valid(P.keys[i]) || std::abs(P.vox[i]->sdf_refined) > cfg.thres_shell) { fs = fa = true; }
        if (cfg.lambda_a < 0.0) fa = true;
        const bool rok = ring_valid(g, P.keys[i]);
        P.ringok[i] = rok;
        if (!rok) { fs = fa = true; }
        if (cfg.fix_sdf) fs = true;
        P.fix_sdf[i] = fs; P.fix_alb[i] = fa;
    }
}

inline void collect_rows(Problem& P, const OptConfig& cfg, const Frames& fr, const CameraIO& cam,
                         const std::vector<double>& voxel_sh, std::vector<double>& xg) {
    Grid<VoxelSBR>& g = *P.grid;
    Colorizer col;
    const Image& l0 = fr.at(0, cfg.rgbd_level);
    const double sc = 1.0 / std::pow(2.0, cfg.rgbd_level);                   // optimizer.cpp:124-127
    col.cam.fx = (float)(cam.intr[0] * sc); col.cam.fy = (float)(cam.intr[1] * sc);
    col.cam.cx = (float)(cam.intr[2] * sc); col.cam.cy = (float)(cam.intr[3] * sc);
    for (int i = 0; i < 5; ++i) col.cam.k[i] = (float)cam.dist[i];
    col.cam.w = l0.w; col.cam.h = l0.h;
    col.max_occlusion_distance = cfg.occlusion_distance; col.max_num_observations = (size_t)cfg.num_observations;

    xg.assign(P.num_global(), 0.0);
    for (int i = 0; i < P.N; ++i) { xg[i] = P.vox[i]->sdf_refined; xg[P.N + i] = P.vox[i]->albedo; }
    for (int i = 0; i < 6 * P.K; ++i) xg[2 * P.N + i] = cam.poses[i];
    for (int i = 0; i < 4; ++i) xg[P.intr_id(i)] = cam.intr[i];
    for (int i = 0; i < 5; ++i) xg[P.dist_id(i)] = cam.dist[i];

    for (int t = 0; t < 4; ++t) P.rows[t].clear();
    P.active.assign(P.N, 0); P.ringok.assign(P.N, 0); P.fix_sdf.assign(P.N, 0); P.fix_alb.assign(P.N, 0);
    std::unordered_set<V3i, V3iHash> voxels_added;
    P.valid_voxels = 0;
    const double t0 = wall_seconds();
    const int threads = collect_threads();
    if (threads <= 1) {                                                  // as in the reference: one thread walks the grid
        std::vector<Row> eg;
        for (int vi = 0; vi < P.N; ++vi)
            if (voxel_eg_rows(P, cfg, col, fr, cam, voxel_sh, vi, xg, eg)) { add_voxel_rows(P, cfg, vi, eg, voxels_added); ++P.valid_voxels; }
    } else {                                                             // the same rows in the same order: first halves of a block of voxels in parallel, second halves in visit order
        const int BLOCK = 1 << 15;
        std::vector<std::vector<Row>> egs(BLOCK); std::vector<uint8_t> taken(BLOCK);
        for (int b0 = 0; b0 < P.N; b0 += BLOCK) {
            const int nb = std::min(BLOCK, P.N - b0);
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
            for (int i = 0; i < nb; ++i) taken[i] = voxel_eg_rows(P, cfg, col, fr, cam, voxel_sh, b0 + i, xg, egs[i]) ? 1 : 0;
            for (int i = 0; i < nb; ++i) if (taken[i]) { add_voxel_rows(P, cfg, b0 + i, egs[i], voxels_added); ++P.valid_voxels; }
        }
    }
    compute_fixed_flags(P, cfg);
    phase_seconds()[0] += wall_seconds() - t0;
    (void)g;
}

// nls_solver.cpp:379-394 + :228-235
inline void normalize_weights(Problem& P, const double lambda[4]) {
    for (int t = 0; t < 4; ++t) {
        double s = 0.0; for (auto& r : P.rows[t]) s += r.weight;
        P.weight_sum[t] = s; P.type_weight[t] = 0.0;
        if (s != 0.0) P.type_weight[t] = (lambda[t] / s) * 1000.0;
        for (auto& r : P.rows[t]) r.weight *= P.type_weight[t];
    }
}

// residual + partials of one row (unscaled); J has r.ncols entries
inline double eval_row(const Row& r, const std::vector<double>& xg, double* Jrow) {
    if (r.type == 0) {
        if (Jrow) {
            Jet<P_TOTAL> prm[P_TOTAL];
            for (int i = 0; i < P_TOTAL; ++i) prm[i] = Jet<P_TOTAL>::var(xg[r.cols[i]], i);
            const Jet<P_TOTAL> res = shading_residual<Jet<P_TOTAL>>(r.k, prm);
            for (int i = 0; i < P_TOTAL; ++i) Jrow[i] = res.v[i];
            return res.a;
        }
        double prm[P_TOTAL]; for (int i = 0; i < P_TOTAL; ++i) prm[i] = xg[r.cols[i]];
        return shading_residual<double>(r.k, prm);
    }
    if (r.type == 1) {       // operators.h:88-106: dxx+dyy+dzz
        const double s = xg[r.cols[0]];
        const double dxx = xg[r.cols[1]] + xg[r.cols[2]] - 2.0 * s, dyy = xg[r.cols[3]] + xg[r.cols[4]] - 2.0 * s, dzz = xg[r.cols[5]] + xg[r.cols[6]] - 2.0 * s;
        if (Jrow) { Jrow[0] = -6.0; for (int i = 1; i < 7; ++i) Jrow[i] = 1.0; }
        return dxx + dyy + dzz;
    }
    if (r.type == 2) {       // surface_stab_regularizer.h:59-66
        double res = xg[r.cols[0]] - r.sdf0;
        if (Jrow) Jrow[0] = 1.0;
        if (res == 0.0) { res = 0.0000001; if (Jrow) Jrow[0] = 0.0; }
        return res;
    }
    if (Jrow) { Jrow[0] = 1.0; Jrow[1] = -1.0; }
    return xg[r.cols[0]] - xg[r.cols[1]];
}

struct Reduced {     // Ceres' reduced program: constant blocks removed, rows without a free block dropped
    std::vector<int> col_of;                 // global param id -> reduced column or -1
    std::vector<int> global_of;              // reduced column -> global id
    std::vector<int> block_start, block_size;
    std::vector<const Row*> rows;
};

inline void block_of(const Problem& P, int gid, int* first, int* size) {
    const int N = P.N, K = P.K;
    if (gid < 2 * N) { *first = gid; *size = 1; }
    else if (gid < 2 * N + 6 * K) { *first = 2 * N + ((gid - 2 * N) / 6) * 6; *size = 6; }
    else if (gid < 2 * N + 6 * K + 4) { *first = 2 * N + 6 * K; *size = 4; }
    else { *first = 2 * N + 6 * K + 4; *size = 5; }
}
inline bool is_fixed(const Problem& P, const OptConfig& cfg, int gid) {
    const int N = P.N, K = P.K;
    if (gid < N) return P.fix_sdf[gid];
    if (gid < 2 * N) return P.fix_alb[gid - N];
    if (gid < 2 * N + 6 * K) return cfg.fix_poses != 0;
    if (gid < 2 * N + 6 * K + 4) return cfg.fix_intrinsics != 0;
    return cfg.fix_distortion != 0;
}

inline void build_reduced(const Problem& P, const OptConfig& cfg, Reduced& R) {
    R.col_of.assign(P.num_global(), -1); R.global_of.clear(); R.block_start.clear(); R.block_size.clear(); R.rows.clear();
    for (int t = 0; t < 4; ++t) for (const Row& r : P.rows[t]) {
        bool any = false;
        for (int i = 0; i < r.ncols; ++i) if (!is_fixed(P, cfg, r.cols[i])) { any = true; break; }
        if (!any) continue;
        R.rows.push_back(&r);
        for (int i = 0; i < r.ncols; ++i) {
            const int gid = r.cols[i];
            if (is_fixed(P, cfg, gid) || R.col_of[gid] >= 0) continue;
            int first, size; block_of(P, gid, &first, &size);
            R.block_start.push_back((int)R.global_of.size()); R.block_size.push_back(size);
            for (int j = 0; j < size; ++j) { R.col_of[first + j] = (int)R.global_of.size(); R.global_of.push_back(first + j); }
        }
    }
}

inline void write_back(Problem& P, CameraIO& cam, const std::vector<double>& xg) {
    for (int i = 0; i < P.N; ++i) { P.vox[i]->sdf_refined = xg[i]; P.vox[i]->albedo = xg[P.N + i]; }
    for (int i = 0; i < 6 * P.K; ++i) cam.poses[i] = xg[2 * P.N + i];
    for (int i = 0; i < 4; ++i) cam.intr[i] = xg[P.intr_id(i)];
    for (int i = 0; i < 5; ++i) cam.dist[i] = xg[P.dist_id(i)];
}

// Solve the assembled problem (NLSSolver::solve, nls_solver.cpp:296-367)
inline LMSummary solve_problem(Problem& P, const OptConfig& cfg, std::vector<double>& xg, IterStats* st, double initial_radius = 1e4) {
    Reduced R; build_reduced(P, cfg, R);
    const int n = (int)R.global_of.size(), m = (int)R.rows.size();
    CRS J; J.rows = m; J.cols = n; J.ptr.assign(m + 1, 0);
    for (int r = 0; r < m; ++r) {
        int c = 0; for (int i = 0; i < R.rows[r]->ncols; ++i) if (R.col_of[R.rows[r]->cols[i]] >= 0) ++c;
        J.ptr[r + 1] = J.ptr[r] + c;
    }
    J.col.resize(J.ptr[m]); J.val.assign(J.ptr[m], 0.0);
    for (int r = 0; r < m; ++r) { int k = J.ptr[r]; for (int i = 0; i < R.rows[r]->ncols; ++i) { const int c = R.col_of[R.rows[r]->cols[i]]; if (c >= 0) J.col[k++] = c; } }
    std::vector<double> x(n); for (int c = 0; c < n; ++c) x[c] = xg[R.global_of[c]];
    std::vector<double> xtmp = xg, costs;
    EvalFn eval = [&](const double* xr, double* cost, std::vector<double>* res, CRS* Jout) -> bool {
        for (int c = 0; c < n; ++c) xtmp[R.global_of[c]] = xr[c];
        res->resize(m); costs.resize(m);
#pragma omp parallel for schedule(dynamic, 256)
        for (int r = 0; r < m; ++r) {
            const Row& row = *R.rows[r]; double Jr[P_TOTAL];
            const double raw = eval_row(row, xtmp, Jout ? Jr : nullptr);
            const double s = std::sqrt(row.weight);          // ScaledLoss(nullptr, w): r,J scaled by sqrt(w)
            (*res)[r] = s * raw; costs[r] = 0.5 * row.weight * raw * raw;
            if (Jout) { int k = Jout->ptr[r]; for (int i = 0; i < row.ncols; ++i) if (R.col_of[row.cols[i]] >= 0) Jout->val[k++] = s * Jr[i]; }
        }
        double cs = 0.0; for (int r = 0; r < m; ++r) cs += costs[r];    // serial: thread-count independent
        *cost = cs; return true;
    };
    LMOptions lo; lo.max_num_iterations = cfg.lm_steps; lo.stop_after_first_successful_step = true; lo.initial_radius = initial_radius;
    lo.cg_fixed_iterations = cfg.cg_fixed_iterations; lo.verbose = cfg.verbose != 0;
    LMSummary s;
    if (n == 0 || m == 0) { s.termination = 1; return s; }
    s = lm_minimize(eval, J, R.block_start, R.block_size, x, lo);
    for (int c = 0; c < n; ++c) xg[R.global_of[c]] = x[c];
    if (st) { st->num_params = n; st->num_rows_reduced = m; }
    return s;
}

// optimizer.cpp:109-173
inline bool optimize(Grid<VoxelSBR>& g, const Frames& fr, CameraIO& cam, const OptConfig& cfg,
                     const std::vector<double>& voxel_sh, std::vector<IterStats>* stats) {
    if (cfg.iterations < 1) return false;
    Problem P; P.bind(&g, fr.K);
    double carried_radius = 1e4;
    for (int itr = 0; itr < cfg.iterations; ++itr) {
        const double lambda[4] = {cfg.lambda_g, varying_lambda(itr, cfg.iterations, cfg.lambda_r0, cfg.lambda_r1),
                                  varying_lambda(itr, cfg.iterations, cfg.lambda_s0, cfg.lambda_s1), cfg.lambda_a};
        std::vector<double> xg;
        collect_rows(P, cfg, fr, cam, voxel_sh, xg);
        IterStats st; std::memset(&st, 0, sizeof(st));
        st.valid_voxels = P.valid_voxels;
        if (P.valid_voxels > 0) {
            const double t1 = wall_seconds();
            normalize_weights(P, lambda);
            LMSummary s = solve_problem(P, cfg, xg, &st, cfg.carry_trust_radius ? carried_radius : 1e4);
            phase_seconds()[2] += wall_seconds() - t1;               // (problem reduction + weight normalisation + LM: time_build + time_solve of the reference)
            if (s.final_radius > 0.0) carried_radius = s.final_radius;
            write_back(P, cam, xg);
            st.cost_initial = s.initial_cost; st.cost_final = s.final_cost; st.lm_iterations = s.iterations;
            st.successful = s.successful_steps; st.final_radius = s.final_radius; st.termination = s.termination;
            st.n_attempts = (int)std::min<size_t>(50, s.cg_iterations.size());
            for (int i = 0; i < st.n_attempts; ++i) { st.cg_iters[i] = s.cg_iterations[i]; st.accepted[i] = s.step_accepted[i]; }
        }
        for (int t = 0; t < 4; ++t) { st.rows[t] = (int)P.rows[t].size(); st.weight_sum[t] = P.weight_sum[t]; st.type_weight[t] = P.type_weight[t]; }
        if (stats) stats->push_back(st);
        if (cfg.verbose) std::printf("[oracle] itr %d rows %d/%d/%d/%d valid %d cost %.9e -> %.9e\n", itr, st.rows[0], st.rows[1], st.rows[2], st.rows[3], st.valid_voxels, st.cost_initial, st.cost_final);
    }
    return true;
}

}  // namespace orc
