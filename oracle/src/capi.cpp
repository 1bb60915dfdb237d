// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
// extern "C" surface of the CPU restatement (see ../i3d_oracle.h).
#include "../i3d_oracle.h"
#include "lighting.hpp"
#include "fusion.hpp"
#include "mesh.hpp"

using namespace orc;

static OptConfig to_cfg(const orc_opt_config* c) {
    OptConfig o;
    o.iterations = c->iterations; o.lm_steps = c->lm_steps; o.lambda_g = c->lambda_g;
    o.lambda_r0 = c->lambda_r0; o.lambda_r1 = c->lambda_r1; o.lambda_s0 = c->lambda_s0; o.lambda_s1 = c->lambda_s1; o.lambda_a = c->lambda_a;
    o.fix_poses = c->fix_poses; o.fix_intrinsics = c->fix_intrinsics; o.fix_distortion = c->fix_distortion;
    o.occlusion_distance = c->occlusion_distance; o.num_observations = c->num_observations;
    o.thres_shell = c->thres_shell; o.grid_level = c->grid_level; o.rgbd_level = c->rgbd_level;
    o.cg_fixed_iterations = c->cg_fixed_iterations; o.verbose = c->verbose; o.fix_sdf = c->fix_sdf; o.carry_trust_radius = c->carry_trust_radius;
    return o;
}

struct ProblemHandle { Problem P; OptConfig cfg; std::vector<double> xg; };

extern "C" {

void* orc_grid_from_voxels(float voxel_size, int64_t n, const int32_t* keys, const float* sdf, const float* weight, const uint8_t* color) {
    Grid<Voxel> g(voxel_size);                       // SparseVoxelGrid<Voxel>::load order (sparse_voxel_grid.cpp:545-568)
    for (int64_t i = 0; i < n; ++i) {
        Voxel v; v.sdf = sdf[i]; v.weight = weight[i];
        for (int c = 0; c < 3; ++c) v.color[c] = color[3 * i + c];
        g.setVoxel({keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]}, v);
    }
    return convert(g);
}
int64_t orc_grid_size(void* g) { return (int64_t)((Grid<VoxelSBR>*)g)->size(); }
float orc_grid_voxel_size(void* g) { return ((Grid<VoxelSBR>*)g)->voxel_size; }
void orc_grid_export(void* gp, int32_t* keys, double* sdf, double* sdf_refined, double* albedo, float* weight, uint8_t* color) {
    auto* g = (Grid<VoxelSBR>*)gp; size_t i = 0;
    for (auto it = g->data.begin(); it != g->data.end(); ++it, ++i) {
        if (keys) { keys[3 * i] = it->first.x; keys[3 * i + 1] = it->first.y; keys[3 * i + 2] = it->first.z; }
        if (sdf) sdf[i] = it->second.sdf;
        if (sdf_refined) sdf_refined[i] = it->second.sdf_refined;
        if (albedo) albedo[i] = it->second.albedo;
        if (weight) weight[i] = it->second.weight;
        if (color) for (int c = 0; c < 3; ++c) color[3 * i + c] = it->second.color[c];
    }
}
void orc_grid_import(void* gp, const double* sdf_refined, const double* albedo, const uint8_t* color) {
    auto* g = (Grid<VoxelSBR>*)gp; size_t i = 0;
    for (auto it = g->data.begin(); it != g->data.end(); ++it, ++i) {
        if (sdf_refined) it->second.sdf_refined = sdf_refined[i];
        if (albedo) it->second.albedo = albedo[i];
        if (color) for (int c = 0; c < 3; ++c) it->second.color[c] = color[3 * i + c];
    }
}
void orc_grid_clear_outside_shell(void* g, double thres) { clear_voxels_outside_thin_shell(*(Grid<VoxelSBR>*)g, thres); }
void* orc_grid_upsample(void* g) { return upsample(*(Grid<VoxelSBR>*)g); }
void orc_grid_free(void* g) { delete (Grid<VoxelSBR>*)g; }

void* orc_frames_create(int32_t K, int32_t levels) { auto* f = new Frames; f->K = K; f->levels = levels; f->img.resize((size_t)K * levels); return f; }
void orc_frames_set(void* fr, int32_t f, int32_t lvl, int32_t w, int32_t h, const float* lum, const float* depth, const uint8_t* bgr) {
    Image& im = ((Frames*)fr)->img[(size_t)f * ((Frames*)fr)->levels + lvl]; im.w = w; im.h = h; im.lum = lum; im.depth = depth; im.bgr = bgr;
}
void orc_frames_free(void* fr) { delete (Frames*)fr; }

// measurement aids of bench.py's CPU baseline: threads of the residual collection (<= 1: one thread, as in the reference; the rows and their order do not
// depend on it) and the seconds spent collecting residuals [0] / building + solving [2] since the last reset
void orc_set_collect_threads(int32_t n) { collect_threads_ref() = n; }
void orc_phase_seconds(double* out3, int32_t reset) { for (int i = 0; i < 3; ++i) { out3[i] = phase_seconds()[i]; if (reset) phase_seconds()[i] = 0.0; } }

int32_t orc_optimize(void* g, void* fr, const orc_opt_config* c, double* intr, double* dist, double* poses,
                     const double* voxel_sh, orc_iter_stats* stats) {
    auto* G = (Grid<VoxelSBR>*)g; auto* F = (Frames*)fr; OptConfig cfg = to_cfg(c);
    CameraIO cam; for (int i = 0; i < 4; ++i) cam.intr[i] = intr[i]; for (int i = 0; i < 5; ++i) cam.dist[i] = dist[i];
    cam.poses.assign(poses, poses + 6 * F->K);
    std::vector<double> sh(voxel_sh, voxel_sh + G->size() * 9);
    std::vector<IterStats> st;
    const bool ok = optimize(*G, *F, cam, cfg, sh, &st);
    for (int i = 0; i < 4; ++i) intr[i] = cam.intr[i]; for (int i = 0; i < 5; ++i) dist[i] = cam.dist[i];
    for (int i = 0; i < 6 * F->K; ++i) poses[i] = cam.poses[i];
    if (stats) for (size_t i = 0; i < st.size(); ++i) {
        static_assert(sizeof(orc_iter_stats) == sizeof(IterStats), "stats layout");
        std::memcpy(&stats[i], &st[i], sizeof(IterStats));
    }
    return ok ? 0 : 1;
}

void* orc_collect(void* g, void* fr, const orc_opt_config* c, const double* intr, const double* dist, const double* poses,
                  const double* voxel_sh, int32_t iteration) {
    auto* G = (Grid<VoxelSBR>*)g; auto* F = (Frames*)fr;
    auto* h = new ProblemHandle; h->cfg = to_cfg(c);
    CameraIO cam; for (int i = 0; i < 4; ++i) cam.intr[i] = intr[i]; for (int i = 0; i < 5; ++i) cam.dist[i] = dist[i];
    cam.poses.assign(poses, poses + 6 * F->K);
    std::vector<double> sh(voxel_sh, voxel_sh + G->size() * 9);
    h->P.bind(G, F->K);
    collect_rows(h->P, h->cfg, *F, cam, sh, h->xg);
    const OptConfig& q = h->cfg;
    const double lambda[4] = {q.lambda_g, varying_lambda(iteration, q.iterations, q.lambda_r0, q.lambda_r1),
                              varying_lambda(iteration, q.iterations, q.lambda_s0, q.lambda_s1), q.lambda_a};
    normalize_weights(h->P, lambda);
    return h;
}
void orc_problem_counts(void* p, int32_t rows[4], double ws[4], double tw[4]) {
    auto* h = (ProblemHandle*)p;
    for (int t = 0; t < 4; ++t) { rows[t] = (int)h->P.rows[t].size(); if (ws) ws[t] = h->P.weight_sum[t]; if (tw) tw[t] = h->P.type_weight[t]; }
}
void orc_problem_flags(void* p, uint8_t* active, uint8_t* ring_ok, uint8_t* fix_sdf, uint8_t* fix_alb) {
    auto* h = (ProblemHandle*)p;
    for (int i = 0; i < h->P.N; ++i) { if (active) active[i] = h->P.active[i]; if (ring_ok) ring_ok[i] = h->P.ringok[i];
        if (fix_sdf) fix_sdf[i] = h->P.fix_sdf[i]; if (fix_alb) fix_alb[i] = h->P.fix_alb[i]; }
}
void orc_problem_eg(void* p, int32_t* v, int32_t* f, double* weight, double* residual, double* J) {
    auto* h = (ProblemHandle*)p; const auto& rows = h->P.rows[0];
#pragma omp parallel for schedule(dynamic, 256)
    for (size_t i = 0; i < rows.size(); ++i) {
        v[i] = rows[i].v; f[i] = rows[i].f; weight[i] = rows[i].weight;
        double Jr[P_TOTAL]; residual[i] = eval_row(rows[i], h->xg, J ? Jr : nullptr);
        if (J) for (int k = 0; k < P_TOTAL; ++k) J[i * P_TOTAL + k] = Jr[k];
    }
}
void orc_problem_reg(void* p, int32_t type, int32_t* v, int32_t* dir, double* weight, double* residual) {
    auto* h = (ProblemHandle*)p; const auto& rows = h->P.rows[type];
    for (size_t i = 0; i < rows.size(); ++i) { v[i] = rows[i].v; if (dir) dir[i] = rows[i].dir; weight[i] = rows[i].weight;
        if (residual) residual[i] = eval_row(rows[i], h->xg, nullptr); }
}
double orc_problem_normal_eq(void* p, const orc_opt_config* c, double* gradient, double* jtj_diag, int32_t* is_free) {
    auto* h = (ProblemHandle*)p; OptConfig cfg = to_cfg(c); const Problem& P = h->P;
    const int ng = P.num_global();
    for (int i = 0; i < ng; ++i) { if (gradient) gradient[i] = 0.0; if (jtj_diag) jtj_diag[i] = 0.0; if (is_free) is_free[i] = 0; }
    double cost = 0.0;
    for (int t = 0; t < 4; ++t) for (const Row& r : P.rows[t]) {
        bool any = false; for (int i = 0; i < r.ncols; ++i) if (!is_fixed(P, cfg, r.cols[i])) any = true;
        if (!any) continue;
        double Jr[P_TOTAL]; const double raw = eval_row(r, h->xg, Jr);
        cost += 0.5 * r.weight * raw * raw;
        for (int i = 0; i < r.ncols; ++i) { const int gid = r.cols[i]; if (is_fixed(P, cfg, gid)) continue;
            if (is_free) is_free[gid] = 1;
            if (gradient) gradient[gid] += r.weight * Jr[i] * raw;
            if (jtj_diag) jtj_diag[gid] += r.weight * Jr[i] * Jr[i]; }
    }
    return cost;
}
void orc_problem_jtj_apply(void* p, const orc_opt_config* c, const double* x, double* y) {
    auto* h = (ProblemHandle*)p; OptConfig cfg = to_cfg(c); const Problem& P = h->P;
    const int ng = P.num_global(); for (int i = 0; i < ng; ++i) y[i] = 0.0;
    for (int t = 0; t < 4; ++t) for (const Row& r : P.rows[t]) {
        double Jr[P_TOTAL]; eval_row(r, h->xg, Jr);
        double s = 0.0; for (int i = 0; i < r.ncols; ++i) if (!is_fixed(P, cfg, r.cols[i])) s += Jr[i] * x[r.cols[i]];
        s *= r.weight;
        for (int i = 0; i < r.ncols; ++i) if (!is_fixed(P, cfg, r.cols[i])) y[r.cols[i]] += Jr[i] * s;
    }
}
void orc_problem_free(void* p) { delete (ProblemHandle*)p; }

// Diagnostic for chained comparisons: how close the top-n cut of SDFColorization::filter (colorization.cpp:357-370) is to a tie.  For every voxel that
// would get rows at the current state: (w_n - w_{n+1}) / w_n over its positive observation weights sorted descending (1 when there are at most n
// candidates); -1 for voxels without rows.  A second implementation whose state differs by round-off can only pick another keyframe where this is tiny.
void orc_observation_margins(void* g, void* fr, const orc_opt_config* c, const double* intr, const double* dist, const double* poses, double* margin) {
    auto* G = (Grid<VoxelSBR>*)g; auto* F = (Frames*)fr; OptConfig cfg = to_cfg(c);
    Colorizer col; const Image& l0 = F->at(0, cfg.rgbd_level);
    const double sc = 1.0 / std::pow(2.0, cfg.rgbd_level);
    col.cam.fx = (float)(intr[0] * sc); col.cam.fy = (float)(intr[1] * sc); col.cam.cx = (float)(intr[2] * sc); col.cam.cy = (float)(intr[3] * sc);
    for (int i = 0; i < 5; ++i) col.cam.k[i] = (float)dist[i];
    col.cam.w = l0.w; col.cam.h = l0.h; col.max_occlusion_distance = cfg.occlusion_distance; col.max_num_observations = 0;      // 0: keep every observation
    std::vector<double> pv(poses, poses + 6 * F->K);
    std::vector<V3i> keys; for (auto it = G->data.begin(); it != G->data.end(); ++it) keys.push_back(it->first);
    const size_t n = (size_t)cfg.num_observations;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)keys.size(); ++i) {
        margin[i] = -1.0;
        const V3i p = keys[(size_t)i];
        if (!G->valid(p) || std::abs(G->voxel(p).sdf_refined) > cfg.thres_shell) continue;
        float nrm[3]; surface_normal(*G, p, nrm);
        if (is_zero3(nrm)) continue;
        std::vector<Observation> obs; col.collect(*G, pv, *F, p, nrm, cfg.rgbd_level, obs);
        std::vector<float> w; for (auto& o : obs) if (o.weight > 0.0f) w.push_back(o.weight);
        std::sort(w.begin(), w.end(), [](float a, float b) { return a > b; });
        margin[i] = (n == 0 || w.size() <= n) ? 1.0 : (double)(w[n - 1] - w[n]) / (double)w[n - 1];
    }
}

int32_t orc_estimate_sh(void* g, float subvolume_size, double lambda_reg, double thres_shell, int32_t cg_fixed,
                        int32_t* num_subvolumes, double* sh, int32_t* sub_index, int32_t cap,
                        double* voxel_sh, uint8_t* voxel_has, orc_sh_stats* st) {
    auto* G = (Grid<VoxelSBR>*)g;
    Lighting L; L.sub.size = subvolume_size; L.lambda_reg = lambda_reg; L.thres_shell = thres_shell; L.weighted = true;
    ShStats s; std::memset(&s, 0, sizeof(s));
    const bool ok = L.estimate(*G, &s, cg_fixed, false);
    if (st) std::memcpy(st, &s, sizeof(s));
    if (!ok) return 1;
    const int S = (int)L.sub.count(); *num_subvolumes = S;
    if (S > cap) return 2;
    for (int i = 0; i < S; ++i) { for (int j = 0; j < 9; ++j) sh[9 * i + j] = L.sh[9 * (size_t)i + j];
        if (sub_index) { sub_index[3 * i] = L.sub.indices[i].x; sub_index[3 * i + 1] = L.sub.indices[i].y; sub_index[3 * i + 2] = L.sub.indices[i].z; } }
    if (voxel_sh) { std::vector<double> out; std::vector<uint8_t> has; L.voxel_sh(*G, out, &has);
        std::memcpy(voxel_sh, out.data(), out.size() * sizeof(double)); if (voxel_has) std::memcpy(voxel_has, has.data(), has.size()); }
    return 0;
}

// rgbd/pyramid.cpp:59-166 — keyframe pyramids.  cv::cvtColor(BGR2GRAY) on floats and cv::pyrDown are OpenCV (un-vendored): restated from
// their documented kernels — gray = 0.114 b + 0.587 g + 0.299 r; pyrDown = [1 4 6 4 1] x [1 4 6 4 1] / 256, BORDER_REFLECT_101, horizontal
// pass first, size (w/2, h/2).  PARITY UNPINNED against OpenCV's exact float summation order.
static int orc_reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i; return i; }
void orc_lum_from_bgr(int32_t n, const uint8_t* bgr, float* lum) {
    const float s = (float)(1.0 / 255.0);
    for (int i = 0; i < n; ++i) { const float b = (float)bgr[3 * i] * s, g = (float)bgr[3 * i + 1] * s, r = (float)bgr[3 * i + 2] * s; lum[i] = (b * 0.114f + g * 0.587f) + r * 0.299f; }
}
void orc_pyr_down(int32_t w, int32_t h, const float* src, float* dst) {
    const int ow = w / 2, oh = h / 2;
    for (int y = 0; y < oh; ++y) for (int x = 0; x < ow; ++x) {
        float row[5];
        for (int j = 0; j < 5; ++j) {
            const float* line = src + (size_t)orc_reflect101(2 * y - 2 + j, h) * w;
            const float m2 = line[orc_reflect101(2 * x - 2, w)], m1 = line[orc_reflect101(2 * x - 1, w)], c0 = line[orc_reflect101(2 * x, w)],
                        p1 = line[orc_reflect101(2 * x + 1, w)], p2 = line[orc_reflect101(2 * x + 2, w)];
            row[j] = ((c0 * 6.0f + (m1 + p1) * 4.0f) + m2) + p2;
        }
        dst[(size_t)y * ow + x] = (((row[2] * 6.0f + (row[1] + row[3]) * 4.0f) + row[0]) + row[4]) * (1.0f / 256.0f);
    }
}
void orc_depth_down(int32_t w, int32_t h, const float* src, float* dst) {       // Pyramid::downsampleDepth (pyramid.cpp:115-143)
    const int ow = w / 2, oh = h / 2;
    for (int y = 0; y < oh; ++y) for (int x = 0; x < ow; ++x) {
        int cnt = 0; float sum = 0.0f;
        const float d[4] = {src[(size_t)(2 * y) * w + 2 * x], src[(size_t)(2 * y) * w + 2 * x + 1], src[(size_t)(2 * y + 1) * w + 2 * x], src[(size_t)(2 * y + 1) * w + 2 * x + 1]};
        for (int i = 0; i < 4; ++i) if (d[i] > 0.0f) { sum += d[i]; ++cnt; }
        dst[(size_t)y * ow + x] = cnt > 0 ? sum / (float)cnt : 0.0f;
    }
}

// rgbd/processing.cpp:129-181 + interpolate<float> :236-283
void orc_resize_depth(int32_t iw, int32_t ih, const float* din, const float* in_intr, int32_t ow, int32_t oh, const float* out_intr, float* dout) {
    if (iw == ow && ih == oh) { std::memcpy(dout, din, sizeof(float) * (size_t)iw * ih); return; }      // :135-139: same size -> a clone, whatever the intrinsics
    const float in_fx = in_intr[0], in_fy = in_intr[1], in_cx = in_intr[2], in_cy = in_intr[3];
    const float out_cx = out_intr[2], out_cy = out_intr[3], out_fx_inv = 1.0f / out_intr[0], out_fy_inv = 1.0f / out_intr[1];
    for (int y = 0; y < oh; ++y) for (int x = 0; x < ow; ++x) {
        dout[(size_t)y * ow + x] = 0.0f;
        const float x0n = ((float)x - out_cx) * out_fx_inv, y0n = ((float)y - out_cy) * out_fy_inv;
        const float px = (in_fx * x0n / 1.0f) + in_cx, py = (in_fy * y0n / 1.0f) + in_cy;
        const int pxi = (int)(px + 0.5f), pyi = (int)(py + 0.5f);
        if (pxi < 0 || pyi < 0 || pxi >= iw || pyi >= ih) continue;
        const int x0 = (int)std::floor(px), y0 = (int)std::floor(py), x1 = x0 + 1, y1 = y0 + 1;
        float x1w = px - (float)x0, y1w = py - (float)y0, x0w = 1.0f - x1w, y0w = 1.0f - y1w;
        if (x0 < 0 || x0 >= iw) x0w = 0.0f; if (x1 < 0 || x1 >= iw) x1w = 0.0f; if (y0 < 0 || y0 >= ih) y0w = 0.0f; if (y1 < 0 || y1 >= ih) y1w = 0.0f;
        const float w00 = x0w * y0w, w10 = x1w * y0w, w01 = x0w * y1w, w11 = x1w * y1w;
        const float sw = w00 + w10 + w01 + w11;
        float sum = 0.0f;
        if (w00 > 0.0f) sum += din[(size_t)y0 * iw + x0] * w00;
        if (w01 > 0.0f) sum += din[(size_t)y1 * iw + x0] * w01;
        if (w10 > 0.0f) sum += din[(size_t)y0 * iw + x1] * w10;
        if (w11 > 0.0f) sum += din[(size_t)y1 * iw + x1] * w11;
        const float d = sw > 0.0f ? sum / sw : 0.0f;
        if (d == 0.0f) continue;
        dout[(size_t)y * ow + x] = d;
    }
}

// intrinsic3d.cpp:381-409 + colorization.cpp:113-189,318-354 (recolourisation at pyramid level 0)
int32_t orc_recompute_colors(void* g, void* fr, const double* intr, const double* dist, const double* poses,
                             float occlusion_distance, int32_t num_observations) {
    auto* G = (Grid<VoxelSBR>*)g; auto* F = (Frames*)fr;
    Colorizer col; const Image& l0 = F->at(0, 0);
    col.cam.fx = (float)intr[0]; col.cam.fy = (float)intr[1]; col.cam.cx = (float)intr[2]; col.cam.cy = (float)intr[3];
    for (int i = 0; i < 5; ++i) col.cam.k[i] = (float)dist[i];
    col.cam.w = l0.w; col.cam.h = l0.h; col.max_occlusion_distance = occlusion_distance; col.max_num_observations = (size_t)num_observations;
    std::vector<std::vector<Observation>> obs(G->size());
    for (int f = 0; f < F->K; ++f) {
        double Rd[9], td[3]; pose_aa_to_mat(poses + 6 * f, Rd, td);
        float R[9], t[3]; for (int i = 0; i < 9; ++i) R[i] = (float)Rd[i]; for (int i = 0; i < 3; ++i) t[i] = (float)td[i];
        size_t vi = 0;
        for (auto it = G->data.begin(); it != G->data.end(); ++it, ++vi) {
            float n[3]; surface_normal(*G, it->first, n);
            if (is_zero3(n)) continue;
            Observation o = col.compute_observation(*G, it->first, n, R, t, F->at(f, 0));
            if (o.weight > 0.0f) { o.frame = f; obs[vi].push_back(o); }
        }
    }
    size_t vi = 0;
    for (auto it = G->data.begin(); it != G->data.end(); ++it, ++vi) {
        if (obs[vi].empty()) continue;
        if (col.max_num_observations > 0) Colorizer::filter(obs[vi], col.max_num_observations);
        float c[3]; Colorizer::mean_color(obs[vi], c);
        for (int k = 0; k < 3; ++k) it->second.color[k] = (uint8_t)c[k];
    }
    return 0;
}

// intrinsic3d.cpp:206-290 — the double-hierarchical refine schedule over the restated pieces above.
// *grid_io is replaced when a level is upsampled.  The camera arrays are updated in place.
int32_t orc_refine(void** grid_io, void* fr, const orc_opt_config* c, int32_t num_grid_levels, int32_t num_rgbd_levels,
                   double thres_shell_factor, double thres_shell_factor_final, int32_t clear_distant_voxels,
                   float subvolume_size_sh, double sh_lambda_reg, double* intr, double* dist, double* poses, int32_t* levels_done) {
    auto* G = (Grid<VoxelSBR>*)*grid_io; auto* F = (Frames*)fr;
    if (!G || num_grid_levels <= 0 || num_rgbd_levels <= 0) return 1;
    OptConfig cfg = to_cfg(c);
    CameraIO cam; for (int i = 0; i < 4; ++i) cam.intr[i] = intr[i]; for (int i = 0; i < 5; ++i) cam.dist[i] = dist[i];
    cam.poses.assign(poses, poses + 6 * F->K);
    int done = 0;
    orc_recompute_colors(G, F, cam.intr, cam.dist, cam.poses.data(), cfg.occlusion_distance, cfg.num_observations);     // init(): :196-201
    const int coarsest = num_grid_levels - 1;
    for (int gl = coarsest; gl >= 0; --gl) {
        double factor = thres_shell_factor;                                                                               // prepareGridLevel :298-316
        if (thres_shell_factor_final > 0.0) factor = varying_lambda(coarsest - gl, num_grid_levels, thres_shell_factor, thres_shell_factor_final);
        const double thres = factor * (double)G->voxel_size;
        if (clear_distant_voxels) clear_voxels_outside_thin_shell(*G, thres);
        for (int pl = num_rgbd_levels - 1; pl >= 0; --pl) {
            if (pl > 0 && gl < coarsest) continue;
            Lighting L; L.sub.size = subvolume_size_sh; L.lambda_reg = sh_lambda_reg; L.thres_shell = thres; L.weighted = true;
            ShStats ss; std::memset(&ss, 0, sizeof(ss));
            if (!L.estimate(*G, &ss, cfg.cg_fixed_iterations, false)) break;
            std::vector<double> vsh; L.voxel_sh(*G, vsh, nullptr);
            OptConfig o = cfg; o.thres_shell = thres; o.grid_level = gl; o.rgbd_level = pl;
            optimize(*G, *F, cam, o, vsh, nullptr);
            orc_recompute_colors(G, F, cam.intr, cam.dist, cam.poses.data(), cfg.occlusion_distance, cfg.num_observations);
            ++done;
        }
        if (gl > 0) { Grid<VoxelSBR>* up = upsample(*G); delete G; G = up; }
    }
    *grid_io = G;
    for (int i = 0; i < 4; ++i) intr[i] = cam.intr[i]; for (int i = 0; i < 5; ++i) dist[i] = cam.dist[i];
    for (int i = 0; i < 6 * F->K; ++i) poses[i] = cam.poses[i];
    if (levels_done) *levels_done = done;
    return 0;
}

double orc_shading_row(int32_t vx, int32_t vy, int32_t vz, const double* sh9, double pyr_scale, double voxel_size,
                       int32_t w, int32_t h, const float* lum, const double* prm, double* J29) {
    ShadingRowConst k; k.vx = vx; k.vy = vy; k.vz = vz; for (int i = 0; i < 9; ++i) k.sh[i] = sh9[i];
    k.pyr_scale = pyr_scale; k.voxel_size = voxel_size; k.w = w; k.h = h; k.lum = lum;
    if (!J29) return shading_residual<double>(k, prm);
    Jet<P_TOTAL> p[P_TOTAL]; for (int i = 0; i < P_TOTAL; ++i) p[i] = Jet<P_TOTAL>::var(prm[i], i);
    const Jet<P_TOTAL> r = shading_residual<Jet<P_TOTAL>>(k, p);
    for (int i = 0; i < P_TOTAL; ++i) J29[i] = r.v[i];
    return r.a;
}
void orc_bicubic(const float* img, int32_t w, int32_t h, double r, double c, double* f, double* dfdr, double* dfdc) { bicubic(img, w, h, r, c, f, dfdr, dfdc); }
void orc_pose_to_mat(const double* pose6, float* R9, float* t3) {
    double R[9], t[3]; pose_aa_to_mat(pose6, R, t); for (int i = 0; i < 9; ++i) R9[i] = (float)R[i]; for (int i = 0; i < 3; ++i) t3[i] = (float)t[i];
}
uint64_t orc_hash(int32_t x, int32_t y, int32_t z) { return (uint64_t)V3iHash()({x, y, z}); }
int32_t orc_round_trunc(float v) { return round_trunc(v); }

// ---- primitive probes ------------------------------------------------------------------------------------------------
double orc_sdf_to_weight(double sdf, double truncation) { return sdf_to_weight(sdf, truncation); }
double orc_varying_lambda(int32_t it, int32_t n, double l0, double l1) { return varying_lambda(it, n, l0, l1); }
int32_t orc_project_f(const float* k4, const float* dist5, int32_t w, int32_t h, const float* p3, float* p2f, int32_t* p2i) {
    CameraF cam; cam.fx = k4[0]; cam.fy = k4[1]; cam.cx = k4[2]; cam.cy = k4[3]; cam.w = w; cam.h = h; for (int i = 0; i < 5; ++i) cam.k[i] = dist5[i];
    int pi[2]; const bool ok = cam.project(p3, p2f, pi); p2i[0] = pi[0]; p2i[1] = pi[1]; return ok ? 1 : 0;
}
static Image probe_image(int32_t w, int32_t h, const float* depth) { Image im; im.w = w; im.h = h; im.depth = depth; return im; }
int32_t orc_voxel_visible(float max_occlusion_distance, const float* pt3, int32_t w, int32_t h, const float* depth, int32_t x, int32_t y) {
    Colorizer col; col.max_occlusion_distance = max_occlusion_distance; return col.visible(pt3, probe_image(w, h, depth), x, y) ? 1 : 0;
}
float orc_observation_weight(int32_t w, int32_t h, const float* depth, const float* n3, int32_t x, int32_t y, const float* v3) {
    Colorizer col; return col.weight(probe_image(w, h, depth), n3, x, y, v3);
}
void orc_compute_color(int32_t n, const uint8_t* rgb, const float* weights, float* out3) {
    std::vector<Observation> obs((size_t)n);
    for (int i = 0; i < n; ++i) { for (int k = 0; k < 3; ++k) obs[i].color[k] = rgb[3 * i + k]; obs[i].weight = weights[i]; obs[i].frame = i; }
    Colorizer::mean_color(obs, out3);
}
void orc_filter(int32_t count, float* weights, int32_t keep, int32_t* order) {
    std::vector<Observation> obs((size_t)count);
    for (int i = 0; i < count; ++i) { obs[i].weight = weights[i]; obs[i].frame = i; }
    Colorizer::filter(obs, (size_t)keep);
    for (int i = 0; i < count; ++i) { weights[i] = obs[i].weight; order[i] = obs[i].frame; }
}
double orc_chroma_weight(const uint8_t* c3, const uint8_t* cn3) { return chroma_weight(c3, cn3); }
double orc_reg_row(int32_t type, const double* x, double sdf0, double* J) {
    Row r; r.type = type; r.v = 0; r.f = -1; r.dir = 0; r.weight = 1.0; r.sdf0 = sdf0; r.ncols = type == 1 ? 7 : (type == 2 ? 1 : 2);
    std::vector<double> xg(x, x + r.ncols); for (int i = 0; i < r.ncols; ++i) r.cols[i] = i;
    return eval_row(r, xg, J);
}
double orc_sh_data_row(double luminance, const float* n3, double albedo, const double* sh9, double* J9) {
    double b[9]; sh_basis((double)n3[0], (double)n3[1], (double)n3[2], b);
    double ab[9]; for (int j = 0; j < 9; ++j) { ab[j] = albedo * b[j]; if (J9) J9[j] = ab[j]; }
    return sh_data_raw(albedo, ab, sh9, luminance);
}
void orc_world_to_voxel(float voxel_size, const float* p3, int32_t* out3) {
    Grid<VoxelSBR> g(voxel_size); const V3i v = g.worldToVoxel({p3[0], p3[1], p3[2]}); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
void* orc_mc_extract(void* g, int32_t use_refined) { auto* M = new MeshOut(); marching_cubes(*(Grid<VoxelSBR>*)g, use_refined != 0, *M); return M; }
void orc_mesh_counts(void* mesh, int64_t* nv, int64_t* nf) { auto* M = (MeshOut*)mesh; *nv = (int64_t)(M->vertices.size() / 3); *nf = (int64_t)(M->faces.size() / 3); }
void orc_mesh_get(void* mesh, float* verts, uint8_t* colors, int32_t* faces) {
    auto* M = (MeshOut*)mesh;
    if (verts) std::memcpy(verts, M->vertices.data(), M->vertices.size() * sizeof(float));
    if (colors) std::memcpy(colors, M->colors.data(), M->colors.size());
    if (faces) std::memcpy(faces, M->faces.data(), M->faces.size() * sizeof(int32_t));
}
void orc_mesh_free(void* mesh) { delete (MeshOut*)mesh; }

static void dense_to_crs(int m, int n, const double* A, CRS& J) {
    J.rows = m; J.cols = n; J.ptr.assign(m + 1, 0); J.col.clear(); J.val.clear();
    for (int r = 0; r < m; ++r) { for (int c = 0; c < n; ++c) { J.col.push_back(c); J.val.push_back(A[(size_t)r * n + c]); } J.ptr[r + 1] = (int)J.col.size(); }
}
int32_t orc_test_lm_dense(int32_t m, int32_t n, int32_t nblocks, const int32_t* bs, const double* A, const double* b, double* x_io,
                          int32_t max_it, int32_t stop_first, int32_t cg_fixed, int32_t* cg_iters, double* costs2) {
    CRS J; dense_to_crs(m, n, A, J);
    const std::vector<double> Aval = J.val;
    std::vector<int> bstart, bsize; int o = 0; for (int i = 0; i < nblocks; ++i) { bstart.push_back(o); bsize.push_back(bs[i]); o += bs[i]; }
    EvalFn eval = [&](const double* x, double* cost, std::vector<double>* res, CRS* Jout) -> bool {
        res->resize(m); double cs = 0.0;
        for (int r = 0; r < m; ++r) { double s = -b[r]; for (int c = 0; c < n; ++c) s += A[(size_t)r * n + c] * x[c]; (*res)[r] = s; cs += 0.5 * s * s; }
        if (Jout) Jout->val = Aval;
        *cost = cs; return true; };
    LMOptions lo; lo.max_num_iterations = max_it; lo.stop_after_first_successful_step = stop_first != 0; lo.cg_fixed_iterations = cg_fixed;
    std::vector<double> x(x_io, x_io + n);
    LMSummary s = lm_minimize(eval, J, bstart, bsize, x, lo);
    for (int i = 0; i < n; ++i) x_io[i] = x[i];
    if (cg_iters) for (size_t i = 0; i < s.cg_iterations.size() && i < 50; ++i) cg_iters[i] = s.cg_iterations[i];
    if (costs2) { costs2[0] = s.initial_cost; costs2[1] = s.final_cost; }
    return s.iterations;
}
int32_t orc_test_cgnr(int32_t m, int32_t n, int32_t nblocks, const int32_t* bs, const double* A, const double* b, const double* D,
                      int32_t cg_fixed, double* x_out) {
    CRS J; dense_to_crs(m, n, A, J);
    BlockJacobi M; int o = 0, off = 0;
    std::vector<int> col_block(n);
    for (int i = 0; i < nblocks; ++i) { M.start.push_back(o); M.size.push_back(bs[i]); M.off.push_back(off); for (int k = 0; k < bs[i]; ++k) col_block[o + k] = i; o += bs[i]; off += bs[i] * bs[i]; }
    M.inv.assign(off, 0.0); M.update(J, col_block, D);
    LMOptions lo; lo.cg_fixed_iterations = cg_fixed;
    return cgnr_solve(J, b, D, M, lo, x_out);
}

// ---- TSDF fusion (app_fusion.cpp:107-200)
void* orc_fusion_create(float voxel_size, float depth_min, float depth_max, const float* clip6) {
    auto* f = new Fusion(voxel_size, depth_min, depth_max);
    if (clip6) for (int i = 0; i < 6; ++i) f->clip[i] = clip6[i];
    return f;
}
void orc_fusion_integrate(void* fp, int32_t dw, int32_t dh, const float* dc, int32_t cw, int32_t ch, const float* cc, const float* depth, const uint8_t* bgr,
                          const float* pose16, int32_t erode_window) {
    auto* f = (Fusion*)fp;
    const PinCam dcam{dc[0], dc[1], dc[2], dc[3], dw, dh}, ccam{cc[0], cc[1], cc[2], cc[3], cw, ch};
    std::vector<float> er((size_t)dw * dh), nrm((size_t)dw * dh * 3);
    erode_discontinuities(dw, dh, depth, erode_window, 0.5f, er.data());           // processing.h:57 default max_depth_diff
    compute_normals(dcam, er.data(), 0.3f, nrm.data());                             // processing.h:53 default depth_threshold
    f->integrate(dcam, ccam, er.data(), bgr, nrm.data(), pose16);
}
void orc_fusion_finish(void* fp, int32_t iters) { auto* f = (Fusion*)fp; f->correct_sdf((unsigned)iters); f->clear_invalid(); }
int64_t orc_fusion_size(void* fp) { return (int64_t)((Fusion*)fp)->grid.size(); }
void orc_fusion_export(void* fp, int32_t* keys, float* sdf, float* weight, uint8_t* color) {
    auto& g = ((Fusion*)fp)->grid; size_t i = 0;
    for (auto it = g.data.begin(); it != g.data.end(); ++it, ++i) {
        keys[3 * i] = it->first.x; keys[3 * i + 1] = it->first.y; keys[3 * i + 2] = it->first.z;
        sdf[i] = it->second.sdf; weight[i] = it->second.weight;
        for (int c = 0; c < 3; ++c) color[3 * i + c] = it->second.color[c];
    }
}
void orc_fusion_free(void* fp) { delete (Fusion*)fp; }
void orc_erode_discontinuities(int32_t w, int32_t h, const float* in, int32_t window, float max_diff, float* out) { erode_discontinuities(w, h, in, window, max_diff, out); }
void orc_compute_normals(int32_t w, int32_t h, const float* c, const float* depth, float thr, float* normals) {
    compute_normals(PinCam{c[0], c[1], c[2], c[3], w, h}, depth, thr, normals);
}

}  // extern "C"
