// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// CPU restatement of the reference's TSDF fusion, the stage in front of the path.  PARITY UNPINNED like the
// rest of the oracle: the reference ships no golden volumes and cannot be built here.  Float expressions are evaluated left to right
// without FMA contraction (oracle/Makefile), 3-vector dot products as ((a0*b0 + a1*b1) + a2*b2) like the observation pass.
//   SparseVoxelGrid<Voxel>::integrate / alloc / computeFrustumBounds      sparse_voxel_grid.cpp:301-467,573-606
//   math::computeFrustumPoints, robustKernel, withinBounds                math.cpp:43-71,131-148
//   Camera::project2 / unproject2                                         camera.cpp:157-199
//   erodeDiscontinuities, computeVertexMap, computeNormals                rgbd/processing.cpp:49-127,184-232
//   SDFAlgorithms::correctSDF / clearInvalidVoxels                        sdf/algorithms.cpp:260-366
//   AppFusion::fuseSDF                                                    apps/src/app_fusion.cpp:107-200
#pragma once
#include "grid.hpp"
#include <cstring>
#include <limits>

namespace orc {

struct PinCam { float fx, fy, cx, cy; int w, h; };

inline float robust_kernel(float val, float thres = 2.0f) { const float div = 1.0f + thres * val; return 1.0f / (div * div * div); }

// Matrix4f::inverse(): adjugate over determinant from the 2x2 minors of the row pairs (01) and (23), in float
inline void inverse4f(const float* m, float* inv) {
    const float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const float id = 1.0f / (((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0);
    inv[0] = ((m[5] * c5 - m[6] * c4) + m[7] * c3) * id;      inv[1] = ((-m[1] * c5 + m[2] * c4) - m[3] * c3) * id;
    inv[2] = ((m[13] * s5 - m[14] * s4) + m[15] * s3) * id;   inv[3] = ((-m[9] * s5 + m[10] * s4) - m[11] * s3) * id;
    inv[4] = ((-m[4] * c5 + m[6] * c2) - m[7] * c1) * id;     inv[5] = ((m[0] * c5 - m[2] * c2) + m[3] * c1) * id;
    inv[6] = ((-m[12] * s5 + m[14] * s2) - m[15] * s1) * id;  inv[7] = ((m[8] * s5 - m[10] * s2) + m[11] * s1) * id;
    inv[8] = ((m[4] * c4 - m[5] * c2) + m[7] * c0) * id;      inv[9] = ((-m[0] * c4 + m[1] * c2) - m[3] * c0) * id;
    inv[10] = ((m[12] * s4 - m[13] * s2) + m[15] * s0) * id;  inv[11] = ((-m[8] * s4 + m[9] * s2) - m[11] * s0) * id;
    inv[12] = ((-m[4] * c3 + m[5] * c1) - m[6] * c0) * id;    inv[13] = ((m[0] * c3 - m[1] * c1) + m[2] * c0) * id;
    inv[14] = ((-m[12] * s3 + m[13] * s1) - m[14] * s0) * id; inv[15] = ((m[8] * s3 - m[9] * s1) + m[10] * s0) * id;
}
// pose.topLeftCorner<3,3>() * p + pose.topRightCorner<3,1>(): a FIXED-SIZE product, every coefficient a halving reduction a0 + (a1 + a2)
inline void xform(const float* T /*4x4 row-major*/, const float p[3], float q[3]) {
    for (int i = 0; i < 3; ++i) q[i] = (T[4 * i] * p[0] + (T[4 * i + 1] * p[1] + T[4 * i + 2] * p[2])) + T[4 * i + 3];
}

// rgbd/processing.cpp:184-232
inline void erode_discontinuities(int w, int h, const float* in, int window, float max_diff, float* out) {
    std::memcpy(out, in, sizeof(float) * (size_t)w * h);
    if (window <= 0) return;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const size_t idx = (size_t)y * w + x; const float d_ref = in[idx];
        if (d_ref == 0.0f) { out[idx] = 0.0f; continue; }
        bool valid = true;
        for (int v = std::max(0, y - window); v <= std::min(y + window, h - 1) && valid; ++v)
            for (int u = std::max(0, x - window); u <= std::min(x + window, w - 1); ++u) {
                const float d = in[(size_t)v * w + u];
                if (d == 0.0f || std::abs(d - d_ref) > max_diff) { valid = false; break; }
            }
        if (!valid) out[idx] = 0.0f;
    }
}
// rgbd/processing.cpp:49-127: vertex map (x0*d, y0*d, d) then cross(tangent_y, tangent_x).normalized(); border and invalid pixels stay 0
inline void compute_normals(const PinCam& cam, const float* depth, float thr, float* normals /*[h][w][3]*/) {
    const int w = cam.w, h = cam.h;
    const float fx_inv = 1.0f / cam.fx, fy_inv = 1.0f / cam.fy;
    std::vector<float> vm((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const float d = depth[(size_t)y * w + x];
        const float x0 = ((float)x - cam.cx) * fx_inv, y0 = ((float)y - cam.cy) * fy_inv;
        float* v = &vm[((size_t)y * w + x) * 3]; v[0] = x0 * d; v[1] = y0 * d; v[2] = d;
    }
    std::memset(normals, 0, sizeof(float) * (size_t)w * h * 3);
    for (int y = 1; y < h - 1; ++y) for (int x = 1; x < w - 1; ++x) {
        const float* v = &vm[((size_t)y * w + x) * 3];
        if (v[2] == 0.0f) continue;
        const float* x0 = &vm[((size_t)y * w + x - 1) * 3]; const float* x1 = &vm[((size_t)y * w + x + 1) * 3];
        const float* y0 = &vm[((size_t)(y - 1) * w + x) * 3]; const float* y1 = &vm[((size_t)(y + 1) * w + x) * 3];
        if (x0[2] == 0.0f || x1[2] == 0.0f || y0[2] == 0.0f || y1[2] == 0.0f) continue;
        const float tx[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]}, ty[3] = {y1[0] - y0[0], y1[1] - y0[1], y1[2] - y0[2]};
        const float ntx = std::sqrt(tx[0] * tx[0] + (tx[1] * tx[1] + tx[2] * tx[2])), nty = std::sqrt(ty[0] * ty[0] + (ty[1] * ty[1] + ty[2] * ty[2]));
        if (ntx < thr && nty < thr) {
            float n[3] = {ty[1] * tx[2] - ty[2] * tx[1], ty[2] * tx[0] - ty[0] * tx[2], ty[0] * tx[1] - ty[1] * tx[0]};
            const float sq = n[0] * n[0] + (n[1] * n[1] + n[2] * n[2]);
            if (sq > 0.0f) { const float l = std::sqrt(sq); n[0] /= l; n[1] /= l; n[2] /= l; }
            float* o = &normals[((size_t)y * w + x) * 3]; o[0] = n[0]; o[1] = n[1]; o[2] = n[2];
        }
    }
}

struct Fusion {
    Grid<Voxel> grid;
    float depth_min, depth_max, integration_weight_sample = 10.0f;
    float clip[6] = {0, 0, 0, 0, 0, 0};
    Fusion(float voxel_size, float dmin, float dmax) : grid(voxel_size), depth_min(dmin), depth_max(dmax) {}

    static bool within(const int b[6], const V3i& p) { return !(p.x < b[0] || p.x > b[1] || p.y < b[2] || p.y > b[3] || p.z < b[4] || p.z > b[5]); }
    bool within_clip(const V3f& p) const { return !(p.x < clip[0] || p.x > clip[1] || p.y < clip[2] || p.y > clip[3] || p.z < clip[4] || p.z > clip[5]); }
    static void unproject2(const PinCam& c, int ux, int uy, float depth, float out[3]) {
        if (depth == 0.0f) { out[0] = out[1] = out[2] = 0.0f; return; }
        const float x = ((float)ux - c.cx) / c.fx, y = ((float)uy - c.cy) / c.fy;
        out[0] = depth * x; out[1] = depth * y; out[2] = depth;
    }

    // sparse_voxel_grid.cpp:573-606 — note floor()/ceil() act on METRES before the voxel conversion
    void frustum_bounds(const PinCam& cam, const float* pose, int b[6]) const {
        const int lo = std::numeric_limits<int>::min(), hi = std::numeric_limits<int>::max();
        b[0] = hi; b[1] = lo; b[2] = hi; b[3] = lo; b[4] = hi; b[5] = lo;
        const int px[4] = {0, cam.w - 1, cam.w - 1, 0}, py[4] = {0, 0, cam.h - 1, cam.h - 1};
        for (int i = 0; i < 8; ++i) {
            float c[3], pt[3]; unproject2(cam, px[i & 3], py[i & 3], i < 4 ? depth_min : depth_max, c);
            xform(pose, c, pt);
            const V3i pl = grid.worldToVoxel({(float)(int)std::floor(pt[0]), (float)(int)std::floor(pt[1]), (float)(int)std::floor(pt[2])});
            const V3i pu = grid.worldToVoxel({(float)(int)std::ceil(pt[0]), (float)(int)std::ceil(pt[1]), (float)(int)std::ceil(pt[2])});
            b[0] = std::min(b[0], std::min(pl.x, pu.x)); b[1] = std::max(b[1], std::max(pl.x, pu.x));
            b[2] = std::min(b[2], std::min(pl.y, pu.y)); b[3] = std::max(b[3], std::max(pl.y, pu.y));
            b[4] = std::min(b[4], std::min(pl.z, pu.z)); b[5] = std::max(b[5], std::max(pl.z, pu.z));
        }
    }

    // sparse_voxel_grid.cpp:401-467
    void alloc(const PinCam& cam, const float* depth, const float* pose, const int bounds[6]) {
        const float ray_step = grid.voxel_size * 0.25f, trunc = grid.truncation;
        float cn = 0.0f; for (int i = 0; i < 6; ++i) cn += clip[i] * clip[i];
        const bool use_clip = std::sqrt(cn) > 0.0f;
        const Voxel v_init;
        for (int y = 0; y < cam.h; ++y) for (int x = 0; x < cam.w; ++x) {
            const float d = depth[(size_t)y * cam.w + x];
            if (d == 0.0f) continue;
            float pc[3]; unproject2(cam, x, y, 1.0f, pc);
            V3i last{0, 0, 0};
            for (float d_off = -trunc; d_off <= trunc; d_off += ray_step) {
                const float s = d + d_off; const float pr[3] = {pc[0] * s, pc[1] * s, pc[2] * s};
                float pw[3]; xform(pose, pr, pw);
                const V3i pg = grid.worldToVoxel({pw[0], pw[1], pw[2]});
                if (pg == last) continue;
                last = pg;
                if (!within(bounds, pg)) continue;
                if (use_clip && !within_clip(grid.voxelToWorld(pg))) continue;
                for (int bz = -1; bz <= 1; ++bz) for (int by = -1; by <= 1; ++by) for (int bx = -1; bx <= 1; ++bx) {
                    const V3i pb{pg.x + bx, pg.y + by, pg.z + bz};
                    if (!grid.exists(pb)) grid.setVoxel(pb, v_init);
                }
            }
        }
    }

    // sparse_voxel_grid.cpp:301-398
    void integrate(const PinCam& dcam, const PinCam& ccam, const float* depth, const uint8_t* bgr, const float* normals /* may be null */, const float* pose_c2w) {
        float w2c[16]; inverse4f(pose_c2w, w2c);
        int bounds[6]; frustum_bounds(dcam, pose_c2w, bounds);
        alloc(dcam, depth, pose_c2w, bounds);
        const float trunc = grid.truncation, iws = integration_weight_sample;
        for (auto it = grid.data.begin(); it != grid.data.end(); ++it) {
            const V3i& pg = it->first;
            if (!within(bounds, pg)) continue;
            const V3f pw = grid.voxelToWorld(pg); Voxel& v = it->second;
            const float pwv[3] = {pw.x, pw.y, pw.z}; float p[3]; xform(w2c, pwv, p);
            if (p[2] < 0.0f) continue;
            int px = round_trunc((p[0] * dcam.fx) / p[2] + dcam.cx), py = round_trunc((p[1] * dcam.fy) / p[2] + dcam.cy);
            if (px < 0 || py < 0 || px >= dcam.w || py >= dcam.h) continue;
            const float d = depth[(size_t)py * dcam.w + px];
            if (d <= 0.0f) continue;
            const float sdf = d - p[2];
            if (sdf <= -trunc) continue;
            const float tsdf = sdf >= 0.0f ? std::min(trunc, sdf) : std::max(-trunc, sdf);
            float wu = 1.0f;
            if (iws > 0) {
                float wn = 1.0f;
                if (normals) {
                    const float* n = &normals[((size_t)py * dcam.w + px) * 3];
                    const float sq = p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]);
                    float pn[3] = {p[0], p[1], p[2]};
                    if (sq > 0.0f) { const float l = std::sqrt(sq); pn[0] /= l; pn[1] /= l; pn[2] /= l; }
                    wn = 1.0f - std::abs(pn[0] * n[0] + (pn[1] * n[1] + pn[2] * n[2]));
                    wn = std::max(std::min(wn, 1.0f), 0.0f);
                    wn = std::max(iws * robust_kernel(wn), 1.0f);
                }
                const float wd = std::max(iws * robust_kernel(2.0f * std::abs(tsdf) / trunc), 1.0f);
                const float dn = (d - depth_min) / (depth_max - depth_min);
                const float wz = std::max(iws * (1.0f - dn), 1.0f);
                wu = std::max(((wn + wd) + wz) / 3.0f, 3.0f);
            }
            const float w_old = v.weight, w_new = w_old + wu;
            v.sdf = (v.sdf * w_old + sdf * wu) / w_new;
            px = round_trunc((p[0] * ccam.fx) / p[2] + ccam.cx); py = round_trunc((p[1] * ccam.fy) / p[2] + ccam.cy);
            if (px >= 0 && py >= 0 && px < ccam.w && py < ccam.h) {
                const uint8_t* c = &bgr[((size_t)py * ccam.w + px) * 3];
                for (int k = 0; k < 3; ++k) {                     // voxel colour is R,G,B; the image is B,G,R
                    const float c_old = (float)v.color[k], c_new = (float)c[2 - k];
                    v.color[k] = (uint8_t)((c_old * v.weight + c_new * wu) / w_new);
                }
            }
            v.weight = w_new;
        }
    }

    // sdf/algorithms.cpp:260-331: in-place sweeps in iteration order; a voxel compares against its value at the START of its visit and
    // keeps the LAST qualifying neighbour of the (k, j, i) loop
    void correct_sdf(unsigned num_iter = 10) {
        for (unsigned iter = 0; iter < num_iter; ++iter) {
            bool has_update = false;
            for (auto it = grid.data.begin(); it != grid.data.end(); ++it) {
                const V3i& vp = it->first; Voxel& v = it->second;
                if (!grid.valid(vp)) continue;
                const V3f vc = grid.voxelToWorld(vp);
                const double sdf = v.sdf; const double sgn = sdf >= 0.0 ? 1.0 : -1.0;       // weight > 0 here: the "unseen voxel" branch is dead
                for (int k = -1; k <= 1; ++k) for (int j = -1; j <= 1; ++j) for (int i = -1; i <= 1; ++i) {
                    if (k == 0 && j == 0 && i == 0) continue;
                    const V3i nb{vp.x + i, vp.y + j, vp.z + k};
                    if (!grid.valid(nb)) continue;
                    const Voxel& vn = grid.voxel(nb); const V3f nc = grid.voxelToWorld(nb);
                    const double sdf_nb = vn.sdf, sgn_nb = sdf_nb >= 0.0 ? 1.0 : -1.0;
                    const float dx = vc.x - nc.x, dy = vc.y - nc.y, dz = vc.z - nc.z;
                    const double dist_nb = sdf_nb + sgn_nb * (double)std::sqrt(dx * dx + (dy * dy + dz * dz));     // Vec3f::norm(): halving reduction
                    if (std::abs(dist_nb) < std::abs(sdf) && sgn == sgn_nb) { v.sdf = (float)dist_nb; v.weight = 1.0f; has_update = true; }
                }
            }
            if (!has_update) break;
        }
    }
    // sdf/algorithms.cpp:341-362; erase keeps the order of the survivors
    void clear_invalid() {
        std::vector<V3i> bad;
        for (auto it = grid.data.begin(); it != grid.data.end(); ++it) if (!(it->second.weight > 0.0f)) bad.push_back(it->first);
        for (const V3i& p : bad) grid.data.erase(p);
    }
};

}  // namespace orc
