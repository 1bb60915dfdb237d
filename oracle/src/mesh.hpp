// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Restatement of the iso-surface extraction that writes mesh_g*_p*.ply:
//   mesh/marching_cubes.cpp:64-95    extractMesh  (cells in grid iteration order, merge, removeDegenerateFaces)
//   mesh/marching_cubes.cpp:97-140   merge        (std::map<tuple<float,float,float>,int>: vertex index = order of first appearance,
//                                                  colour of the first appearance, (c * 255).cast<uchar>())
//   mesh/marching_cubes.cpp:143-243  extractSurfaceAt (three forward neighbours must exist; edge e runs corner EA[e] -> EB[e])
//   mesh/marching_cubes.cpp:250-275  computeLutIndex  (all 8 corners stored with weight != 0, else 0; bit i = sdf < 0)
//   mesh/marching_cubes.cpp:278-317  interpolate / getVertex
//   mesh/util.cpp:174-200            removeDegenerateFaces
// The triangulation table is Bourke's (the product's packed copy, itself checked against the reference's table by
// tests/test_io_cpu.py where /root/reference exists).
#pragma once
#include <cmath>
#include <map>
#include <tuple>
#include <vector>
#include "grid.hpp"
#include "../../intrinsic3d_amd/csrc/host/mc_table.hpp"

namespace orc {

struct MeshOut { std::vector<float> vertices; std::vector<uint8_t> colors; std::vector<int32_t> faces; };

inline void mc_interp3(float t0, float t1, const float v0[3], const float v1[3], float out[3]) {
    const float iso = 0.0f;
    if (std::fabs(iso - t0) < 0.00001f) { for (int i = 0; i < 3; ++i) out[i] = v0[i]; return; }
    if (std::fabs(iso - t1) < 0.00001f) { for (int i = 0; i < 3; ++i) out[i] = v1[i]; return; }
    if (std::fabs(t0 - t1) < 0.00001f) { for (int i = 0; i < 3; ++i) out[i] = v0[i]; return; }
    float mu = (iso - t0) / (t1 - t0);
    mu = std::max(std::min(mu, 1.0f), 0.0f);
    for (int i = 0; i < 3; ++i) out[i] = v0[i] + mu * (v1[i] - v0[i]);
}

// use_refined: the application meshes a clone whose sdf has been overwritten by sdf_refined (applyRefinedSdf, app_intrinsic3d.cpp:170)
inline void marching_cubes(const Grid<VoxelSBR>& g, bool use_refined, MeshOut& M) {
    static const int CO[8][3] = {{1, 1, 0}, {1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 1}, {1, 0, 1}, {0, 0, 1}, {0, 1, 1}};
    static const int EA[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, EB[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    struct Vtx { float p[3], c[3]; };
    std::vector<Vtx> tri;                                     // 3 per triangle, emission order
    for (auto it = g.data.begin(); it != g.data.end(); ++it) {
        const V3i v = it->first;
        if (!g.exists({v.x + 1, v.y, v.z}) || !g.exists({v.x, v.y + 1, v.z}) || !g.exists({v.x, v.y, v.z + 1})) continue;
        V3i c[8]; const VoxelSBR* vx[8]; bool ok = true;
        for (int i = 0; i < 8 && ok; ++i) { c[i] = {v.x + CO[i][0], v.y + CO[i][1], v.z + CO[i][2]}; if (!g.exists(c[i])) { ok = false; break; } vx[i] = &g.voxel(c[i]); if (vx[i]->weight == 0.0f) ok = false; }
        if (!ok) continue;
        int idx = 0;
        for (int i = 0; i < 8; ++i) if ((use_refined ? vx[i]->sdf_refined : vx[i]->sdf) < (double)0.0f) idx += 1 << i;
        if (idx == 0 || idx == 255) continue;
        Vtx ev[12]; const int mask = i3d::mc_edge_mask(idx);
        for (int e = 0; e < 12; ++e) if (mask & (1 << e)) {
            const int a = EA[e], b = EB[e];
            const float s1 = (float)(use_refined ? vx[a]->sdf_refined : vx[a]->sdf), s2 = (float)(use_refined ? vx[b]->sdf_refined : vx[b]->sdf);
            const V3f w1 = g.voxelToWorld(c[a]), w2 = g.voxelToWorld(c[b]);
            const float p1[3] = {w1.x, w1.y, w1.z}, p2[3] = {w2.x, w2.y, w2.z};
            mc_interp3(s1, s2, p1, p2, ev[e].p);
            const float sc = 1.0f / 255.0f;
            const float c1[3] = {(float)vx[a]->color[0] * sc, (float)vx[a]->color[1] * sc, (float)vx[a]->color[2] * sc};
            const float c2[3] = {(float)vx[b]->color[0] * sc, (float)vx[b]->color[1] * sc, (float)vx[b]->color[2] * sc};
            mc_interp3(s1, s2, c1, c2, ev[e].c);
        }
        for (int k = 0; k < 3 * i3d::mc_num_triangles(idx); ++k) tri.push_back(ev[i3d::mc_edge(idx, k)]);
    }
    M.vertices.clear(); M.colors.clear(); M.faces.clear();
    if (tri.empty()) return;
    std::map<std::tuple<float, float, float>, int> index;
    std::vector<int32_t> faces; faces.reserve(tri.size());
    for (const Vtx& t : tri) {
        const auto key = std::make_tuple(t.p[0], t.p[1], t.p[2]);
        auto f = index.find(key);
        if (f == index.end()) {
            const int id = (int)(M.vertices.size() / 3); index[key] = id;
            for (int k = 0; k < 3; ++k) { M.vertices.push_back(t.p[k]); M.colors.push_back((uint8_t)(t.c[k] * 255.0f)); }
            faces.push_back(id);
        } else faces.push_back(f->second);
    }
    for (size_t f = 0; f + 2 < faces.size(); f += 3) {        // removeDegenerateFaces
        const int v0 = faces[f], v1 = faces[f + 1], v2 = faces[f + 2];
        if (v0 == v1 || v0 == v2 || v1 == v2) continue;
        const float* a = &M.vertices[3 * (size_t)v0]; const float* b = &M.vertices[3 * (size_t)v1]; const float* c = &M.vertices[3 * (size_t)v2];
        const float e0[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, e1[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
        const float cr[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
        const double area = (double)std::sqrt(cr[0] * cr[0] + (cr[1] * cr[1] + cr[2] * cr[2]));      // Eigen norm(): halving reduction of 3 terms
        if (area == 0.0 || std::isnan(area) || std::isinf(area)) continue;
        M.faces.push_back(v0); M.faces.push_back(v1); M.faces.push_back(v2);
    }
}

}  // namespace orc
