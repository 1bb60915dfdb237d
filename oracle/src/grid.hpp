// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// CPU restatement of the reference's hashed voxel grid and the level-transition
// algorithms.  It literally uses std::unordered_map with the reference hash,
// reserve(64) and max_load_factor(0.6) so that iteration ("visit") order — on
// which the albedo-regulariser edge set depends — is the one libstdc++ gives the
// reference (SURVEY.md hazard H1).
//   mat.h:88-93,112-125            round() / hash
//   sparse_voxel_grid.h:56-77      Voxel, VoxelSBR
//   sparse_voxel_grid.cpp:44-54    ctor (reserve / load factor / truncation = 5*voxel)
//   sparse_voxel_grid.cpp:211-259  worldToVoxel / voxelToWorld / exists / valid
//   sdf/algorithms.cpp:47-91,118-247,342-458   convert, ring, interpolate, upsample, clear*
#pragma once
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <algorithm>

namespace orc {

struct V3i {
    int x, y, z;
    bool operator==(const V3i& o) const { return x == o.x && y == o.y && z == o.z; }
};
inline V3i operator+(const V3i& a, const V3i& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

struct V3iHash {  // mat.h:117-124 — int -> size_t sign-extends before the multiply
    size_t operator()(const V3i& v) const {
        const size_t p0 = 73856093, p1 = 19349669, p2 = 83492791;
        return ((size_t)v.x * p0) ^ ((size_t)v.y * p1) ^ ((size_t)v.z * p2);
    }
};

struct V3f { float x, y, z; };

// mat.h:90 — "round" adds 0.5 and truncates toward zero
inline int round_trunc(float v) { return (int)(v + 0.5f); }

struct Voxel {      // sparse_voxel_grid.h:56-62
    float sdf = 0.0f; float weight = 0.0f; uint8_t color[3] = {0, 0, 0};
};
struct VoxelSBR {   // sparse_voxel_grid.h:69-77
    double sdf = 0.0; float weight = 0.0f; uint8_t color[3] = {0, 0, 0};
    double albedo = 0.6; double sdf_refined = 0.0;
};

template <class T>
struct Grid {
    std::unordered_map<V3i, T, V3iHash> data;
    float voxel_size, truncation;
    explicit Grid(float vs) : voxel_size(vs), truncation(vs * 5.0f) {
        data.reserve(64); data.max_load_factor(0.6f);   // sparse_voxel_grid.cpp:52-53
    }
    bool exists(const V3i& p) const { return data.find(p) != data.end(); }
    bool valid(const V3i& p) const {                    // sparse_voxel_grid.cpp:254-259
        auto it = data.find(p); return it != data.end() && it->second.weight > 0.0f; }
    T& voxel(const V3i& p) { return data.find(p)->second; }
    const T& voxel(const V3i& p) const { return data.find(p)->second; }
    V3f voxelToWorld(const V3i& v) const { return {(float)v.x * voxel_size, (float)v.y * voxel_size, (float)v.z * voxel_size}; }
    V3i worldToVoxel(const V3f& p) const {              // sparse_voxel_grid.cpp:211-221
        const float inv = 1.0f / voxel_size;
        return {round_trunc(p.x * inv), round_trunc(p.y * inv), round_trunc(p.z * inv)}; }
    void setVoxel(const V3i& p, const T& v) { data[p] = v; }
    size_t size() const { return data.size(); }
};

// algorithms.cpp:75-91 — order +x,-x,+y,-y,+z,-z
inline void ring6(const V3i& p, V3i out[6]) {
    out[0] = {p.x + 1, p.y, p.z}; out[1] = {p.x - 1, p.y, p.z};
    out[2] = {p.x, p.y + 1, p.z}; out[3] = {p.x, p.y - 1, p.z};
    out[4] = {p.x, p.y, p.z + 1}; out[5] = {p.x, p.y, p.z - 1};
}
template <class T> inline bool ring_valid(const Grid<T>& g, const V3i& p) {   // algorithms.cpp:240-247
    V3i nb[6]; ring6(p, nb); bool ok = true;
    for (int i = 0; i < 6; ++i) if (!g.valid(nb[i])) ok = false;
    return ok;
}

template <class T> inline void clear_invalid_voxels(Grid<T>& g) {            // algorithms.cpp:342-363
    std::vector<V3i> bad;
    for (auto it = g.data.begin(); it != g.data.end(); ++it) if (!g.valid(it->first)) bad.push_back(it->first);
    for (auto& p : bad) g.data.erase(p);
}

// algorithms.cpp:47-72 — Voxel -> VoxelSBR, re-inserted in the source map's iteration order
inline Grid<VoxelSBR>* convert(const Grid<Voxel>& g) {
    auto* out = new Grid<VoxelSBR>(g.voxel_size);
    for (auto it = g.data.begin(); it != g.data.end(); ++it) {
        VoxelSBR s; s.sdf = (double)it->second.sdf; s.weight = it->second.weight;
        for (int c = 0; c < 3; ++c) s.color[c] = it->second.color[c];
        s.sdf_refined = (double)it->second.sdf;
        out->setVoxel(it->first, s);
    }
    clear_invalid_voxels(*out);
    return out;
}

// math.cpp:103-128
inline void interpolation_weights(const float pos[3], V3i coords[8], float w[8]) {
    const int x0 = (int)std::floor(pos[0]), y0 = (int)std::floor(pos[1]), z0 = (int)std::floor(pos[2]);
    coords[0] = {x0, y0, z0};         coords[1] = {x0 + 1, y0, z0};
    coords[2] = {x0, y0 + 1, z0};     coords[3] = {x0, y0, z0 + 1};
    coords[4] = {x0 + 1, y0 + 1, z0}; coords[5] = {x0, y0 + 1, z0 + 1};
    coords[6] = {x0 + 1, y0, z0 + 1}; coords[7] = {x0 + 1, y0 + 1, z0 + 1};
    const float wx = pos[0] - (float)x0, wy = pos[1] - (float)y0, wz = pos[2] - (float)z0;
    w[0] = (1.0f - wx) * (1.0f - wy) * (1.0f - wz);
    w[1] = wx * (1.0f - wy) * (1.0f - wz);
    w[2] = (1.0f - wx) * wy * (1.0f - wz);
    w[3] = (1.0f - wx) * (1.0f - wy) * wz;
    w[4] = wx * wy * (1.0f - wz);
    w[5] = (1.0f - wx) * wy * wz;
    w[6] = wx * (1.0f - wy) * wz;
    w[7] = wx * wy * wz;
}

// algorithms.cpp:118-199 — note double fields pass through float accumulators (hazard 11)
inline VoxelSBR interpolate_voxel(const Grid<VoxelSBR>& g, const float pos[3]) {
    float avg_weight = 0.0f, avg_sdf = 0.0f, avg_albedo = 0.0f, avg_sdf_refined = 0.0f, avg_col[3] = {0, 0, 0};
    V3i coords[8]; float w8[8];
    interpolation_weights(pos, coords, w8);
    float sum_w = 0.0f; int cnt_valid = 0;
    for (int i = 0; i < 8; ++i) {
        if (!g.valid(coords[i])) continue;
        const VoxelSBR& v = g.voxel(coords[i]);
        const float w = w8[i];
        avg_sdf += w * (float)v.sdf;
        for (int c = 0; c < 3; ++c) avg_col[c] += w * (float)v.color[c];
        avg_weight += w * v.weight;
        avg_albedo += w * (float)v.albedo;
        avg_sdf_refined += w * (float)v.sdf_refined;
        sum_w += w; ++cnt_valid;
    }
    if (sum_w > 0.0f) {
        avg_sdf /= sum_w; for (int c = 0; c < 3; ++c) avg_col[c] /= sum_w;
        avg_weight /= sum_w; avg_albedo /= sum_w; avg_sdf_refined /= sum_w;
    }
    if (cnt_valid <= 4) avg_weight = 0.0f;
    VoxelSBR o;
    o.sdf = avg_sdf;
    for (int c = 0; c < 3; ++c) o.color[c] = (uint8_t)round_trunc(avg_col[c]);
    o.weight = std::max(avg_weight, 0.0f);
    o.albedo = avg_albedo; o.sdf_refined = avg_sdf_refined;
    return o;
}

// algorithms.cpp:202-235
inline Grid<VoxelSBR>* upsample(const Grid<VoxelSBR>& g) {
    auto* up = new Grid<VoxelSBR>(g.voxel_size * 0.5f);
    for (auto it = g.data.begin(); it != g.data.end(); ++it) {
        const V3i& p = it->first;
        for (int z = 0; z < 2; ++z) for (int y = 0; y < 2; ++y) for (int x = 0; x < 2; ++x) {
            const V3i pn = {2 * p.x + x, 2 * p.y + y, 2 * p.z + z};
            const float pf[3] = {(float)p.x + (float)x * 0.5f, (float)p.y + (float)y * 0.5f, (float)p.z + (float)z * 0.5f};
            up->setVoxel(pn, interpolate_voxel(g, pf));
        }
    }
    return up;
}

// algorithms.cpp:368-458
inline void clear_voxels_outside_thin_shell(Grid<VoxelSBR>& g, double thres_shell) {
    std::unordered_set<V3i, V3iHash> keep, drop;
    for (auto it = g.data.begin(); it != g.data.end(); ++it) {
        const V3i& p = it->first;
        if (!g.valid(p) || std::abs(it->second.sdf_refined) > thres_shell) continue;
        keep.insert(p);
        V3i nb[9]; ring6(p, nb);
        nb[6] = {p.x + 2, p.y, p.z}; nb[7] = {p.x, p.y + 2, p.z}; nb[8] = {p.x, p.y, p.z + 2};
        for (int i = 0; i < 9; ++i) if (g.exists(nb[i])) keep.insert(nb[i]);
    }
    for (auto it = g.data.begin(); it != g.data.end(); ++it) {
        const V3i& p = it->first;
        if (keep.find(p) != keep.end()) continue;
        const bool neg = it->second.sdf_refined < 0.0;
        bool crossing = false;
        for (int dz = -2; dz <= 2 && !crossing; ++dz) for (int dy = -2; dy <= 2 && !crossing; ++dy)
            for (int dx = -2; dx <= 2 && !crossing; ++dx) {
                if (!dx && !dy && !dz) continue;
                auto f = g.data.find({p.x + dx, p.y + dy, p.z + dz});
                if (f == g.data.end()) continue;
                if (neg ? (f->second.sdf_refined >= 0.0) : (f->second.sdf_refined < 0.0)) crossing = true;
            }
        if (!crossing) drop.insert(p);
    }
    for (auto& p : drop) g.data.erase(p);
}

}  // namespace orc
