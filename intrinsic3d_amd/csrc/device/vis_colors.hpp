// The debug colour modes of SDFVisualization (sdf/visualization.cpp:228-416 of the reference): what `colorize` paints on every voxel before the mesh of a mode is
// extracted.  ONE definition for both sides of the build: the device kernel (mesh_kernels.hip: k_vis_colors, neighbours through the grid's stencil table) and the
// host entry point on caller arrays (host/mesh.cpp: i3d_visualization_colors, neighbours through a map) instantiate the same function, so the CPU tests that
// hold the host instantiation to the reference's compiled SDFVisualization also pin the arithmetic the kernel runs.  Float arithmetic in the reference's
// operation order; both translation units are compiled without FMA contraction (8-bit truncations follow).
// The two subvolume modes ("subvol", "subvol_interp") paint Subvolumes::color(), which the reference draws from rand(): nothing to reproduce, not offered.
#pragma once
#include <cmath>
#include <cstdint>

#ifdef __HIPCC__
#define I3D_VIS_HD __host__ __device__
#else
#define I3D_VIS_HD
#endif

namespace i3d {

enum VisMode : int { VIS_VOXEL = 0, VIS_ALBEDO = 1, VIS_NORMALS = 2, VIS_LAPLACIAN = 3, VIS_INTENSITY = 4, VIS_INTENSITY_GRAD = 5, VIS_SHADING = 6,
                     VIS_SHADING_CONST_ALBEDO = 7, VIS_CHROMACITY = 8, VIS_NUM_MODES = 9 };
I3D_VIS_HD inline bool vis_mode_needs_sh(int mode) { return mode == VIS_SHADING || mode == VIS_SHADING_CONST_ALBEDO; }

// what a voxel's debug colour reads: itself and its 6-ring, in the order of SDFAlgorithms::collectRingNeighborhood (algorithms.cpp:75-92)
struct VisStencil {
    bool valid[7];               // [0] the voxel, [1..6] +x -x +y -y +z -z: stored AND weight > 0 (SparseVoxelGrid::valid, sparse_voxel_grid.cpp:253-259)
    float sdf[7];                // (float) sdf_refined; [0] always, the others where valid
    unsigned char color[3];      // the voxel's colour
    unsigned char color_px[3];   // "lum_grad" with a valid ring: the +x neighbour's colour AS THE REFERENCE'S LOOP SEES IT (vis_lum_grad_px below)
    double albedo;
    float sh[9];                 // shading modes: the subvolume coefficients interpolated at the voxel centre, cast to float
};

// std::min(std::max(v, lo), hi) with the standard's comparison order (color_util.cpp:70-90)
I3D_VIS_HD inline float vis_clampf(float v, float lo, float hi) { const float a = (v < lo) ? lo : v; return (hi < a) ? hi : a; }
// intensity(r, g, b) (color_util.cpp:41-46)
I3D_VIS_HD inline float vis_intensity(const unsigned char c[3]) { return 0.299f * (float)c[0] + 0.587f * (float)c[1] + 0.114f * (float)c[2]; }
// a fixed-size 3-vector's norm: Eigen sums a0 + (a1 + a2)
I3D_VIS_HD inline float vis_norm3(const float n[3]) { return sqrtf(n[0] * n[0] + (n[1] * n[1] + n[2] * n[2])); }

// SDFOperators::computeSurfaceNormal (operators.cpp:58-77): forward differences of the refined distance, zero unless the voxel and +x, +y, +z are valid
I3D_VIS_HD inline void vis_normal(const VisStencil& v, float n[3]) {
    n[0] = n[1] = n[2] = 0.0f;
    if (!(v.valid[0] && v.valid[1] && v.valid[3] && v.valid[5])) return;
    n[0] = v.sdf[1] - v.sdf[0]; n[1] = v.sdf[3] - v.sdf[0]; n[2] = v.sdf[5] - v.sdf[0];
    const float len = vis_norm3(n);
    if (len != 0.0f) { n[0] /= len; n[1] /= len; n[2] /= len; }
}

// applyColorIntensityGradient (:273-305) + SDFOperators::intensityGradient (operators.cpp:108-140): grey = x component of the forward intensity difference
I3D_VIS_HD inline unsigned char vis_lum_grad_byte(const unsigned char c0[3], const unsigned char cpx[3], bool ring) {
    float dx = 0.0f;
    if (ring) dx = vis_intensity(cpx) - vis_intensity(c0);
    return (unsigned char)vis_clampf(dx * 0.5f + 127.0f, 0.0f, 255.0f);
}
// The reference paints "lum_grad" IN PLACE while it walks the grid: the +x neighbour whose colour a voxel reads has already been repainted when the walk came
// past it earlier — and what it was repainted with depended on ITS +x neighbour in the same way.  So the colour a voxel sees there is defined along the chain of
// +x neighbours with falling visit rank (linear in its length: out along +x, back along -x).  G: px(i) / mx(i) -> index of the +x / -x neighbour, rank(i) -> position
// in the reference's walk (its unordered_map order),
// ring(i) -> all six neighbours valid, color(i, c[3]).  Call with ring(s) true; e = the colour of s's +x neighbour at the moment s is visited.
template <class G>
I3D_VIS_HD inline void vis_lum_grad_px(const G& g, long long s, unsigned char e[3]) {
    const long long t1 = g.px(s);
    g.color(t1, e);
    if (!(g.rank(t1) < g.rank(s))) return;                                    // not visited before s: still its own colour
    long long cur = t1;                                                       // the far end of the chain of +x neighbours visited ever earlier ...
    while (g.ring(cur)) { const long long nx = g.px(cur); if (!(g.rank(nx) < g.rank(cur))) break; cur = nx; }
    unsigned char carry = 0; bool have = false;
    for (;;) {                                                                // ... and back along -x to t1: what each one was repainted with, given what ITS +x neighbour showed then
        unsigned char c0[3], cx[3] = {carry, carry, carry};
        g.color(cur, c0);
        const bool ring = g.ring(cur);
        if (!have && ring) g.color(g.px(cur), cx);
        carry = vis_lum_grad_byte(c0, cx, ring); have = true;
        if (cur == t1) break;
        cur = g.mx(cur);
    }
    e[0] = e[1] = e[2] = carry;
}

I3D_VIS_HD inline void vis_color(int mode, const VisStencil& v, float truncation, unsigned char out[3]) {
    out[0] = v.color[0]; out[1] = v.color[1]; out[2] = v.color[2];
    bool ring = true;
    for (int i = 1; i < 7; ++i) ring = ring && v.valid[i];                        // SDFAlgorithms::checkVoxelsValid over the 1-ring (the voxel itself is not asked)
    switch (mode) {
    case VIS_ALBEDO: {                                                            // applyColorAlbedo (:308-315): scalarToColor<double>(albedo, 255)
        const double s = v.albedo * 255.0, a = (s < 0.0) ? 0.0 : s, b = (255.0 < a) ? 255.0 : a;
        out[0] = out[1] = out[2] = (unsigned char)b; break; }
    case VIS_NORMALS: {                                                           // applyColorNormals (:228-240)
        float n[3]; vis_normal(v, n);
        const float len = vis_norm3(n);
        float c[3] = {0.0f, 0.0f, 0.0f};
        if (len != 0.0f && !std::isnan(len)) for (int i = 0; i < 3; ++i) c[i] = 0.5f * n[i] + 0.5f;
        for (int i = 0; i < 3; ++i) out[i] = (unsigned char)(c[i] * 255.0f);
        break; }
    case VIS_LAPLACIAN: {                                                         // applyColorLaplacian (:243-259) + SDFOperators::laplacian (operators.cpp:80-105)
        float lap = 0.0f;
        if (ring) {
            const float sdf = v.sdf[0];
            const float dxx = v.sdf[1] + v.sdf[2] - 2.0f * sdf, dyy = v.sdf[3] + v.sdf[4] - 2.0f * sdf, dzz = v.sdf[5] + v.sdf[6] - 2.0f * sdf;
            lap = 0.5f * ((dxx + dyy + dzz) / truncation) + 0.5f;
        }
        out[0] = out[1] = out[2] = (unsigned char)vis_clampf(lap * 255.0f, 0.0f, 255.0f);
        break; }
    case VIS_INTENSITY:                                                           // applyColorIntensity (:262-270)
        out[0] = out[1] = out[2] = (unsigned char)vis_clampf(vis_intensity(v.color) * 1.0f, 0.0f, 255.0f);
        break;
    case VIS_INTENSITY_GRAD:                                                      // applyColorIntensityGradient (:273-305): the x component of the forward difference
        out[0] = out[1] = out[2] = vis_lum_grad_byte(v.color, v.color_px, ring);
        break;
    case VIS_SHADING: case VIS_SHADING_CONST_ALBEDO: {                            // applyColorShading (:318-359) + Shading::computeShading (shading.cpp:61-73)
        float n[3]; vis_normal(v, n);
        const float len = vis_norm3(n);
        if (len == 0.0f || std::isnan(len)) { out[0] = out[1] = out[2] = 0; break; }
        const float albedo = mode == VIS_SHADING_CONST_ALBEDO ? 0.7f : (float)v.albedo;
        float b[9] = {1.0f, n[1], n[2], n[0], n[0] * n[1], n[1] * n[2], (-n[0] * n[0]) - (n[1] * n[1]) + 2.0f * (n[2] * n[2]), n[0] * n[2], (n[0] * n[0]) - (n[1] * n[1])};
        bool ok = true;
        for (int i = 0; i < 9; ++i) if (std::isnan(b[i]) || std::isinf(b[i])) ok = false;
        if (!ok) for (int i = 0; i < 9; ++i) b[i] = 0.0f;
        float shad = 0.0f;
        if (ok && albedo != 0.0f && !std::isnan(albedo)) {                       // (the basis has a constant 1: its norm is not zero unless it was reset)
            float dot = 0.0f;
            for (int i = 0; i < 9; ++i) dot += v.sh[i] * b[i];                    // (Eigen's order for a run-time sized dot product is not pinned)
            shad = albedo * dot;
        }
        const double s = (double)shad * (double)255.0f;
        out[0] = out[1] = out[2] = (unsigned char)vis_clampf((float)s * 1.0f, 0.0f, 255.0f);
        break; }
    case VIS_CHROMACITY: {                                                        // applyColorChromacity (:362-373) + chromacity (color_util.cpp:61-67)
        const float lum = vis_intensity(v.color), f = 1.0f / ((lum < 0.001f) ? 0.001f : lum);
        for (int i = 0; i < 3; ++i) out[i] = (unsigned char)vis_clampf(((float)v.color[i] * f) * 255.0f * 0.5f, 0.0f, 255.0f);
        break; }
    default: break;                                                               // VIS_VOXEL: the voxel's own colour
    }
}

// Subvolumes::interpolate(values, point, linear = true) (subvolumes.cpp:164-205) over math::interpolationWeights / average (math.cpp:74-128), for the nine
// coefficients of the subvolumes whose packed indices are listed in `keys` (ascending).  applyColorShading takes the only subvolume's coefficients as they
// are when there is just one (:340-343).
I3D_VIS_HD inline unsigned long long vis_pack3(int x, int y, int z) {
    const long long B = 1ll << 20;
    return ((unsigned long long)(x + B) & 0x1fffffull) | (((unsigned long long)(y + B) & 0x1fffffull) << 21) | (((unsigned long long)(z + B) & 0x1fffffull) << 42);
}
I3D_VIS_HD inline int vis_find(const unsigned long long* keys, int S, unsigned long long k) {
    int lo = 0, hi = S;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
    return (lo < S && keys[lo] == k) ? lo : -1;
}
I3D_VIS_HD inline void vis_interpolate_sh(float wx_, float wy_, float wz_ /* voxelToWorld of the voxel */, float subvolume_size, const unsigned long long* keys, int S,
                                         const double* sh /* [S][9] */, float out[9]) {
    if (S == 1) { for (int j = 0; j < 9; ++j) out[j] = (float)sh[j]; return; }
    double o[9];
    for (int j = 0; j < 9; ++j) o[j] = 0.0;
    const float inv = 1.0f / subvolume_size;
    const float px = wx_ * inv - 0.5f, py = wy_ * inv - 0.5f, pz = wz_ * inv - 0.5f;
    const int x0 = (int)floorf(px), y0 = (int)floorf(py), z0 = (int)floorf(pz);
    const float wx = px - (float)x0, wy = py - (float)y0, wz = pz - (float)z0;
    const int dx[8] = {0, 1, 0, 0, 1, 0, 1, 1}, dy[8] = {0, 0, 1, 0, 1, 1, 0, 1}, dz[8] = {0, 0, 0, 1, 0, 1, 1, 1};
    const float w8[8] = {(1.0f - wx) * (1.0f - wy) * (1.0f - wz), wx * (1.0f - wy) * (1.0f - wz), (1.0f - wx) * wy * (1.0f - wz), (1.0f - wx) * (1.0f - wy) * wz,
                         wx * wy * (1.0f - wz), (1.0f - wx) * wy * wz, wx * (1.0f - wy) * wz, wx * wy * wz};
    float sum_w = 0.0f;
    for (int i = 0; i < 8; ++i) {
        const int id = vis_find(keys, S, vis_pack3(x0 + dx[i], y0 + dy[i], z0 + dz[i]));
        const float w = id >= 0 ? w8[i] : 0.0f;
        if (w == 0.0f) continue;
        const double* v = sh + (size_t)id * 9;
        if (sum_w == 0.0f) { for (int j = 0; j < 9; ++j) o[j] = (double)w * v[j]; }
        else { for (int j = 0; j < 9; ++j) o[j] += (double)w * v[j]; }
        sum_w += w;
    }
    if (sum_w != 0.0f) { const double sc = (double)(1.0f / sum_w); for (int j = 0; j < 9; ++j) o[j] = o[j] * sc; }
    for (int j = 0; j < 9; ++j) out[j] = (float)o[j];
}

}  // namespace i3d
