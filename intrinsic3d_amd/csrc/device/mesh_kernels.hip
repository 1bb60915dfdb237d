// Iso-surface extraction on the resident grid: MarchingCubes<T>::extractSurfaceAt / computeLutIndex / getVertex / interpolate
// (mesh/marching_cubes.cpp:178-343), one lane per stored voxel in VISIT order, two passes (count, scan, emit).
// Cube corners and edges use the reference's numbering (corner 0 = (x+1,y+1,z), 1 = (x+1,y,z), 2 = (x,y,z), 3 = (x,y+1,z), 4..7 the same
// at z+1; edge e joins corners EA[e] -> EB[e] in that direction, which fixes the interpolation formula's operand order).
// The triangulation table is Bourke's (host/mc_table.hpp, unpacked by host/mesh.cpp): the triangle sequence of a cell equals the reference's.
#include "kernels.hpp"
#include "level_kernels.hpp"
#include "vis_colors.hpp"

namespace i3d {

static __device__ inline unsigned long long pack_key_m(int x, int y, int z) {
    const long long B = 1ll << 20;
    return ((unsigned long long)(x + B) & 0x1fffffull) | (((unsigned long long)(y + B) & 0x1fffffull) << 21) | (((unsigned long long)(z + B) & 0x1fffffull) << 42);
}
static __device__ inline unsigned int mix64_m(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned int)k;
}
static __device__ inline int hash_find_m(const HashTable& t, int x, int y, int z) {
    const unsigned long long key = pack_key_m(x, y, z);
    unsigned int h = mix64_m(key) & t.mask;
    for (;;) {
        const unsigned long long k = t.keys[h];
        if (k == key) return t.vals[h];
        if (k == ~0ull) return -1;
        h = (h + 1) & t.mask;
    }
}

// the 8 corners of the cell of voxel s (device indices, -1 = missing) and its configuration index; 0 when the cell produces nothing
static __device__ inline int cell_config(const GridView& g, const HashTable& t, int s, bool refined, int corner[8]) {
    const int N = g.N;
    const int px = g.nbr[(size_t)NB_PX * N + s], py = g.nbr[(size_t)NB_PY * N + s], pz = g.nbr[(size_t)NB_PZ * N + s];
    if (px < 0 || py < 0 || pz < 0) return 0;                                   // extractSurfaceAt: the three forward neighbours must exist
    corner[0] = g.nbr[(size_t)NB_PXY * N + s]; corner[1] = px; corner[2] = s; corner[3] = py;
    corner[5] = g.nbr[(size_t)NB_PXZ * N + s]; corner[6] = pz; corner[7] = g.nbr[(size_t)NB_PYZ * N + s];
    corner[4] = hash_find_m(t, g.cx[s] + 1, g.cy[s] + 1, g.cz[s] + 1);
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (corner[i] < 0 || g.weight[corner[i]] == 0.0f) return 0;             // computeLutIndex: every corner valid (weight != 0)
        const double v = refined ? g.x_sdf[corner[i]] : g.sdf0[corner[i]];
        if (v < 0.0) idx |= 1 << i;                                             // sdf < iso_value (0.0f; the comparison is done in double)
    }
    return idx;
}

__global__ void __launch_bounds__(256) k_mc_count(GridView g, HashTable t, const int* __restrict__ inv_rank, int refined, const unsigned char* __restrict__ ntri, int* __restrict__ counts) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.N) return;
    int corner[8];
    const int idx = cell_config(g, t, inv_rank[v], refined != 0, corner);
    counts[v] = (idx == 0 || idx == 255) ? 0 : (int)ntri[idx];
}

// MarchingCubes::interpolate (marching_cubes.cpp:297-309) on one component
static __device__ inline float mc_lerp(float t0, float t1, float v0, float v1) {
    if (fabsf(0.0f - t0) < 0.00001f) return v0;
    if (fabsf(0.0f - t1) < 0.00001f) return v1;
    if (fabsf(t0 - t1) < 0.00001f) return v0;
    float mu = (0.0f - t0) / (t1 - t0);
    mu = fmaxf(fminf(mu, 1.0f), 0.0f);
    return v0 + mu * (v1 - v0);
}

__global__ void __launch_bounds__(256) k_mc_emit(GridView g, HashTable t, const int* __restrict__ inv_rank, int refined, int color_mode,
                                                 const unsigned char* __restrict__ ntri, const signed char* __restrict__ tri, int tri_stride,
                                                 const int* __restrict__ offsets, float* __restrict__ pos, unsigned char* __restrict__ col,
                                                 const uchar4* __restrict__ mode_color /* colour modes >= 2: what k_vis_colors painted, by device index */) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.N) return;
    int corner[8];
    const int idx = cell_config(g, t, inv_rank[v], refined != 0, corner);
    if (idx == 0 || idx == 255) return;
    constexpr int EA[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, EB[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    const int nt = ntri[idx];
    size_t o = (size_t)offsets[v] * 3;
    for (int k = 0; k < 3 * nt; ++k, ++o) {
        const int e = tri[idx * tri_stride + k];
        const int a = corner[EA[e]], b = corner[EB[e]];
        const float s1 = (float)(refined ? g.x_sdf[a] : g.sdf0[a]), s2 = (float)(refined ? g.x_sdf[b] : g.sdf0[b]);
        // voxelToWorld = float(i) * voxel_size (sparse_voxel_grid.cpp:224-228)
        pos[3 * o + 0] = mc_lerp(s1, s2, (float)g.cx[a] * g.voxel_size, (float)g.cx[b] * g.voxel_size);
        pos[3 * o + 1] = mc_lerp(s1, s2, (float)g.cy[a] * g.voxel_size, (float)g.cy[b] * g.voxel_size);
        pos[3 * o + 2] = mc_lerp(s1, s2, (float)g.cz[a] * g.voxel_size, (float)g.cz[b] * g.voxel_size);
        uchar4 ca = mode_color ? mode_color[a] : g.color[a], cb = mode_color ? mode_color[b] : g.color[b];
        if (color_mode == 1) {                     // SDFVisualization::applyColorAlbedo: scalarToColor(albedo, 255) (visualization.cpp:308-315, color_util.cpp:70-78)
            const unsigned char ga = (unsigned char)fmin(fmax(g.x_alb[a] * 255.0, 0.0), 255.0), gb = (unsigned char)fmin(fmax(g.x_alb[b] * 255.0, 0.0), 255.0);
            ca = make_uchar4(ga, ga, ga, 0); cb = make_uchar4(gb, gb, gb, 0);
        }
        const float inv = 1.0f / 255.0f;
        // colours are interpolated as floats in [0,1] and converted with (c * 255).cast<uchar>() in merge() (marching_cubes.cpp:121)
        col[3 * o + 0] = (unsigned char)(mc_lerp(s1, s2, (float)ca.x * inv, (float)cb.x * inv) * 255.0f);
        col[3 * o + 1] = (unsigned char)(mc_lerp(s1, s2, (float)ca.y * inv, (float)cb.y * inv) * 255.0f);
        col[3 * o + 2] = (unsigned char)(mc_lerp(s1, s2, (float)ca.z * inv, (float)cb.z * inv) * 255.0f);
    }
}

void launch_mc_count(hipStream_t st, GridView g, HashTable t, const int* inv_rank, int refined, const unsigned char* ntri, int* counts) {
    if (g.N > 0) k_mc_count<<<(g.N + 255) / 256, 256, 0, st>>>(g, t, inv_rank, refined, ntri, counts);
}
void launch_mc_emit(hipStream_t st, GridView g, HashTable t, const int* inv_rank, int refined, int color_mode, const unsigned char* ntri, const signed char* tri,
                    int tri_stride, const int* offsets, float* pos, unsigned char* col, const uchar4* mode_color) {
    if (g.N > 0) k_mc_emit<<<(g.N + 255) / 256, 256, 0, st>>>(g, t, inv_rank, refined, mode_color ? 0 : color_mode, ntri, tri, tri_stride, offsets, pos, col, mode_color);
}

// SDFVisualization::applyColor* (sdf/visualization.cpp:228-416) for the debug colour modes: one lane per stored voxel, its 6-ring through the stencil table.
// The arithmetic is vis_color() of vis_colors.hpp — the function the host entry point i3d_visualization_colors instantiates for the CPU tests.
struct VisGridDev {                              // the accessors vis_lum_grad_px walks the resident grid with
    const GridView& g;
    __device__ long long px(long long i) const { return g.nbr[(size_t)NB_PX * g.N + i]; }
    __device__ long long mx(long long i) const { return g.nbr[(size_t)NB_MX * g.N + i]; }
    __device__ int rank(long long i) const { return g.rank[i]; }
    __device__ bool ring(long long i) const {
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 6; ++d) { const int nb = g.nbr[(size_t)d * g.N + i]; ok = ok && nb >= 0 && g.weight[nb] > 0.0f; }      // NB_PX .. NB_MZ are 0..5
        return ok;
    }
    __device__ void color(long long i, unsigned char c[3]) const { const uchar4 q = g.color[i]; c[0] = q.x; c[1] = q.y; c[2] = q.z; }
};
__global__ void __launch_bounds__(256) k_vis_colors(GridView g, int mode, float subvolume_size, const unsigned long long* __restrict__ sub_keys, int S,
                                                    const double* __restrict__ sub_sh, uchar4* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.N) return;
    const size_t N = g.N;
    constexpr int RING[6] = {NB_PX, NB_MX, NB_PY, NB_MY, NB_PZ, NB_MZ};
    VisStencil v;
    v.valid[0] = g.weight[s] > 0.0f; v.sdf[0] = (float)g.x_sdf[s];
    const uchar4 c = g.color[s]; v.color[0] = c.x; v.color[1] = c.y; v.color[2] = c.z;
    v.color_px[0] = v.color_px[1] = v.color_px[2] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int nb = g.nbr[(size_t)RING[i] * N + s];
        v.valid[i + 1] = nb >= 0 && g.weight[nb] > 0.0f;
        v.sdf[i + 1] = nb >= 0 ? (float)g.x_sdf[nb] : 0.0f;
        if (i == 0 && nb >= 0) { const uchar4 cn = g.color[nb]; v.color_px[0] = cn.x; v.color_px[1] = cn.y; v.color_px[2] = cn.z; }
    }
    v.albedo = g.x_alb[s];
#pragma unroll
    for (int j = 0; j < 9; ++j) v.sh[j] = 0.0f;
    if (vis_mode_needs_sh(mode) && S > 0)              // voxelToWorld = float(i) * voxel_size (sparse_voxel_grid.cpp:224-228)
        vis_interpolate_sh((float)g.cx[s] * g.voxel_size, (float)g.cy[s] * g.voxel_size, (float)g.cz[s] * g.voxel_size, subvolume_size, sub_keys, S, sub_sh, v.sh);
    if (mode == VIS_INTENSITY_GRAD && v.valid[1] && v.valid[2] && v.valid[3] && v.valid[4] && v.valid[5] && v.valid[6]) vis_lum_grad_px(VisGridDev{g}, (long long)s, v.color_px);
    unsigned char o[3];
    vis_color(mode, v, g.truncation, o);
    out[s] = make_uchar4(o[0], o[1], o[2], 0);
}
void launch_vis_colors(hipStream_t st, GridView g, int mode, float subvolume_size, const unsigned long long* sub_keys, int S, const double* sub_sh, uchar4* out) {
    if (g.N > 0) k_vis_colors<<<(g.N + 255) / 256, 256, 0, st>>>(g, mode, subvolume_size, sub_keys, S, sub_sh, out);
}

}  // namespace i3d
