// Mesh export of the resident grid: MarchingCubes<VoxelSBR>::extractSurface -> merge -> MeshUtil::removeDegenerateFaces
// [-> MeshUtil::removeLooseComponents] -> Mesh::save  (mesh/marching_cubes.cpp:65-155, mesh/util.cpp:47-200, mesh/mesh.cpp:41-100),
// as driven by SDFVisualization::exportMesh (sdf/visualization.cpp:167-196).  Triangles are produced on the device
// (mesh_kernels.hip); unification, cleaning and the PLY stream are host work.
//
// Triangulation table: Bourke's 256-case table (the one the reference carries, marching_cubes.cpp:330-623) in the packed form of
// mc_table.hpp, so cells, triangles, faces and therefore the PLY stream are identical to the reference's, not just the vertex set.
#include "context.hpp"
#include "../device/level_kernels.hpp"
#include "../device/vis_colors.hpp"
#include "mc_table.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <fstream>
#include <numeric>
#include <unordered_map>

namespace i3d {
namespace {

constexpr int MC_STRIDE = 16;              // up to 5 triangles per configuration (15 edge ids), -1 padded like the reference's rows
struct McTables { unsigned char ntri[256]; signed char tri[256 * MC_STRIDE]; int max_tri = 0; bool ready = false; };

void build_tables(McTables& T) {           // unpack mc_table.hpp into the byte table the kernels read
    for (int idx = 0; idx < 256; ++idx) {
        const int nt = mc_num_triangles(idx);
        T.ntri[idx] = (unsigned char)nt; T.max_tri = std::max(T.max_tri, nt);
        for (int k = 0; k < MC_STRIDE; ++k) T.tri[idx * MC_STRIDE + k] = k < 3 * nt ? (signed char)mc_edge(idx, k) : (signed char)-1;
    }
    T.ready = true;
}

const McTables& tables() { static McTables T; if (!T.ready) build_tables(T); return T; }

struct F3 { float x, y, z; bool operator==(const F3& o) const { return std::memcmp(this, &o, sizeof(F3)) == 0 || (x == o.x && y == o.y && z == o.z); } };
struct F3Hash { size_t operator()(const F3& v) const {
    auto b = [](float f) { if (f == 0.0f) f = 0.0f; uint32_t u; std::memcpy(&u, &f, 4); return (size_t)u; };       // -0 and +0 are the same key, like operator<
    return (b(v.x) * 73856093u) ^ (b(v.y) * 19349669u) ^ (b(v.z) * 83492791u); } };

}  // namespace

struct MeshData { std::vector<float> vertices; std::vector<uint8_t> colors; std::vector<int32_t> faces; };
static thread_local MeshData g_mesh;       // the last extracted mesh of this host thread (i3d_get_mesh)

// MeshUtil::removeLooseComponents + removeUnusedVertices (mesh/util.cpp:47-171): faces that share a vertex position are connected; only the largest
// connected component is kept — boost numbers components in the order of their first face and std::max_element keeps the first maximum — then the
// vertices no face uses are dropped and the indices renumbered in vertex order.  (Vertices are unique per position after merge(), so sharing an index is
// sharing a position.)
// canon (optional): canonical vertex of every vertex — the first vertex at the same POSITION.  The reference connects faces through a position-keyed
// vertex map (mesh/util.cpp:52-62); the internal path has unique positions after merge(), caller-supplied meshes (a triangle soup) may not.
static void remove_loose_components(MeshData& M, const std::vector<int32_t>* canon = nullptr) {
    std::vector<int32_t>& faces = M.faces;
    if (faces.empty()) return;
    const size_t nf = faces.size() / 3, nv = M.vertices.size() / 3;
    std::vector<int> parent(nf); std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    std::vector<int> first_face(nv, -1);
    for (size_t f = 0; f < nf; ++f) for (int k = 0; k < 3; ++k) {
        const int v = canon ? (*canon)[(size_t)faces[3 * f + k]] : faces[3 * f + k];
        if (first_face[v] < 0) first_face[v] = (int)f;
        else { int a = find(first_face[v]), b = find((int)f); if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; } }      // root = smallest face index
    }
    std::vector<int> size(nf, 0);
    for (size_t f = 0; f < nf; ++f) size[find((int)f)]++;
    int best = -1;
    for (size_t f = 0; f < nf; ++f) if (parent[f] == (int)f && (best < 0 || size[f] > size[best])) best = (int)f;
    std::vector<int32_t> out; out.reserve(3 * (size_t)size[best]);
    for (size_t f = 0; f < nf; ++f) if (find((int)f) == best) { out.push_back(faces[3 * f]); out.push_back(faces[3 * f + 1]); out.push_back(faces[3 * f + 2]); }
    faces.swap(out);
    // removeUnusedVertices
    std::vector<char> used(nv, 0);
    for (int32_t v : faces) used[(size_t)v] = 1;
    size_t kept = 0; for (size_t v = 0; v < nv; ++v) kept += used[v] ? 1 : 0;
    if (kept == nv) return;
    const bool has_colors = !M.colors.empty();
    std::vector<int32_t> remap(nv, 0); std::vector<float> verts; std::vector<uint8_t> cols; verts.reserve(3 * kept); if (has_colors) cols.reserve(3 * kept);
    int32_t idx = 0;
    for (size_t v = 0; v < nv; ++v) if (used[v]) {
        verts.push_back(M.vertices[3 * v]); verts.push_back(M.vertices[3 * v + 1]); verts.push_back(M.vertices[3 * v + 2]);
        if (has_colors) { cols.push_back(M.colors[3 * v]); cols.push_back(M.colors[3 * v + 1]); cols.push_back(M.colors[3 * v + 2]); }
        remap[v] = idx++;
    }
    M.vertices.swap(verts); M.colors.swap(cols);
    for (int32_t& v : faces) v = remap[(size_t)v];
}

int extract_mesh(i3d_context* c, int use_refined, int color_mode, int largest_only, MeshData& M, int64_t* raw_triangles) {
    if (!c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "extract_mesh: no grid");
    if (color_mode < 0 || color_mode >= VIS_NUM_MODES) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "extract_mesh: unknown colour mode");
    CTX_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream; const int N = c->N;
    const McTables& T = tables();
    GridView g = c->grid_view(); HashTable ht{c->hkeys.p, c->hvals.p, c->hmask};
    DevBuf<int> inv, counts, offs; DevBuf<unsigned char> d_ntri; DevBuf<signed char> d_tri;
    CTX_HIP(c, inv.alloc(N)); CTX_HIP(c, counts.alloc(N)); CTX_HIP(c, offs.alloc(N)); CTX_HIP(c, d_ntri.alloc(256)); CTX_HIP(c, d_tri.alloc(256 * MC_STRIDE));
    CTX_HIP(c, hipMemcpyAsync(d_ntri.p, T.ntri, 256, hipMemcpyHostToDevice, st));
    CTX_HIP(c, hipMemcpyAsync(d_tri.p, T.tri, 256 * MC_STRIDE, hipMemcpyHostToDevice, st));
    launch_inv_rank(st, N, c->rank.p, inv.p);
    launch_mc_count(st, g, ht, inv.p, use_refined, d_ntri.p, counts.p);
    CTX_HIP(c, rocprim::exclusive_scan(c->scan_tmp.p, c->scan_tmp_bytes, counts.p, offs.p, 0, (size_t)N, rocprim::plus<int>(), st));
    int tail[2];
    CTX_HIP(c, hipMemcpyAsync(&tail[0], offs.p + (N - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipMemcpyAsync(&tail[1], counts.p + (N - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipStreamSynchronize(st));
    const size_t nt = (size_t)tail[0] + (size_t)tail[1];
    if (raw_triangles) *raw_triangles = (int64_t)nt;
    M.vertices.clear(); M.colors.clear(); M.faces.clear();
    if (nt == 0) return I3D_OK;                                      // extractMesh returns nullptr
    DevBuf<float> d_pos; DevBuf<unsigned char> d_col;
    CTX_HIP(c, d_pos.alloc(nt * 9)); CTX_HIP(c, d_col.alloc(nt * 9));
    // colour modes >= 2 (SDFVisualization::colorize's debug views): paint every voxel first, the emit kernel interpolates what was painted
    DevBuf<uchar4> d_mode; DevBuf<unsigned long long> d_svk; DevBuf<double> d_svs;
    if (color_mode >= 2) {
        int S = 0;
        if (vis_mode_needs_sh(color_mode)) {                                              // applyColorShading without subvolumes paints nothing (:322-323): refuse instead
            if (!c->have_subvolumes || c->sv_keys.empty()) return ctx_fail(c, I3D_ERR_STATE, "extract_mesh: the shading colour modes need a lighting estimate (i3d_estimate_sh / i3d_refine)");
            S = (int)c->sv_keys.size();
            CTX_HIP(c, d_svk.alloc(S)); CTX_HIP(c, d_svs.alloc((size_t)S * 9));
            CTX_HIP(c, hipMemcpyAsync(d_svk.p, c->sv_keys.data(), sizeof(unsigned long long) * S, hipMemcpyHostToDevice, st));
            CTX_HIP(c, hipMemcpyAsync(d_svs.p, c->sv_sh.data(), sizeof(double) * (size_t)S * 9, hipMemcpyHostToDevice, st));
        }
        CTX_HIP(c, d_mode.alloc(N));
        launch_vis_colors(st, g, color_mode, c->sv_size, d_svk.p, S, d_svs.p, d_mode.p);
    }
    launch_mc_emit(st, g, ht, inv.p, use_refined, color_mode, d_ntri.p, d_tri.p, MC_STRIDE, offs.p, d_pos.p, d_col.p, d_mode.p);
    std::vector<float> pos(nt * 9); std::vector<uint8_t> col(nt * 9);
    CTX_HIP(c, hipMemcpyAsync(pos.data(), d_pos.p, sizeof(float) * nt * 9, hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipMemcpyAsync(col.data(), d_col.p, nt * 9, hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipStreamSynchronize(st));
    // merge(): vertices with the same position are one vertex; index = order of first appearance (marching_cubes.cpp:98-152)
    std::unordered_map<F3, int, F3Hash> index; index.reserve(nt * 2);
    std::vector<int32_t> faces; faces.reserve(nt * 3);
    for (size_t i = 0; i < nt * 3; ++i) {
        const F3 p{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        auto it = index.find(p);
        int id;
        if (it == index.end()) { id = (int)(M.vertices.size() / 3); index.emplace(p, id);
            M.vertices.push_back(p.x); M.vertices.push_back(p.y); M.vertices.push_back(p.z);
            M.colors.push_back(col[3 * i]); M.colors.push_back(col[3 * i + 1]); M.colors.push_back(col[3 * i + 2]); }
        else id = it->second;
        faces.push_back(id);
    }
    // removeDegenerateFaces (mesh/util.cpp:174-200)
    std::vector<int32_t> kept; kept.reserve(faces.size());
    for (size_t f = 0; f < nt; ++f) {
        const int v0 = faces[3 * f], v1 = faces[3 * f + 1], v2 = faces[3 * f + 2];
        if (v0 == v1 || v0 == v2 || v1 == v2) continue;
        const float* a = &M.vertices[3 * (size_t)v0]; const float* b = &M.vertices[3 * (size_t)v1]; const float* cc = &M.vertices[3 * (size_t)v2];
        const float e0[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]}, e1[3] = {cc[0] - b[0], cc[1] - b[1], cc[2] - b[2]};
        const float cr[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
        const double area = (double)std::sqrt(cr[0] * cr[0] + (cr[1] * cr[1] + cr[2] * cr[2]));
        if (area == 0.0 || std::isnan(area) || std::isinf(area)) continue;
        kept.push_back(v0); kept.push_back(v1); kept.push_back(v2);
    }
    faces.swap(kept);
    M.faces.swap(faces);
    if (largest_only) remove_loose_components(M);
    return I3D_OK;
}

}  // namespace i3d

using namespace i3d;

extern "C" {

// MeshUtil::removeLooseComponents (mesh/util.cpp:47-101) on caller arrays, in place: the largest connected component, unused vertices dropped.
// colors may be NULL.  Host only.
int i3d_mesh_remove_loose_components(int64_t* num_vertices, float* vertices, uint8_t* colors, int64_t* num_faces, int32_t* faces) {
    if (!num_vertices || !num_faces || *num_vertices < 0 || *num_faces < 0 || (*num_vertices > 0 && !vertices) || (*num_faces > 0 && !faces)) return I3D_ERR_INVALID_ARGUMENT;
    for (int64_t i = 0; i < 3 * *num_faces; ++i) if (faces[i] < 0 || faces[i] >= *num_vertices) return I3D_ERR_INVALID_ARGUMENT;
    MeshData M; M.vertices.assign(vertices, vertices + 3 * *num_vertices); if (colors) M.colors.assign(colors, colors + 3 * *num_vertices); M.faces.assign(faces, faces + 3 * *num_faces);
    std::vector<int32_t> canon((size_t)*num_vertices);
    { std::unordered_map<F3, int, F3Hash> first; first.reserve((size_t)*num_vertices * 2);
      for (int64_t v = 0; v < *num_vertices; ++v) canon[(size_t)v] = first.emplace(F3{vertices[3 * v], vertices[3 * v + 1], vertices[3 * v + 2]}, (int)v).first->second; }
    remove_loose_components(M, &canon);
    *num_vertices = (int64_t)(M.vertices.size() / 3); *num_faces = (int64_t)(M.faces.size() / 3);
    std::copy(M.vertices.begin(), M.vertices.end(), vertices); if (colors) std::copy(M.colors.begin(), M.colors.end(), colors); std::copy(M.faces.begin(), M.faces.end(), faces);
    return I3D_OK;
}

// Mesh::save (mesh/mesh.cpp:41-100): binary little-endian PLY, float positions, optional uchar colours, "uchar int" face lists
int i3d_write_ply(const char* path, int64_t num_vertices, const float* vertices, const uint8_t* colors, int64_t num_faces, const int32_t* faces) {
    if (!path || num_vertices <= 0 || !vertices || (num_faces > 0 && !faces)) return I3D_ERR_INVALID_ARGUMENT;     // empty meshes are not saved
    std::ofstream f(path, std::ios::binary); if (!f.is_open()) return I3D_ERR_IO;
    f << "ply\n" << "format binary_little_endian 1.0\n" << "element vertex " << (int)num_vertices << "\n"
      << "property float x\n" << "property float y\n" << "property float z\n";
    if (colors) f << "property uchar red\n" << "property uchar green\n" << "property uchar blue\n";
    f << "element face " << (int)num_faces << "\n" << "property list uchar int vertex_indices\n" << "end_header\n";
    for (int64_t i = 0; i < num_vertices; ++i) { f.write((const char*)&vertices[3 * i], 12); if (colors) f.write((const char*)&colors[3 * i], 3); }
    const unsigned char three = 3;
    for (int64_t i = 0; i < num_faces; ++i) { f.write((const char*)&three, 1); f.write((const char*)&faces[3 * i], 12); }
    return f.good() ? I3D_OK : I3D_ERR_IO;
}

int i3d_extract_mesh(i3d_context* c, int32_t use_refined_sdf, int32_t color_mode, int32_t largest_component_only, int64_t* num_vertices, int64_t* num_faces) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    const int rc = extract_mesh(c, use_refined_sdf, color_mode, largest_component_only, g_mesh, nullptr);
    if (rc) return rc;
    if (num_vertices) *num_vertices = (int64_t)(g_mesh.vertices.size() / 3);
    if (num_faces) *num_faces = (int64_t)(g_mesh.faces.size() / 3);
    return I3D_OK;
}
int i3d_get_mesh(i3d_context* c, float* vertices, uint8_t* colors, int32_t* faces) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    if (vertices) std::memcpy(vertices, g_mesh.vertices.data(), g_mesh.vertices.size() * sizeof(float));
    if (colors) std::memcpy(colors, g_mesh.colors.data(), g_mesh.colors.size());
    if (faces) std::memcpy(faces, g_mesh.faces.data(), g_mesh.faces.size() * sizeof(int32_t));
    return I3D_OK;
}
// SDFVisualization::exportMesh for one colour mode: extract (+ largest component) + save
int i3d_export_mesh_ply(i3d_context* c, const char* path, int32_t use_refined_sdf, int32_t color_mode, int32_t largest_component_only) {
    if (!c || !path) return I3D_ERR_INVALID_ARGUMENT;
    MeshData M;
    const int rc = extract_mesh(c, use_refined_sdf, color_mode, largest_component_only, M, nullptr);
    if (rc) return rc;
    if (M.vertices.empty()) return ctx_fail(c, I3D_ERR_STATE, "i3d_export_mesh_ply: mesh could not be generated (no iso-surface)");
    return i3d_write_ply(path, (int64_t)(M.vertices.size() / 3), M.vertices.data(), M.colors.data(), (int64_t)(M.faces.size() / 3), M.faces.data());
}
// SDFVisualization::applyColor* (sdf/visualization.cpp:228-416) on caller arrays: the colour every voxel gets in a colour mode (I3D_COLOR_*).  The voxels may come
// in any order; visit_rank (or NULL: the array order) is the position of every voxel in the reference's walk over the grid, which only "lum_grad" depends on.
// Host instantiation of the function the export kernel runs (device/vis_colors.hpp); neighbours through a map over the keys.
int i3d_visualization_colors(int32_t color_mode, float voxel_size, int64_t n, const int32_t* keys, const double* sdf_refined, const double* albedo, const float* weight,
                             const uint8_t* color, const int64_t* visit_rank, float subvolume_size, int32_t num_subvolumes, const int32_t* subvolume_index, const double* subvolume_sh,
                             uint8_t* color_out) {
    if (color_mode < 0 || color_mode >= VIS_NUM_MODES || n < 0 || !keys || !sdf_refined || !albedo || !weight || !color || !color_out) return I3D_ERR_INVALID_ARGUMENT;
    if (vis_mode_needs_sh(color_mode) && (num_subvolumes <= 0 || !subvolume_index || !subvolume_sh)) return I3D_ERR_INVALID_ARGUMENT;
    std::unordered_map<unsigned long long, int64_t> at; at.reserve((size_t)n * 2);
    for (int64_t i = 0; i < n; ++i) at.emplace(vis_pack3(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]), i);
    std::vector<unsigned long long> sk; std::vector<double> ss;
    if (vis_mode_needs_sh(color_mode)) {                                   // the interpolation looks subvolumes up by packed index: sort them
        std::vector<int> order((size_t)num_subvolumes); std::iota(order.begin(), order.end(), 0);
        std::vector<unsigned long long> raw((size_t)num_subvolumes);
        for (int i = 0; i < num_subvolumes; ++i) raw[(size_t)i] = vis_pack3(subvolume_index[3 * i], subvolume_index[3 * i + 1], subvolume_index[3 * i + 2]);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return raw[(size_t)a] < raw[(size_t)b]; });
        sk.resize((size_t)num_subvolumes); ss.resize((size_t)num_subvolumes * 9);
        for (int i = 0; i < num_subvolumes; ++i) { sk[(size_t)i] = raw[(size_t)order[(size_t)i]]; std::memcpy(&ss[(size_t)i * 9], subvolume_sh + (size_t)order[(size_t)i] * 9, 9 * sizeof(double)); }
    }
    const float truncation = voxel_size * 5.0f;                            // sparse_voxel_grid.cpp:48
    const int off[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    auto find = [&](int x, int y, int z) -> int64_t { const auto it = at.find(vis_pack3(x, y, z)); return it == at.end() ? -1 : it->second; };
    struct HostGrid {                                                      // the accessors vis_lum_grad_px walks the arrays with
        const int32_t* keys; const float* weight; const uint8_t* col; const int64_t* visit_rank; const decltype(find)& f; const int (*off)[3];
        long long px(long long i) const { return f(keys[3 * i] + 1, keys[3 * i + 1], keys[3 * i + 2]); }
        long long mx(long long i) const { return f(keys[3 * i] - 1, keys[3 * i + 1], keys[3 * i + 2]); }
        long long rank(long long i) const { return visit_rank ? visit_rank[i] : i; }
        bool ring(long long i) const { bool ok = true; for (int d = 0; d < 6; ++d) { const int64_t nb = f(keys[3 * i] + off[d][0], keys[3 * i + 1] + off[d][1], keys[3 * i + 2] + off[d][2]); ok = ok && nb >= 0 && weight[nb] > 0.0f; } return ok; }
        void color(long long i, unsigned char c[3]) const { for (int k = 0; k < 3; ++k) c[k] = col[3 * i + k]; }
    } hg{keys, weight, color, visit_rank, find, off};
    for (int64_t i = 0; i < n; ++i) {
        const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
        VisStencil v;
        v.valid[0] = weight[i] > 0.0f; v.sdf[0] = (float)sdf_refined[i];
        for (int k = 0; k < 3; ++k) { v.color[k] = color[3 * i + k]; v.color_px[k] = 0; }
        for (int d = 0; d < 6; ++d) {
            const int64_t nb = find(x + off[d][0], y + off[d][1], z + off[d][2]);
            v.valid[d + 1] = nb >= 0 && weight[nb] > 0.0f;
            v.sdf[d + 1] = nb >= 0 ? (float)sdf_refined[nb] : 0.0f;
            if (d == 0 && nb >= 0) for (int k = 0; k < 3; ++k) v.color_px[k] = color[3 * nb + k];
        }
        v.albedo = albedo[i];
        if (color_mode == VIS_INTENSITY_GRAD && v.valid[1] && v.valid[2] && v.valid[3] && v.valid[4] && v.valid[5] && v.valid[6]) vis_lum_grad_px(hg, (long long)i, v.color_px);
        for (int j = 0; j < 9; ++j) v.sh[j] = 0.0f;
        if (vis_mode_needs_sh(color_mode))
            vis_interpolate_sh((float)x * voxel_size, (float)y * voxel_size, (float)z * voxel_size, subvolume_size, sk.data(), num_subvolumes, ss.data(), v.sh);
        vis_color(color_mode, v, truncation, color_out + 3 * i);
    }
    return I3D_OK;
}
// the triangulation table the kernels use, for inspection / tests: ntri[256], tri[256][16] (edge ids, -1 padded); returns max triangles per cell
int i3d_mc_tables(uint8_t* ntri, int8_t* tri) {
    const McTables& T = tables();
    if (ntri) std::memcpy(ntri, T.ntri, 256);
    if (tri) std::memcpy(tri, T.tri, 256 * MC_STRIDE);
    return T.max_tri;
}

}  // extern "C"
