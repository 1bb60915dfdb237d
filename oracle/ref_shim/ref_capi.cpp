// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C ABI over bodies of the REFERENCE ITSELF: every `#include "gen/*.inc"` below is a line range cut out of /root/reference by
// oracle/extract_ref.py at build time (never committed).  What is ours in this file: the stand-in declarations that the absent
// third-party headers / the reference's un-extractable class declarations would have provided (marked "shim"), and the extern "C"
// wrappers at the end.  The library is the second opinion tests/test_oracle_vs_ref.py holds the restated oracle against.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>
#include <omp.h>
#include "mini_eigen.hpp"
#include "mini_ceres.hpp"

#include "gen/invalid_residual.inc"

namespace cv {                                       // shim: the one cv::Mat operation the extracted bodies use (row-major float image)
struct Mat {
    int rows = 0, cols = 0; const float* p = nullptr;
    template <class T> const T& at(int y, int x) const { return reinterpret_cast<const T*>(p)[(size_t)y * cols + x]; }
};
}  // namespace cv

namespace nv {                                       // shim: the typedef names of nv/mat.h:47-86 over the stand-in Matrix
typedef Eigen::Vector2d Vec2; typedef Eigen::Vector3d Vec3; typedef Eigen::Vector4d Vec4;
typedef Eigen::Matrix<double, 5, 1> Vec5; typedef Eigen::Matrix<double, 6, 1> Vec6;
typedef Eigen::Vector2f Vec2f; typedef Eigen::Vector3f Vec3f; typedef Eigen::Vector4f Vec4f;
typedef Eigen::Matrix<float, 5, 1> Vec5f; typedef Eigen::Matrix<float, 6, 1> Vec6f;
typedef Eigen::Matrix3f Mat3f; typedef Eigen::Matrix4f Mat4f;
typedef Eigen::Vector2i Vec2i; typedef Eigen::Vector3i Vec3i; typedef Eigen::Vector4i Vec4i; typedef Eigen::Matrix<int, 6, 1> Vec6i;
typedef Eigen::Matrix<unsigned char, 3, 1> Vec3b;
#include "gen/mat_round.inc"
}  // namespace nv

namespace std {
#include "gen/mat_hash.inc"
}  // namespace std

namespace nv {

#include "gen/grid_voxels.inc"

template <class T>
class SparseVoxelGrid {                              // shim: the members of sparse_voxel_grid.h:83-165 whose DEFINITIONS are extracted below
public:
    SparseVoxelGrid(float voxel_size, float depth_min, float depth_max);
    typedef typename std::unordered_map<Vec3i, T, std::hash<Vec3i>>::iterator iterator;
    typedef typename std::unordered_map<Vec3i, T, std::hash<Vec3i>>::const_iterator const_iterator;
    iterator begin() { return data_.begin(); }
    iterator end() { return data_.end(); }
    const_iterator begin() const { return data_.begin(); }
    const_iterator end() const { return data_.end(); }
    float voxelSize() const { return voxel_size_; }
    float truncation() const { return truncation_; }
    T& voxel(const Vec3i& voxel_pos); T& voxel(const Vec3f& world_pos); T& voxel(int x, int y, int z);
    const T& voxel(const Vec3i& voxel_pos) const; const T& voxel(const Vec3f& world_pos) const; const T& voxel(int x, int y, int z) const;
    Vec3i worldToVoxel(const Vec3f& p) const; Vec3f worldToVoxelFloat(const Vec3f& p) const; Vec3f voxelToWorld(const Vec3i& v) const;
    bool exists(int x, int y, int z) const; bool exists(const Vec3i& voxel_pos) const;
    bool valid(int x, int y, int z) const; bool valid(const Vec3i& voxel_pos) const;
    size_t numVoxels() const; void setVoxel(const Vec3i& voxel_pos, const T& voxel);
    bool empty() const; void clear(); bool remove(const Vec3i& voxel_pos);
private:
    std::unordered_map<Vec3i, T, std::hash<Vec3i>> data_;
    float voxel_size_, depth_min_, depth_max_, truncation_, integration_weight_sample_;
    Vec6f clip_bounds_;
};
#include "gen/grid_ctor.inc"
#include "gen/grid_access.inc"

namespace SDFOperators {
#include "gen/operators_templates.inc"
#include "gen/operators_sdf_weight.inc"
}  // namespace SDFOperators

namespace math {
float robustKernel(float val, float thres = 2.0f);   // shim: the declaration of math.h:47 (default argument)
#include "gen/math_robust_kernel.inc"
}  // namespace math

namespace Shading {
#include "gen/shading_basis.inc"
#include "gen/shading_compute.inc"
#include "gen/shading_graddiff.inc"
}  // namespace Shading

#include "gen/camera_t.inc"

class Camera {                                       // shim: the data members Camera::project reads (camera.h:83-88)
public:
    bool project(const Vec3f& pt, Vec2f& pt2f, Vec2i& pt2i) const;
    Mat3f K_; int width_; int height_; Vec5f dist_coeffs_;
};
#include "gen/camera_project_f.inc"

struct VoxelResidual;                                // (named by nothing we extract)
#include "gen/cost_helpers.inc"
#include "gen/shading_cost_data.inc"

class ShadingCost {                                  // shim: constructor + members of shading_cost.h:78-83,200-203; operator() is the reference's
public:
    ShadingCost(const Vec3i& v_pos, const Eigen::VectorXd& sh_coeffs, const ShadingCostData* data) : v_pos_(v_pos), sh_coeffs_(sh_coeffs), data_(data) {}
#include "gen/shading_cost_functor.inc"
private:
    Vec3i v_pos_;
    const Eigen::VectorXd& sh_coeffs_;
    const ShadingCostData* data_;
};

class VolumetricRegularizer {
public:
#include "gen/volreg_functor.inc"
};
class SurfaceStabRegularizer {
public:
    explicit SurfaceStabRegularizer(double sdf) : sdf_(sdf) {}
#include "gen/stab_functor.inc"
private:
    double sdf_;
};
class AlbedoRegularizer {
public:
#include "gen/albedo_functor.inc"
};

#include "gen/color_intensity.inc"
static double chroma_weight(const Vec3b& color, const Vec3b& color_nb) {     // shim: the two voxels the extracted lines read
    struct { Vec3b color; } v{color}, v_nb{color_nb};
#include "gen/albedo_chroma.inc"
    return w;
}

#include "gen/sh_costs.inc"

#include "gen/vertex_observation.inc"
#include "gen/vertex_observation_lt.inc"
class SDFColorization {                              // shim: declarations of the four member functions extracted below
public:
#include "gen/colorization_config.inc"
    static void filter(std::vector<VertexObservation>& observations, size_t n);
    bool isVoxelVisible(const Vec3f& pt, const cv::Mat& depth, int x, int y) const;
    float computeWeight(const cv::Mat& depth, const Vec3f& n, const int x, const int y, const Vec3f& v) const;
    Vec3f computeColor(const std::vector<VertexObservation>& verts_obs) const;
    Config cfg_;
};
#include "gen/colorization_weights.inc"

#include "gen/mesh_struct.inc"
#include "gen/mesh_save.inc"
namespace MeshUtil {
#include "gen/mesh_degenerate.inc"
}  // namespace MeshUtil
#define private public                               /* shim: the tables are private statics; ref_mc_tables() reads them */
#include "gen/mc_class.inc"
#undef private
#include "gen/mc_extract_mesh.inc"
#include "gen/mc_body.inc"
#include "gen/mc_tables.inc"

}  // namespace nv

// ---------------------------------------------------------------------------------------------------------------- C ABI (ours)
using namespace nv;
typedef ceres::Jet<double, 29> Jet29;

extern "C" {

uint64_t ref_hash(int32_t x, int32_t y, int32_t z) { return (uint64_t)std::hash<Vec3i>()(Vec3i(x, y, z)); }
void ref_round3f(const float* v, int32_t* out) { const Vec3i r = nv::round(Vec3f(v[0], v[1], v[2])); out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; }
void ref_round3d(const double* v, int32_t* out) { const Vec3i r = nv::round(Vec3(v[0], v[1], v[2])); out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; }
double ref_sdf_to_weight(double sdf, double truncation) { return SDFOperators::sdfToWeight(sdf, truncation); }
float ref_robust_kernel(float val) { return math::robustKernel(val); }
double ref_varying_lambda(int32_t it, int32_t n, double l0, double l1) { return computeVaryingLambda(it, n, l0, l1); }
double ref_pyramid_scale(int32_t lvl) { return pyramidLevelToScale(lvl); }

// ShadingCost::operator() on one row: value path (T = double) and the 29 partials through Jets, parameter blocks as
// shading_cost.cpp:90-129 orders them (10 sdf, 4 albedo, pose 6, intrinsics 4, distortion 5)
double ref_shading_row(int32_t vx, int32_t vy, int32_t vz, const double* sh9, int32_t rgbd_level, double voxel_size, int32_t w, int32_t h,
                       const float* lum, const double* params29, double* J29, double* value_double_path) {
    Eigen::VectorXd sh(9); for (int i = 0; i < 9; ++i) sh[i] = sh9[i];
    ShadingCostData data(rgbd_level, voxel_size, w, h, lum);
    ShadingCost cost(Vec3i(vx, vy, vz), sh, &data);
    {   // T = double
        const double* blocks[17];
        for (int i = 0; i < 14; ++i) blocks[i] = params29 + i;
        blocks[14] = params29 + 14; blocks[15] = params29 + 20; blocks[16] = params29 + 24;
        double r = 0.0; cost(blocks, &r);
        if (value_double_path) *value_double_path = r;
    }
    Jet29 p[29]; for (int i = 0; i < 29; ++i) p[i] = Jet29(params29[i], i);
    const Jet29* blocks[17];
    for (int i = 0; i < 14; ++i) blocks[i] = p + i;
    blocks[14] = p + 14; blocks[15] = p + 20; blocks[16] = p + 24;
    Jet29 r; cost(blocks, &r);
    if (J29) for (int i = 0; i < 29; ++i) J29[i] = r.v[i];
    return r.a;
}

int32_t ref_project_t(const double* fxfycxcy, const double* dist5, int32_t w, int32_t h, const double* p3, double* p2d) {
    CameraT<double> cam; cam.fx = fxfycxcy[0]; cam.fy = fxfycxcy[1]; cam.cx = fxfycxcy[2]; cam.cy = fxfycxcy[3]; cam.dist_coeffs = dist5; cam.w = w; cam.h = h;
    return cam.project(p3, p2d) ? 1 : 0;
}
int32_t ref_project_f(const float* fxfycxcy, const float* dist5, int32_t w, int32_t h, const float* p3, float* p2f, int32_t* p2i) {
    Camera cam; cam.K_ = Mat3f::Zero(); cam.K_(0, 0) = fxfycxcy[0]; cam.K_(1, 1) = fxfycxcy[1]; cam.K_(0, 2) = fxfycxcy[2]; cam.K_(1, 2) = fxfycxcy[3]; cam.K_(2, 2) = 1.0f;
    cam.width_ = w; cam.height_ = h; for (int i = 0; i < 5; ++i) cam.dist_coeffs_[i] = dist5[i];
    Vec2f a; Vec2i b; const bool ok = cam.project(Vec3f(p3[0], p3[1], p3[2]), a, b);
    p2f[0] = a[0]; p2f[1] = a[1]; p2i[0] = b[0]; p2i[1] = b[1];
    return ok ? 1 : 0;
}
void ref_bicubic(const float* img, int32_t w, int32_t h, double r, double c, double* f, double* dfdr, double* dfdc) {
    typedef ceres::Jet<double, 2> J2;
    const J2 p2d[2] = {J2(c, 1), J2(r, 0)};           // interpolate() evaluates at (row = p2d[1], col = p2d[0])
    J2 out; nv::interpolate(img, w, h, p2d, &out);
    *f = out.a; *dfdr = out.v[0]; *dfdc = out.v[1];
}
void ref_transform_voxel_iso(double voxel_size, const double* pose6, const int32_t* vc, double sdf, const double* n3, double* out3) {
    int c[3] = {vc[0], vc[1], vc[2]};
    transformVoxelIso(voxel_size, pose6, pose6 + 3, c, sdf, n3, out3);
}
void ref_compute_normal(double s, double sx, double sy, double sz, double* n3) { SDFOperators::computeNormal(s, sx, sy, sz, n3); }

// regulariser functors, residual + partials
void ref_volumetric(const double* s7, double* r, double* J7) {
    typedef ceres::Jet<double, 7> J; J p[7]; for (int i = 0; i < 7; ++i) p[i] = J(s7[i], i);
    J out; VolumetricRegularizer f; f(&p[0], &p[1], &p[2], &p[3], &p[4], &p[5], &p[6], &out);
    *r = out.a; for (int i = 0; i < 7; ++i) J7[i] = out.v[i];
}
void ref_surface_stab(double sdf_refined, double sdf0, double* r, double* J1) {
    typedef ceres::Jet<double, 1> J; J p(sdf_refined, 0); J out; SurfaceStabRegularizer f(sdf0); f(&p, &out);
    *r = out.a; *J1 = out.v[0];
}
void ref_albedo_reg(double a, double a_nb, double* r, double* J2) {
    typedef ceres::Jet<double, 2> J; J p0(a, 0), p1(a_nb, 1); J out; AlbedoRegularizer f; f(&p0, &p1, &out);
    *r = out.a; J2[0] = out.v[0]; J2[1] = out.v[1];
}
double ref_chroma_weight(const uint8_t* c3, const uint8_t* cn3) { return chroma_weight(Vec3b(c3[0], c3[1], c3[2]), Vec3b(cn3[0], cn3[1], cn3[2])); }
void ref_sh_data_cost(double luminance, const float* normal3, double albedo, const double* sh9, double* r, double* J9) {
    typedef ceres::Jet<double, 9> J; J p[9]; for (int i = 0; i < 9; ++i) p[i] = J(sh9[i], i);
    J out; SHDataCost f(luminance, Vec3f(normal3[0], normal3[1], normal3[2]), albedo); f(p, &out);
    *r = out.a; for (int i = 0; i < 9; ++i) J9[i] = out.v[i];
}
void ref_sh_reg_cost(const double* sh9a, const double* sh9b, double* r9) { SHRegularizerCost f; f(sh9a, sh9b, r9); }

// observation weights / colours (SDFColorization)
int32_t ref_voxel_visible(float max_occlusion_distance, const float* pt3, int32_t w, int32_t h, const float* depth, int32_t x, int32_t y) {
    SDFColorization c; c.cfg_.max_occlusion_distance = max_occlusion_distance; cv::Mat d; d.rows = h; d.cols = w; d.p = depth;
    return c.isVoxelVisible(Vec3f(pt3[0], pt3[1], pt3[2]), d, x, y) ? 1 : 0;
}
float ref_observation_weight(int32_t w, int32_t h, const float* depth, const float* n3, int32_t x, int32_t y, const float* v3) {
    SDFColorization c; cv::Mat d; d.rows = h; d.cols = w; d.p = depth;
    return c.computeWeight(d, Vec3f(n3[0], n3[1], n3[2]), x, y, Vec3f(v3[0], v3[1], v3[2]));
}
void ref_compute_color(int32_t n, const uint8_t* rgb, const float* weights, float* out3) {
    std::vector<VertexObservation> obs((size_t)n);
    for (int i = 0; i < n; ++i) { obs[i].color = Vec3b(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]); obs[i].weight = weights[i]; obs[i].frame = i; }
    SDFColorization c; const Vec3f col = c.computeColor(obs); out3[0] = col[0]; out3[1] = col[1]; out3[2] = col[2];
}
// filter: weights in frame order -> weights after keeping the best n (others zeroed), and the frame ids in the sorted order
void ref_filter(int32_t count, float* weights, int32_t keep, int32_t* order) {
    std::vector<VertexObservation> obs((size_t)count);
    for (int i = 0; i < count; ++i) { obs[i].weight = weights[i]; obs[i].frame = i; }
    SDFColorization::filter(obs, (size_t)keep);
    for (int i = 0; i < count; ++i) { weights[i] = obs[i].weight; order[i] = obs[i].frame; }
}

// grid: insertion order -> iteration order of the reference's container (constructor reserve(64) / max_load_factor(0.6), its hash)
void ref_grid_visit_order(float voxel_size, int64_t n, const int32_t* keys, int64_t* visit_to_input) {
    SparseVoxelGrid<VoxelSBR> g(voxel_size, 0.1f, 10.0f);
    std::unordered_map<Vec3i, int64_t, std::hash<Vec3i>> idx;
    for (int64_t i = 0; i < n; ++i) { const Vec3i k(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]); g.setVoxel(k, VoxelSBR()); idx[k] = i; }
    int64_t o = 0; for (auto it = g.begin(); it != g.end(); ++it) visit_to_input[o++] = idx[it->first];
}
void ref_world_to_voxel(float voxel_size, const float* p3, int32_t* out3) {
    SparseVoxelGrid<VoxelSBR> g(voxel_size, 0.1f, 10.0f); const Vec3i v = g.worldToVoxel(Vec3f(p3[0], p3[1], p3[2])); out3[0] = v[0]; out3[1] = v[1]; out3[2] = v[2];
}
float ref_truncation(float voxel_size) { SparseVoxelGrid<VoxelSBR> g(voxel_size, 0.1f, 10.0f); return g.truncation(); }

// marching cubes over a VoxelSBR grid built by inserting the records in the given order; returns the merged, cleaned mesh
void* ref_mc_extract(float voxel_size, int64_t n, const int32_t* keys, const double* sdf, const float* weight, const uint8_t* color) {
    SparseVoxelGrid<VoxelSBR> g(voxel_size, 0.1f, 10.0f);
    for (int64_t i = 0; i < n; ++i) {
        VoxelSBR v; v.sdf = sdf[i]; v.sdf_refined = sdf[i]; v.weight = weight[i]; v.color = Vec3b(color[3 * i], color[3 * i + 1], color[3 * i + 2]);
        g.setVoxel(Vec3i(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]), v);
    }
    std::streambuf* old = std::cout.rdbuf(nullptr);     // the reference prints triangle counts
    Mesh* m = MarchingCubes<VoxelSBR>::extractSurface(g);
    std::cout.rdbuf(old);
    return m;
}
void ref_mesh_counts(void* mesh, int64_t* nv, int64_t* nf) { Mesh* m = (Mesh*)mesh; *nv = m ? (int64_t)m->vertices.size() : 0; *nf = m ? (int64_t)m->face_vertices.size() : 0; }
void ref_mesh_get(void* mesh, float* verts, uint8_t* colors, int32_t* faces) {
    Mesh* m = (Mesh*)mesh; if (!m) return;
    for (size_t i = 0; i < m->vertices.size(); ++i) for (int k = 0; k < 3; ++k) { verts[3 * i + k] = m->vertices[i][k]; colors[3 * i + k] = m->colors[i][k]; }
    for (size_t i = 0; i < m->face_vertices.size(); ++i) for (int k = 0; k < 3; ++k) faces[3 * i + k] = m->face_vertices[i][k];
}
int32_t ref_mesh_save(void* mesh, const char* path) { Mesh* m = (Mesh*)mesh; return (m && m->save(path)) ? 1 : 0; }
void ref_mesh_free(void* mesh) { delete (Mesh*)mesh; }
// the two tables, for the case-by-case check of the product's packed copy
void ref_mc_tables(int32_t* edge256, int32_t* tri256x16) {
    for (int i = 0; i < 256; ++i) { edge256[i] = MarchingCubes<VoxelSBR>::edge_table_[i]; for (int k = 0; k < 16; ++k) tri256x16[16 * i + k] = MarchingCubes<VoxelSBR>::triangle_table_[i][k]; }
}

}  // extern "C"
