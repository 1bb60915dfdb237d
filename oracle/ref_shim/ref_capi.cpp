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
#include <unordered_set>
#include <cassert>
#include <ctime>
#include <limits>
#include <sstream>
#include <iomanip>
#include <type_traits>
#include <vector>
#include <omp.h>
#include "mini_eigen.hpp"
#include "mini_cv.hpp"
#include "mini_boost.hpp"
#include "mini_ceres.hpp"

#include "gen/invalid_residual.inc"

namespace nv {                                       // shim: the typedef names of nv/mat.h:47-86 over the stand-in Matrix
typedef Eigen::Vector2d Vec2; typedef Eigen::Vector3d Vec3; typedef Eigen::Vector4d Vec4;
typedef Eigen::Matrix<double, 5, 1> Vec5; typedef Eigen::Matrix<double, 6, 1> Vec6;
typedef Eigen::Matrix3d Mat3; typedef Eigen::Matrix4d Mat4;
typedef Eigen::Vector2f Vec2f; typedef Eigen::Vector3f Vec3f; typedef Eigen::Vector4f Vec4f;
typedef Eigen::Matrix<float, 5, 1> Vec5f; typedef Eigen::Matrix<float, 6, 1> Vec6f;
typedef Eigen::Matrix3f Mat3f; typedef Eigen::Matrix4f Mat4f;
typedef Eigen::Vector2i Vec2i; typedef Eigen::Vector3i Vec3i; typedef Eigen::Vector4i Vec4i; typedef Eigen::Matrix<int, 6, 1> Vec6i;
typedef Eigen::Matrix<unsigned char, 3, 1> Vec3b;
#include "gen/mat_round.inc"
#include "gen/mat_floor_ceil.inc"
}  // namespace nv

namespace std {
#include "gen/mat_hash.inc"
}  // namespace std

#define private public                               /* shim: the C ABI below fills / reads private members instead of going through files */
#define protected public
namespace nv {

#include "gen/settings_class.inc"                    // the reference's Settings (string map, stream conversions); reading a yml needs cv::FileStorage:
#include "gen/settings_impl.inc"                     // our load() below refuses, the C ABI fills the map through set()
bool Settings::load(const std::string&) { return false; }

#include "gen/grid_voxels.inc"
#include "gen/camera_class.inc"
#include "gen/camera_t.inc"
#include "gen/grid_class.inc"
#include "gen/math_decl.inc"
#include "gen/color_util_decl.inc"
#include "gen/processing_decl.inc"
namespace SDFOperators {
#include "gen/operators_templates.inc"
}  // namespace SDFOperators
#include "gen/algorithms_decl.inc"

#include "gen/camera_impl.inc"
#include "gen/camera_convert.inc"
#include "gen/camera_load_save.inc"
namespace math {
#include "gen/math_impl.inc"
#include "gen/math_pose_mat_to_vec.inc"
}  // namespace math
#include "gen/grid_impl.inc"
#include "gen/grid_frustum.inc"
#include "gen/grid_print_info.inc"
#include "gen/grid_save.inc"
#include "gen/grid_clone.inc"
#include "gen/grid_load.inc"
namespace SDFOperators {
float laplacian(const SparseVoxelGrid<VoxelSBR>* grid, const Vec3i& v_pos);            // (operators.h:88,111: declared in the header, used by SDFVisualization)
Vec3f intensityGradient(const SparseVoxelGrid<VoxelSBR>* grid, const Vec3i& v_pos);
#include "gen/operators_impl.inc"
#include "gen/operators_lap_grad.inc"
#include "gen/operators_sdf_weight.inc"
}  // namespace SDFOperators
#include "gen/color_intensity.inc"
#include "gen/color_chroma.inc"
#include "gen/color_scalar.inc"
#include "gen/color_random.inc"
#include "gen/processing_impl.inc"
#include "gen/pyramid_class.inc"
#include "gen/pyramid_ctor.inc"
#include "gen/pyramid_ctor2.inc"
#include "gen/pyramid_dtor.inc"
#include "gen/pyramid_create.inc"
#include "gen/pyramid_downsample.inc"
#include "gen/pyramid_create_pyr.inc"
#include "gen/pyramid_access.inc"
#include "gen/pyramid_depth_down.inc"
#include "gen/pyramid_depth_pyr.inc"
namespace SDFAlgorithms {
#include "gen/algorithms_impl.inc"
}  // namespace SDFAlgorithms

namespace Shading {
#include "gen/shading_decl.inc"
#include "gen/shading_basis.inc"
#include "gen/shading_compute.inc"
#include "gen/shading_graddiff.inc"
#include "gen/shading_basis_f.inc"
float computeShading(const Vec3f &normal, const Eigen::VectorXf &sh_coeffs, float albedo);
#include "gen/shading_compute_f.inc"
}  // namespace Shading

#include "gen/vertex_observation.inc"
#include "gen/vertex_observation_lt.inc"
#include "gen/colorization_class.inc"
#include "gen/colorization_impl.inc"

#include "gen/voxel_residual.inc"
#include "gen/cost_helpers.inc"
#include "gen/shading_cost_data.inc"
#include "gen/shading_cost_class.inc"
#include "gen/shading_cost_impl.inc"
#include "gen/volreg_class.inc"
#include "gen/volreg_impl.inc"
#include "gen/stab_class.inc"
#include "gen/stab_impl.inc"
#include "gen/albedo_class.inc"
#include "gen/albedo_impl.inc"
static double chroma_weight(const Vec3b& color, const Vec3b& color_nb) {     // shim: the two voxels the extracted lines read
    struct { Vec3b color; } v{color}, v_nb{color_nb};
#include "gen/albedo_chroma.inc"
    return w;
}

#include "gen/timer_class.inc"
#include "gen/nls_class.inc"
#include "gen/nls_impl.inc"
#include "gen/optimizer_class.inc"
#include "gen/optimizer_impl.inc"
#include "gen/subvolumes_class.inc"
#include "gen/subvolumes_impl.inc"
#include "gen/svsh_class.inc"
#include "gen/svsh_impl.inc"

// Sensor / SensorI3d are the reference's own classes (folder conventions, pose / intrinsics text, depth scaling and thresholding); PNG decoding comes from
// the harness (cv::imdecode hook).  Intrinsic3D only needs a Sensor to hold the colour camera and take poses: HolderSensor below.
#include "gen/sensor_class.inc"
#include "gen/sensor_ctor.inc"
#include "gen/sensor_access.inc"
#include "gen/sensor_poses.inc"
#include "gen/sensor_i3d_class.inc"
#include "gen/sensor_i3d_impl.inc"
#include "gen/sensor_create.inc"
class HolderSensor : public Sensor {
public:
    void setPose(int, const Mat4f&) override {}
    Mat4f pose(int) override { return Mat4f::Identity(); }
    double timePose(int) override { return 0.0; }
    double timeDepth(int) override { return 0.0; }
    double timeColor(int) override { return 0.0; }
    bool init(const std::string&) override { return false; }
    cv::Mat loadDepth(int) override { return cv::Mat(); }
    cv::Mat loadColor(int) override { return cv::Mat(); }
};
// KeyframeSelection is the reference's own class (window selection, keyframes.txt, the Crete blur metric over the OpenCV stand-ins of mini_cv.hpp).
#include "gen/kfs_class.inc"
#include "gen/kfs_impl_a.inc"
#include "gen/kfs_impl_b.inc"
#include "gen/kfs_draw.inc"
#include "gen/i3d_class.inc"
#include "gen/i3d_callback_dtor.inc"
#include "gen/i3d_cfg_load.inc"
#include "gen/opt_cfg_load.inc"
#include "gen/i3d_ctor.inc"
#include "gen/i3d_refine.inc"
// shim for Intrinsic3D::init (intrinsic3d.cpp:151-203): its keyframe loop needs Sensor + cv::pyrDown / cvtColor; image_model_ is
// filled by the C ABI instead, and what remains is the configuration of the colouriser (:160-166) and the initial recolouring (:196-201)
bool Intrinsic3D::init() {
    sdf_colorization_.reset(opt_data_.grid, sensor_->colorCamera());
    SDFColorization::Config colorizeCfg;
    colorizeCfg.max_occlusion_distance = cfg_.occlusions_distance;
    colorizeCfg.max_num_observations = cfg_.num_observations;
    sdf_colorization_.setConfig(colorizeCfg);
    return recomputeColors();
}
// ... and the reference's OWN init (intrinsic3d.cpp:151-203) under another name, as a member of a derived class: the keyframe loop over its Sensor /
// KeyframeSelection, resizeDepth, Pyramid(num_levels, colour, depth), the pose inverse and math::poseMatToVecAA.  (The wrappers above keep the shim: they
// are handed pyramids, not a sensor.)
class Intrinsic3DInit : public Intrinsic3D {
public:
    using Intrinsic3D::Intrinsic3D;
    bool init_reference();
    // the schedule once more as members of this class, so that refine() runs the reference's own init (the same text, intrinsic3d.cpp:206-409)
    bool refine(SparseVoxelGrid<Voxel>* grid);
    bool prepareGridLevel(int grid_lvl_coarsest); bool finishGridLevel(); bool prepareRgbdLevel(); bool finishRgbdLevel(); bool recomputeColors();
};
#define Intrinsic3D Intrinsic3DInit
#define init init_reference
#include "gen/i3d_init.inc"
#include "gen/i3d_refine.inc"
#undef init
#undef Intrinsic3D

#include "gen/mesh_struct.inc"
#include "gen/mesh_save.inc"
namespace MeshUtil {
void removeUnusedVertices(Mesh* mesh);               // (declared in the reference's mesh/util.h, defined below its first use)
#include "gen/mesh_degenerate.inc"
#include "gen/mesh_loose.inc"
#include "gen/mesh_unused.inc"
}  // namespace MeshUtil
#include "gen/mc_class.inc"
#include "gen/mc_extract_mesh.inc"
#include "gen/mc_body.inc"
#include "gen/mc_tables.inc"
#include "gen/vis_class.inc"
#include "gen/vis_impl.inc"
#include "gen/app_fusion_class.inc"
#include "gen/app_fusion_ctor.inc"
#include "gen/app_fusion_fuse.inc"
#include "gen/app_keyframes_class.inc"
#include "gen/app_keyframes_ctor.inc"
#include "gen/app_keyframes_select.inc"
#include "gen/app_i3d_class.inc"
#include "gen/app_i3d_ctor.inc"
#include "gen/app_i3d_on_refined.inc"

}  // namespace nv
#undef private
#undef protected

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
    Camera cam; SDFColorization c(cam); c.cfg_.max_occlusion_distance = max_occlusion_distance; const cv::Mat d = cv::Mat::wrap(h, w, CV_32FC1, depth);
    return c.isVoxelVisible(Vec3f(pt3[0], pt3[1], pt3[2]), d, x, y) ? 1 : 0;
}
float ref_observation_weight(int32_t w, int32_t h, const float* depth, const float* n3, int32_t x, int32_t y, const float* v3) {
    Camera cam; SDFColorization c(cam); const cv::Mat d = cv::Mat::wrap(h, w, CV_32FC1, depth);
    return c.computeWeight(d, Vec3f(n3[0], n3[1], n3[2]), x, y, Vec3f(v3[0], v3[1], v3[2]));
}
void ref_compute_color(int32_t n, const uint8_t* rgb, const float* weights, float* out3) {
    std::vector<VertexObservation> obs((size_t)n);
    for (int i = 0; i < n; ++i) { obs[i].color = Vec3b(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]); obs[i].weight = weights[i]; obs[i].frame = i; }
    Camera cam; SDFColorization c(cam); const Vec3f col = c.computeColor(obs); out3[0] = col[0]; out3[1] = col[1]; out3[2] = col[2];
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
    for (size_t i = 0; i < m->vertices.size(); ++i) for (int k = 0; k < 3; ++k) { verts[3 * i + k] = m->vertices[i][k]; if (colors && i < m->colors.size()) colors[3 * i + k] = m->colors[i][k]; }
    for (size_t i = 0; i < m->face_vertices.size(); ++i) for (int k = 0; k < 3; ++k) faces[3 * i + k] = m->face_vertices[i][k];
}
// MeshUtil::removeLooseComponents (mesh/util.cpp:47-171) on a mesh built from caller arrays (colors may be NULL) / on an extracted mesh, in place
void* ref_mesh_from_arrays(int64_t nv, const float* verts, const uint8_t* colors, int64_t nf, const int32_t* faces) {
    Mesh* m = new Mesh;
    for (int64_t i = 0; i < nv; ++i) { m->vertices.push_back(Vec3f(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2])); if (colors) m->colors.push_back(Vec3b(colors[3 * i], colors[3 * i + 1], colors[3 * i + 2])); }
    for (int64_t i = 0; i < nf; ++i) m->face_vertices.push_back(Vec3i(faces[3 * i], faces[3 * i + 1], faces[3 * i + 2]));
    return m;
}
void ref_mesh_remove_loose(void* mesh) { if (mesh) MeshUtil::removeLooseComponents((Mesh*)mesh); }
int32_t ref_mesh_has_colors(void* mesh) { return (mesh && !((Mesh*)mesh)->colors.empty()) ? 1 : 0; }
/* SDFVisualization::applyColorAlbedo's colour of a voxel: scalarToColor(albedo, 255.0) (color_util.cpp:70-80) */
void ref_albedo_colors(int64_t n, const double* albedo, uint8_t* rgb) { for (int64_t i = 0; i < n; ++i) { const Vec3b c = scalarToColor(albedo[i], 255.0); rgb[3 * i] = c[0]; rgb[3 * i + 1] = c[1]; rgb[3 * i + 2] = c[2]; } }
/* SDFVisualization::applyColor* (visualization.cpp:228-416) of the reference on a grid built from caller arrays (voxels inserted in the given order): the colour every
 * voxel carries after the mode's method ran.  mode: "normals" | "lap" | "lum" | "lum_grad" | "albedo" | "shading_sv" | "shading_sv_const" | "chroma".  The shading
 * modes run over the reference's own Subvolumes of the grid (compute()), whose coefficients are looked up by subvolume index in (sub_index, sub_sh); a subvolume
 * that is not listed gets zeros.  Returns the number of subvolumes (0: not a shading mode), -1 for an unknown mode. */
int32_t ref_visualization_colors(const char* mode, float voxel_size, int64_t n, const int32_t* keys, const double* sdf_refined, const double* albedo, const float* weight, const uint8_t* color,
                                 float subvolume_size, int32_t n_sub, const int32_t* sub_index, const double* sub_sh, uint8_t* out, int64_t* visit_rank /* or NULL: position of voxel i in the grid's walk */) {
    SparseVoxelGrid<VoxelSBR>* g = SparseVoxelGrid<VoxelSBR>::create(voxel_size);
    for (int64_t i = 0; i < n; ++i) { VoxelSBR v; v.sdf = sdf_refined[i]; v.sdf_refined = sdf_refined[i]; v.albedo = albedo[i]; v.weight = weight[i];
        v.color = Vec3b(color[3 * i], color[3 * i + 1], color[3 * i + 2]); g->setVoxel(Vec3i(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]), v); }
    if (visit_rank) {
        std::unordered_map<Vec3i, int64_t, std::hash<Vec3i>> pos; int64_t r = 0;
        for (auto it = g->begin(); it != g->end(); ++it, ++r) pos[it->first] = r;
        for (int64_t i = 0; i < n; ++i) visit_rank[i] = pos[Vec3i(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2])];
    }
    SDFVisualization vis(g, "unused");
    const std::string m = mode; int32_t rc = 0;
    if (m == "normals") vis.applyColorNormals();
    else if (m == "lap") vis.applyColorLaplacian();
    else if (m == "lum") vis.applyColorIntensity();
    else if (m == "lum_grad") vis.applyColorIntensityGradient();
    else if (m == "albedo") vis.applyColorAlbedo();
    else if (m == "chroma") vis.applyColorChromacity();
    else if (m == "shading_sv" || m == "shading_sv_const") {
        Subvolumes sv(subvolume_size); sv.compute(g);
        std::vector<Eigen::VectorXd> coeffs((size_t)sv.count());
        for (int i = 0; i < (int)sv.count(); ++i) {
            Eigen::VectorXd c = Eigen::VectorXd::Zero(9); const Vec3i idx = sv.index(i);
            for (int k = 0; k < n_sub; ++k) if (sub_index[3 * k] == idx[0] && sub_index[3 * k + 1] == idx[1] && sub_index[3 * k + 2] == idx[2]) { for (int j = 0; j < 9; ++j) c[j] = sub_sh[9 * k + j]; break; }
            coeffs[(size_t)i] = c;
        }
        vis.applyColorShading(&sv, coeffs, m == "shading_sv_const"); rc = (int32_t)sv.count();
    } else rc = -1;
    for (int64_t i = 0; i < n; ++i) { const Vec3b c = g->voxel(Vec3i(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2])).color; out[3 * i] = c[0]; out[3 * i + 1] = c[1]; out[3 * i + 2] = c[2]; }
    delete g;
    return rc;
}
int32_t ref_mesh_save(void* mesh, const char* path) { Mesh* m = (Mesh*)mesh; return (m && m->save(path)) ? 1 : 0; }
void ref_mesh_free(void* mesh) { delete (Mesh*)mesh; }
// the two tables, for the case-by-case check of the product's packed copy
void ref_mc_tables(int32_t* edge256, int32_t* tri256x16) {
    for (int i = 0; i < 256; ++i) { edge256[i] = MarchingCubes<VoxelSBR>::edge_table_[i]; for (int k = 0; k < 16; ++k) tri256x16[16 * i + k] = MarchingCubes<VoxelSBR>::triangle_table_[i][k]; }
}


// ================================================================================================================================
// Pipeline level (round 3): the reference's OWN Optimizer / NLSSolver / SDFColorization / SDFAlgorithms / Subvolumes / LightingSVSH /
// SparseVoxelGrid::integrate / Intrinsic3D::refine, driven through the same C ABI shapes as oracle/i3d_oracle.h (orc_* -> ref_*) so that
// one Python harness runs both.  ceres::Solve underneath is oracle/ref_shim/mini_ceres_solver.hpp.
// ================================================================================================================================
typedef struct {
    int32_t iterations, lm_steps;
    double lambda_g, lambda_r0, lambda_r1, lambda_s0, lambda_s1, lambda_a;
    int32_t fix_poses, fix_intrinsics, fix_distortion;
    float occlusion_distance; int32_t num_observations;
    double thres_shell; int32_t grid_level, rgbd_level;
    int32_t cg_fixed_iterations; int32_t verbose; int32_t fix_sdf; int32_t carry_trust_radius;
} ref_opt_config;                                    /* = orc_opt_config */
typedef struct {
    int32_t rows[4]; double weight_sum[4]; double type_weight[4];
    int32_t valid_voxels, num_params, num_rows_reduced;
    double cost_initial, cost_final; int32_t lm_iterations, successful;
    int32_t cg_iters[50]; int32_t accepted[50]; int32_t n_attempts; double final_radius; int32_t termination;
} ref_iter_stats;                                    /* = orc_iter_stats */
typedef struct { int32_t data_rows, reg_rows, subvolumes, lm_iterations, termination; double cost_initial, cost_final; } ref_sh_stats;

}  // extern "C" (helpers below are C++)

namespace {

struct CoutSilencer { std::streambuf* o; std::streambuf* e; CoutSilencer() : o(std::cout.rdbuf(nullptr)), e(std::cerr.rdbuf(nullptr)) {} ~CoutSilencer() { std::cout.rdbuf(o); std::cerr.rdbuf(e); } };

struct RefFrames {                                   // K keyframe pyramids; images are borrowed from the caller
    int K = 0, levels = 0; std::vector<Pyramid> pyr;
};

Optimizer::Config to_opt_cfg(const ref_opt_config* c) {
    Optimizer::Config o; o.iterations = c->iterations; o.lm_steps = c->lm_steps; o.lambda_g = c->lambda_g; o.lambda_r0 = c->lambda_r0; o.lambda_r1 = c->lambda_r1;
    o.lambda_s0 = c->lambda_s0; o.lambda_s1 = c->lambda_s1; o.lambda_a = c->lambda_a;
    o.fix_poses = c->fix_poses != 0; o.fix_intrinsics = c->fix_intrinsics != 0; o.fix_distortion = c->fix_distortion != 0; return o;
}

// global parameter ids as the oracle numbers them: sdf_refined of voxel i (visit order) = i, albedo = N + i, pose f = 2N + 6f.., intrinsics, distortion
struct ParamIndex {
    std::unordered_map<const double*, long> id; long N = 0, K = 0;
    std::vector<Vec3i> keys;
    void bind(SparseVoxelGrid<VoxelSBR>* g, Optimizer::ImageFormationModel& im) {
        id.clear(); keys.clear(); N = (long)g->numVoxels(); K = (long)im.poses.size(); id.reserve((size_t)(2 * N + K + 2) * 2);
        long i = 0; for (auto it = g->begin(); it != g->end(); ++it, ++i) { id[&it->second.sdf_refined] = i; id[&it->second.albedo] = N + i; keys.push_back(it->first); }
        for (long f = 0; f < K; ++f) id[im.poses[f].data()] = 2 * N + 6 * f;
        id[im.intrinsics.data()] = 2 * N + 6 * K; id[im.distortion_coeffs.data()] = 2 * N + 6 * K + 4;
    }
};

struct SnapRow { int type, v, f, dir; double weight, residual; double J[29]; };
struct Snapshot { std::vector<SnapRow> rows[4]; std::vector<uint8_t> fix_sdf, fix_alb; long N = 0; };   // flags: 0 free, 1 constant, 2 not in the problem

int row_type(const ceres::CostFunction* c) {
    if (dynamic_cast<const ceres::DynamicAutoDiffCostFunction<ShadingCost, 4>*>(c)) return 0;
    if (dynamic_cast<const ceres::AutoDiffCostFunction<VolumetricRegularizer, 1, 1, 1, 1, 1, 1, 1, 1>*>(c)) return 1;
    if (dynamic_cast<const ceres::AutoDiffCostFunction<SurfaceStabRegularizer, 1, 1>*>(c)) return 2;
    if (dynamic_cast<const ceres::AutoDiffCostFunction<AlbedoRegularizer, 1, 1, 1>*>(c)) return 3;
    return -1;
}

void take_snapshot(const ceres::Problem& P, const ParamIndex& ix, Snapshot* s) {
    s->N = ix.N; s->fix_sdf.assign((size_t)ix.N, 2); s->fix_alb.assign((size_t)ix.N, 2);
    for (double* p : P.parameter_order()) { const long g = ix.id.at(p); if (g < ix.N) s->fix_sdf[g] = P.info(p).constant ? 1 : 0; else if (g < 2 * ix.N) s->fix_alb[g - ix.N] = P.info(p).constant ? 1 : 0; }
    for (const ceres::ResidualBlock& rb : P.blocks()) {
        SnapRow r; std::memset(&r, 0, sizeof r); r.type = row_type(rb.cost); r.f = -1; r.dir = -1;
        if (r.type < 0) continue;
        const long g0 = ix.id.at(rb.params[0]);
        r.v = (int)(r.type == 3 ? g0 - ix.N : g0);
        r.weight = static_cast<const ceres::ScaledLoss*>(rb.loss)->scale();
        std::vector<std::vector<double>> jac(rb.params.size()); std::vector<double*> jp(rb.params.size());
        const std::vector<int32_t>& sz = rb.cost->parameter_block_sizes();
        for (size_t i = 0; i < jac.size(); ++i) { jac[i].assign((size_t)sz[i], 0.0); jp[i] = jac[i].data(); }
        rb.cost->Evaluate(rb.params.data(), &r.residual, r.type == 0 ? jp.data() : nullptr);
        if (r.type == 0) {
            r.f = (int)((ix.id.at(rb.params[14]) - 2 * ix.N) / 6);
            int k = 0; for (size_t i = 0; i < jac.size(); ++i) for (int c = 0; c < sz[i]; ++c) r.J[k++] = jac[i][c];
        } else if (r.type == 3) {
            const Vec3i a = ix.keys[(size_t)r.v], b = ix.keys[(size_t)(ix.id.at(rb.params[1]) - ix.N)];
            const int d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            r.dir = d[0] == 1 ? 0 : d[0] == -1 ? 1 : d[1] == 1 ? 2 : d[1] == -1 ? 3 : d[2] == 1 ? 4 : 5;
        }
        s->rows[r.type].push_back(r);
    }
}

struct RefRun {                                      // everything one Optimizer::optimize call of the reference needs
    Optimizer::Data data; Optimizer::ImageFormationModel im; Camera cam; SDFColorization col;
    RefRun() : col(cam) {}
    void setup(SparseVoxelGrid<VoxelSBR>* g, RefFrames* fr, const ref_opt_config* c, const double* intr, const double* dist, const double* poses, const double* voxel_sh) {
        data.grid = g; data.thres_shell = c->thres_shell; data.grid_level = c->grid_level; data.rgbd_level = c->rgbd_level;
        const size_t N = g->numVoxels();
        data.voxel_sh_coeffs.assign(N, Eigen::VectorXd(9));
        for (size_t i = 0; i < N; ++i) for (int j = 0; j < 9; ++j) data.voxel_sh_coeffs[i][j] = voxel_sh[9 * i + j];
        for (int i = 0; i < 4; ++i) im.intrinsics[i] = intr[i];
        for (int i = 0; i < 5; ++i) im.distortion_coeffs[i] = dist[i];
        im.poses.resize((size_t)fr->K); im.rgbd_pyr = fr->pyr; im.frame_ids.clear();
        for (int f = 0; f < fr->K; ++f) { for (int j = 0; j < 6; ++j) im.poses[f][j] = poses[6 * f + j]; im.frame_ids.push_back(f); }
        data.shading_cost_data.clear();                                                     // Intrinsic3D::prepareRgbdLevel (intrinsic3d.cpp:337-350)
        for (size_t i = 0; i < im.rgbd_pyr.size(); ++i) {
            cv::Mat lum = im.rgbd_pyr[i].intensity(data.rgbd_level);
            data.shading_cost_data.push_back(ShadingCostData(data.rgbd_level, static_cast<double>(g->voxelSize()), lum.cols, lum.rows, reinterpret_cast<const float*>(lum.data)));
        }
        SDFColorization::Config cc; cc.max_occlusion_distance = c->occlusion_distance; cc.max_num_observations = (size_t)c->num_observations;
        col.setConfig(cc);
    }
};

std::unordered_map<void*, SparseVoxelGrid<Voxel>*>& twins() { static std::unordered_map<void*, SparseVoxelGrid<Voxel>*> t; return t; }

}  // namespace

extern "C" {

void* ref_grid_from_voxels(float voxel_size, int64_t n, const int32_t* keys, const float* sdf, const float* weight, const uint8_t* color) {
    SparseVoxelGrid<Voxel>* g = SparseVoxelGrid<Voxel>::create(voxel_size);        // records inserted in file order (SparseVoxelGrid::load), then convert()
    for (int64_t i = 0; i < n; ++i) { Voxel v; v.sdf = sdf[i]; v.weight = weight[i]; v.color = Vec3b(color[3 * i], color[3 * i + 1], color[3 * i + 2]);
        g->setVoxel(Vec3i(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]), v); }
    SparseVoxelGrid<VoxelSBR>* out = SDFAlgorithms::convert(g); twins()[out] = g; return out;     // the Voxel grid is kept: Intrinsic3D::refine starts from it
}
int64_t ref_grid_size(void* g) { return (int64_t)((SparseVoxelGrid<VoxelSBR>*)g)->numVoxels(); }
float ref_grid_voxel_size(void* g) { return ((SparseVoxelGrid<VoxelSBR>*)g)->voxelSize(); }
void ref_grid_export(void* gp, int32_t* keys, double* sdf, double* sdf_refined, double* albedo, float* weight, uint8_t* color) {
    auto* g = (SparseVoxelGrid<VoxelSBR>*)gp; size_t i = 0;
    for (auto it = g->begin(); it != g->end(); ++it, ++i) {
        if (keys) for (int c = 0; c < 3; ++c) keys[3 * i + c] = it->first[c];
        if (sdf) sdf[i] = it->second.sdf;
        if (sdf_refined) sdf_refined[i] = it->second.sdf_refined;
        if (albedo) albedo[i] = it->second.albedo;
        if (weight) weight[i] = it->second.weight;
        if (color) for (int c = 0; c < 3; ++c) color[3 * i + c] = it->second.color[c];
    }
}
void ref_grid_import(void* gp, const double* sdf_refined, const double* albedo, const uint8_t* color) {
    auto* g = (SparseVoxelGrid<VoxelSBR>*)gp; size_t i = 0;
    for (auto it = g->begin(); it != g->end(); ++it, ++i) {
        if (sdf_refined) it->second.sdf_refined = sdf_refined[i];
        if (albedo) it->second.albedo = albedo[i];
        if (color) it->second.color = Vec3b(color[3 * i], color[3 * i + 1], color[3 * i + 2]);
    }
}
void ref_grid_clear_outside_shell(void* g, double thres) { SDFAlgorithms::clearVoxelsOutsideThinShell((SparseVoxelGrid<VoxelSBR>*)g, thres); }
void* ref_grid_upsample(void* g) { return SDFAlgorithms::upsample<VoxelSBR>((SparseVoxelGrid<VoxelSBR>*)g); }
void ref_grid_free(void* g) { auto it = twins().find(g); if (it != twins().end()) { delete it->second; twins().erase(it); } delete (SparseVoxelGrid<VoxelSBR>*)g; }

void* ref_frames_create(int32_t K, int32_t levels) {
    auto* f = new RefFrames; f->K = K; f->levels = levels; f->pyr.resize((size_t)K);
    for (auto& p : f->pyr) { p.color_pyramid_.resize((size_t)levels); p.intensity_pyramid_.resize((size_t)levels); p.depth_pyramid_.resize((size_t)levels); }
    return f;
}
void ref_frames_set(void* fr, int32_t f, int32_t lvl, int32_t w, int32_t h, const float* lum, const float* depth, const uint8_t* bgr) {
    Pyramid& p = ((RefFrames*)fr)->pyr[(size_t)f];
    p.intensity_pyramid_[(size_t)lvl] = cv::Mat::wrap(h, w, CV_32FC1, lum); p.depth_pyramid_[(size_t)lvl] = cv::Mat::wrap(h, w, CV_32FC1, depth);
    p.color_pyramid_[(size_t)lvl] = bgr ? cv::Mat::wrap(h, w, CV_8UC3, bgr) : cv::Mat(h, w, CV_8UC3);
}
void ref_frames_free(void* fr) { delete (RefFrames*)fr; }

// Optimizer::optimize of the reference (optimizer.cpp:109-173), every outer iteration: addVoxelResiduals, NLSSolver::buildProblem,
// fixVoxelParams, NLSSolver::solve -> mini-ceres.  Statistics come from the ceres::Problem / Summary of each iteration.
int32_t ref_optimize(void* g, void* fr, const ref_opt_config* c, double* intr, double* dist, double* poses, const double* voxel_sh, ref_iter_stats* stats) {
    CoutSilencer quiet;
    auto* G = (SparseVoxelGrid<VoxelSBR>*)g; auto* F = (RefFrames*)fr;
    RefRun run; run.setup(G, F, c, intr, dist, poses, voxel_sh);
    ParamIndex ix; ix.bind(G, run.im);
    int call = 0;
    ceres::hooks() = ceres::SolveHooks(); ceres::hooks().cg_fixed_iterations = c->cg_fixed_iterations;
    ceres::hooks().on_problem = [&](const ceres::Problem& P) {
        if (!stats || call >= c->iterations) { ++call; return; }
        ref_iter_stats& st = stats[call++]; std::memset(&st, 0, sizeof st);
        Snapshot s; take_snapshot(P, ix, &s);
        for (int t = 0; t < 4; ++t) st.rows[t] = (int32_t)s.rows[t].size();
    };
    struct After : ceres::IterationCallback { ceres::CallbackReturnType operator()(const ceres::IterationSummary&) override { return ceres::SOLVER_CONTINUE; } };
    ceres::hooks_summary() = [&](const ceres::Solver::Summary& sum) {
        if (!stats || call < 1 || call > c->iterations) return;
        ref_iter_stats& st = stats[call - 1];
        st.cost_initial = sum.initial_cost - sum.fixed_cost; st.cost_final = sum.final_cost - sum.fixed_cost; st.num_params = sum.num_parameters_reduced; st.num_rows_reduced = sum.num_residuals_reduced;
        st.lm_iterations = (int32_t)sum.iterations.size() - 1; st.successful = sum.num_successful_steps; st.n_attempts = 0;
        for (size_t i = 1; i < sum.iterations.size() && st.n_attempts < 50; ++i) { st.cg_iters[st.n_attempts] = sum.iterations[i].linear_solver_iterations; st.accepted[st.n_attempts] = sum.iterations[i].step_is_successful ? 1 : 0; ++st.n_attempts; }
        st.final_radius = sum.iterations.empty() ? 0.0 : sum.iterations.back().trust_region_radius;
        st.termination = sum.termination_type == ceres::USER_SUCCESS ? 2 : sum.termination_type == ceres::CONVERGENCE ? 1 : sum.termination_type == ceres::NO_CONVERGENCE ? 0 : 3;
    };
    Optimizer opt(to_opt_cfg(c));
    const bool ok = opt.optimize(run.col, run.data, run.im);
    ceres::hooks() = ceres::SolveHooks(); ceres::hooks_summary() = nullptr;
    for (int i = 0; i < 4; ++i) intr[i] = run.im.intrinsics[i];
    for (int i = 0; i < 5; ++i) dist[i] = run.im.distortion_coeffs[i];
    for (int f = 0; f < F->K; ++f) for (int j = 0; j < 6; ++j) poses[6 * f + j] = run.im.poses[f][j];
    return ok ? 0 : 1;
}

// one residual collection (no solve) exactly as outer iteration `iteration` of `cfg->iterations` would assemble it
void* ref_collect(void* g, void* fr, const ref_opt_config* c, const double* intr, const double* dist, const double* poses, const double* voxel_sh, int32_t iteration) {
    CoutSilencer quiet;
    auto* G = (SparseVoxelGrid<VoxelSBR>*)g; auto* F = (RefFrames*)fr;
    RefRun run; run.setup(G, F, c, intr, dist, poses, voxel_sh);
    ParamIndex ix; ix.bind(G, run.im);
    auto* snap = new Snapshot; int call = 0;
    ceres::hooks() = ceres::SolveHooks(); ceres::hooks().skip_minimize = true;
    ceres::hooks().on_problem = [&](const ceres::Problem& P) { if (call++ == iteration) take_snapshot(P, ix, snap); };
    Optimizer opt(to_opt_cfg(c)); opt.optimize(run.col, run.data, run.im);      // nothing moves (skip_minimize): iteration k sees the input state
    ceres::hooks() = ceres::SolveHooks();
    snap->N = ix.N;
    return snap;
}
void ref_problem_counts(void* p, int32_t rows[4], double ws[4], double tw[4]) {
    auto* s = (Snapshot*)p; for (int t = 0; t < 4; ++t) { rows[t] = (int32_t)s->rows[t].size(); if (ws) ws[t] = 0.0; if (tw) tw[t] = 0.0; }
}
void ref_problem_flags(void* p, uint8_t* active, uint8_t* ring_ok, uint8_t* fix_sdf, uint8_t* fix_alb) {
    auto* s = (Snapshot*)p; (void)active; (void)ring_ok;
    for (long i = 0; i < s->N; ++i) { if (fix_sdf) fix_sdf[i] = s->fix_sdf[(size_t)i]; if (fix_alb) fix_alb[i] = s->fix_alb[(size_t)i]; }
}
void ref_problem_eg(void* p, int32_t* v, int32_t* f, double* weight, double* residual, double* J) {
    auto* s = (Snapshot*)p; const auto& rows = s->rows[0];
    for (size_t i = 0; i < rows.size(); ++i) { v[i] = rows[i].v; f[i] = rows[i].f; weight[i] = rows[i].weight; residual[i] = rows[i].residual; if (J) for (int k = 0; k < 29; ++k) J[i * 29 + k] = rows[i].J[k]; }
}
void ref_problem_reg(void* p, int32_t type, int32_t* v, int32_t* dir, double* weight, double* residual) {
    auto* s = (Snapshot*)p; const auto& rows = s->rows[type];
    for (size_t i = 0; i < rows.size(); ++i) { v[i] = rows[i].v; if (dir) dir[i] = rows[i].dir; weight[i] = rows[i].weight; if (residual) residual[i] = rows[i].residual; }
}
void ref_problem_free(void* p) { delete (Snapshot*)p; }

// LightingSVSH::estimate + computeVoxelShCoeffs of the reference (lighting_svsh.cpp:83-110,166-346; subvolumes.cpp)
int32_t ref_estimate_sh(void* g, float subvolume_size, double lambda_reg, double thres_shell, int32_t cg_fixed, int32_t* num_subvolumes, double* sh, int32_t* sub_index, int32_t cap,
                        double* voxel_sh, uint8_t* voxel_has, ref_sh_stats* st) {
    CoutSilencer quiet;
    auto* G = (SparseVoxelGrid<VoxelSBR>*)g;
    LightingSVSH L(G, subvolume_size, lambda_reg, thres_shell, true);
    ceres::hooks() = ceres::SolveHooks(); ceres::hooks().cg_fixed_iterations = cg_fixed;
    if (st) std::memset(st, 0, sizeof *st);
    ceres::hooks().on_problem = [&](const ceres::Problem& P) { if (!st) return; for (const auto& rb : P.blocks()) { if (rb.cost->num_residuals() == 1) ++st->data_rows; else ++st->reg_rows; } };
    ceres::hooks_summary() = [&](const ceres::Solver::Summary& sum) { if (!st) return; st->lm_iterations = (int32_t)sum.iterations.size() - 1; st->cost_initial = sum.initial_cost; st->cost_final = sum.final_cost;
        st->termination = sum.termination_type == ceres::CONVERGENCE ? 1 : sum.termination_type == ceres::NO_CONVERGENCE ? 0 : 3; };
    const bool ok = L.estimate();
    ceres::hooks() = ceres::SolveHooks(); ceres::hooks_summary() = nullptr;
    if (!ok) return 1;
    const int S = (int)L.subvolumes().count(); *num_subvolumes = S; if (st) st->subvolumes = S;
    if (S > cap) return 2;
    const std::vector<Eigen::VectorXd> coeffs = L.shCoeffs();
    for (int i = 0; i < S; ++i) { for (int j = 0; j < 9; ++j) sh[9 * i + j] = coeffs[(size_t)i][j];
        if (sub_index) { const Vec3i ix = L.subvolumes().index(i); for (int c = 0; c < 3; ++c) sub_index[3 * i + c] = ix[c]; } }
    if (voxel_sh) {
        std::vector<Eigen::VectorXd> vc; L.computeVoxelShCoeffs(vc);
        for (size_t i = 0; i < vc.size(); ++i) { const bool has = vc[i].size() == 9; if (voxel_has) voxel_has[i] = has ? 1 : 0; for (int j = 0; j < 9; ++j) voxel_sh[9 * i + j] = has ? vc[i][j] : 0.0; }
    }
    return 0;
}

// Intrinsic3D::recomputeColors (intrinsic3d.cpp:381-409) on SDFColorization::add / compute of the reference
int32_t ref_recompute_colors(void* g, void* fr, const double* intr, const double* dist, const double* poses, float occlusion_distance, int32_t num_observations) {
    CoutSilencer quiet;
    auto* G = (SparseVoxelGrid<VoxelSBR>*)g; auto* F = (RefFrames*)fr;
    HolderSensor sensor; KeyframeSelection ks;
    Intrinsic3D::Config cfg; cfg.occlusions_distance = occlusion_distance; cfg.num_observations = (size_t)num_observations; cfg.num_rgbd_levels = F->levels;
    Intrinsic3D i3d(cfg, Optimizer::Config(), &sensor, &ks);
    i3d.opt_data_.grid = G; i3d.image_model_.rgbd_pyr = F->pyr; i3d.image_model_.poses.resize((size_t)F->K);
    for (int i = 0; i < 4; ++i) i3d.image_model_.intrinsics[i] = intr[i];
    for (int i = 0; i < 5; ++i) i3d.image_model_.distortion_coeffs[i] = dist[i];
    for (int f = 0; f < F->K; ++f) for (int j = 0; j < 6; ++j) i3d.image_model_.poses[(size_t)f][j] = poses[6 * f + j];
    SDFColorization::Config cc; cc.max_occlusion_distance = occlusion_distance; cc.max_num_observations = (size_t)num_observations; i3d.sdf_colorization_.setConfig(cc);
    return i3d.recomputeColors() ? 0 : 1;
}

// Intrinsic3D::refine (intrinsic3d.cpp:206-290) of the reference on a grid built by ref_grid_from_voxels' Voxel twin; *grid_io receives the final grid
int32_t ref_refine(void** grid_io, void* fr, const ref_opt_config* c, int32_t num_grid_levels, int32_t num_rgbd_levels, double thres_shell_factor, double thres_shell_factor_final,
                   int32_t clear_distant_voxels, float subvolume_size_sh, double sh_lambda_reg, double* intr, double* dist, double* poses, int32_t* levels_done) {
    CoutSilencer quiet;
    auto* G = (SparseVoxelGrid<VoxelSBR>*)*grid_io; auto* F = (RefFrames*)fr;
    auto tw = twins().find(G); if (tw == twins().end()) return 3;          // refine() takes the Voxel grid the VoxelSBR grid was converted from
    SparseVoxelGrid<Voxel>* twin = tw->second;
    HolderSensor sensor; KeyframeSelection ks;
    Intrinsic3D::Config cfg; cfg.num_grid_levels = num_grid_levels; cfg.num_rgbd_levels = num_rgbd_levels; cfg.thres_shell_factor = thres_shell_factor; cfg.thres_shell_factor_final = thres_shell_factor_final;
    cfg.clear_distant_voxels = clear_distant_voxels != 0; cfg.occlusions_distance = c->occlusion_distance; cfg.num_observations = (size_t)c->num_observations;
    cfg.subvolume_size_sh = subvolume_size_sh; cfg.sh_est_lambda_reg = sh_lambda_reg;
    { Vec4 k; for (int i = 0; i < 4; ++i) k[i] = intr[i]; sensor.cam_color_.setIntrinsics(k); cv::Mat l0 = F->pyr[0].intensity(0); sensor.cam_color_.setWidth(l0.cols); sensor.cam_color_.setHeight(l0.rows); }
    Intrinsic3D i3d(cfg, to_opt_cfg(c), &sensor, &ks);
    i3d.image_model_.rgbd_pyr = F->pyr; i3d.image_model_.poses.resize((size_t)F->K);
    for (int i = 0; i < 4; ++i) i3d.image_model_.intrinsics[i] = intr[i];
    for (int i = 0; i < 5; ++i) i3d.image_model_.distortion_coeffs[i] = dist[i];
    for (int f = 0; f < F->K; ++f) { for (int j = 0; j < 6; ++j) i3d.image_model_.poses[(size_t)f][j] = poses[6 * f + j]; i3d.image_model_.frame_ids.push_back(f); }
    struct Keep : Intrinsic3D::RefinementCallback { SparseVoxelGrid<VoxelSBR>* last = nullptr; int n = 0; std::vector<std::tuple<Vec3i, VoxelSBR>> recs; float vs = 0, dmin = 0, dmax = 0;
        void onSDFRefined(const Intrinsic3D::RefinementInfo& info) override { ++n; recs.clear(); vs = info.grid->voxelSize(); dmin = info.grid->depthMin(); dmax = info.grid->depthMax();
            for (auto it = info.grid->begin(); it != info.grid->end(); ++it) recs.emplace_back(it->first, it->second); } } keep;
    i3d.addRefinementCallback(&keep);
    ceres::hooks() = ceres::SolveHooks(); ceres::hooks().cg_fixed_iterations = c->cg_fixed_iterations;
    const bool ok = i3d.refine(twin);                // deletes its own VoxelSBR grid at the end (:287); the last callback state is what we return
    ceres::hooks() = ceres::SolveHooks();
    twins().erase(G); delete twin; delete G;
    // the records of the last callback go back through a fresh container filled in that visit order; the caller compares BY KEY
    SparseVoxelGrid<VoxelSBR>* out = SparseVoxelGrid<VoxelSBR>::create(keep.vs > 0 ? keep.vs : 0.004f, keep.dmin, keep.dmax);
    for (auto& r : keep.recs) out->setVoxel(std::get<0>(r), std::get<1>(r));
    *grid_io = out;
    for (int i = 0; i < 4; ++i) intr[i] = i3d.image_model_.intrinsics[i];
    for (int i = 0; i < 5; ++i) dist[i] = i3d.image_model_.distortion_coeffs[i];
    for (int f = 0; f < F->K; ++f) for (int j = 0; j < 6; ++j) poses[6 * f + j] = i3d.image_model_.poses[(size_t)f][j];
    if (levels_done) *levels_done = keep.n;
    return ok ? 0 : 1;
}

// TSDF fusion: AppFusion::fuseSDF's per-frame body (app_fusion.cpp:150-170) on SparseVoxelGrid<Voxel>::integrate / alloc of the reference
struct RefFusion { SparseVoxelGrid<Voxel>* grid; };
void* ref_fusion_create(float voxel_size, float depth_min, float depth_max, const float* clip6) {
    auto* f = new RefFusion; f->grid = SparseVoxelGrid<Voxel>::create(voxel_size, depth_min, depth_max);
    if (clip6) { Vec6f cb; for (int i = 0; i < 6; ++i) cb[i] = clip6[i]; if (cb.norm() > 0.0f) f->grid->setClipBounds(cb); }
    return f;
}
void ref_fusion_integrate(void* fp, int32_t dw, int32_t dh, const float* dc, int32_t cw, int32_t ch, const float* cc, const float* depth_in, const uint8_t* bgr, const float* pose16, int32_t erode_window) {
    auto* f = (RefFusion*)fp;
    auto mk = [](const float* k, int w, int h) { Mat3f K = Mat3f::Identity(); K(0, 0) = k[0]; K(1, 1) = k[1]; K(0, 2) = k[2]; K(1, 2) = k[3]; return Camera(K, w, h); };
    Camera dcam = mk(dc, dw, dh), ccam = mk(cc, cw, ch);
    cv::Mat depth = cv::Mat::wrap(dh, dw, CV_32FC1, depth_in); const cv::Mat color = cv::Mat::wrap(ch, cw, CV_8UC3, bgr);
    if (erode_window > 0) depth = erodeDiscontinuities(depth, erode_window);
    cv::Mat normals = computeNormals(dcam.intrinsics(), depth);
    Mat4f pose; for (int r = 0; r < 4; ++r) for (int c2 = 0; c2 < 4; ++c2) pose(r, c2) = pose16[4 * r + c2];
    f->grid->integrate(dcam, ccam, depth, color, normals, pose);
}
void ref_fusion_finish(void* fp, int32_t iters) { auto* f = (RefFusion*)fp; SDFAlgorithms::correctSDF(f->grid, (unsigned)iters); SDFAlgorithms::clearInvalidVoxels(f->grid); }
int64_t ref_fusion_size(void* fp) { return (int64_t)((RefFusion*)fp)->grid->numVoxels(); }
void ref_fusion_export(void* fp, int32_t* keys, float* sdf, float* weight, uint8_t* color) {
    auto* g = ((RefFusion*)fp)->grid; size_t i = 0;
    for (auto it = g->begin(); it != g->end(); ++it, ++i) { for (int c = 0; c < 3; ++c) { keys[3 * i + c] = it->first[c]; color[3 * i + c] = it->second.color[c]; } sdf[i] = it->second.sdf; weight[i] = it->second.weight; }
}
/* on-disk formats, the reference's own writers / readers (sparse_voxel_grid.cpp:484-569, camera.cpp:202-274) */
int32_t ref_fusion_save(void* fp, const char* path) { return ((RefFusion*)fp)->grid->save(path) ? 1 : 0; }
void* ref_fusion_load(const char* path) {
    auto* f = new RefFusion; f->grid = SparseVoxelGrid<Voxel>::create(0.004f, 0.1f, 5.0f);
    if (!f->grid->load(path)) { delete f->grid; delete f; return nullptr; }
    return f;
}
void ref_fusion_header(void* fp, float* voxel_size, float* truncation, float* integration_weight_sample) {
    auto* g = ((RefFusion*)fp)->grid; *voxel_size = g->voxel_size_; *truncation = g->truncation_; *integration_weight_sample = g->integration_weight_sample_;
}
int32_t ref_grid_save(void* gp, const char* path) { return ((SparseVoxelGrid<VoxelSBR>*)gp)->save(path) ? 1 : 0; }
void* ref_grid_load(const char* path) {
    auto* g = SparseVoxelGrid<VoxelSBR>::create(0.004f, 0.1f, 5.0f);
    if (!g->load(path)) { delete g; return nullptr; }
    return g;
}
int32_t ref_camera_save(const char* path, int32_t w, int32_t h, const float* k4, const float* dist5) {
    Mat3f K = Mat3f::Identity(); K(0, 0) = k4[0]; K(1, 1) = k4[1]; K(0, 2) = k4[2]; K(1, 2) = k4[3];
    Camera cam(K, w, h); Vec5f d; for (int i = 0; i < 5; ++i) d[i] = dist5[i]; cam.setDistortion(d);
    return cam.save(path) ? 1 : 0;
}
int32_t ref_camera_load(const char* path, int32_t* w, int32_t* h, float* k4, float* dist5) {
    Camera cam; const bool ok = cam.load(path);
    *w = cam.width(); *h = cam.height(); const Mat3f K = cam.intrinsics(); k4[0] = K(0, 0); k4[1] = K(1, 1); k4[2] = K(0, 2); k4[3] = K(1, 2);
    const Vec5f d = cam.distortion(); for (int i = 0; i < 5; ++i) dist5[i] = d[i];
    return ok ? 1 : 0;
}
/* SensorI3d on a dataset folder (rgbd/sensor_i3d.cpp:48-345, rgbd/sensor.cpp:50-63,121-220) */
void ref_set_imdecode_hook(cv::imdecode_hook_t h) { cv::imdecode_hook() = h; }
void* ref_sensor_open(const char* folder, int32_t max_frames, float depth_min, float depth_max) {
    SensorI3d* s = new SensorI3d; s->setNumFramesMax(max_frames); s->setDepthMin(depth_min); s->setDepthMax(depth_max);
    std::streambuf* o1 = std::cout.rdbuf(nullptr); std::streambuf* o2 = std::cerr.rdbuf(nullptr);
    const bool ok = s->init(folder);
    std::cout.rdbuf(o1); std::cerr.rdbuf(o2);
    if (!ok) { delete s; return nullptr; }
    return s;
}
/* Sensor::create(Settings&) (rgbd/sensor.cpp:64-118) with the (key, value) strings of a sensor.yml; range2 = depthMin(), depthMax() of what it built */
void* ref_sensor_create(int32_t n, const char* const* keys, const char* const* values, float* range2, int32_t* max_frames) {
    CoutSilencer quiet;
    Settings cfg; for (int i = 0; i < n; ++i) cfg.set<std::string>(keys[i], values[i]);
    Sensor* s = Sensor::create(cfg);
    if (s) { range2[0] = s->depthMin(); range2[1] = s->depthMax(); *max_frames = s->numFramesMax(); }
    return s;
}
void ref_sensor_info(void* h, int32_t* num_frames, int32_t* num_stored, int32_t* cwh, int32_t* dwh, float* ci4, float* di4) {
    SensorI3d* s = (SensorI3d*)h; *num_frames = s->numFrames(); *num_stored = (int32_t)s->depth_images_.size();
    cwh[0] = s->colorCamera().width(); cwh[1] = s->colorCamera().height(); dwh[0] = s->depthCamera().width(); dwh[1] = s->depthCamera().height();
    const Mat3f Kc = s->colorCamera().intrinsics(), Kd = s->depthCamera().intrinsics();
    ci4[0] = Kc(0, 0); ci4[1] = Kc(1, 1); ci4[2] = Kc(0, 2); ci4[3] = Kc(1, 2); di4[0] = Kd(0, 0); di4[1] = Kd(1, 1); di4[2] = Kd(0, 2); di4[3] = Kd(1, 2);
}
void ref_sensor_pose(void* h, int32_t id, float* m16) { const Mat4f p = ((SensorI3d*)h)->pose(id); for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m16[4 * r + c] = p(r, c); }
double ref_sensor_time(void* h, int32_t id) { return ((SensorI3d*)h)->timeDepth(id); }
/* returns rows * cols (0 = empty image); out may be NULL */
int64_t ref_sensor_depth(void* h, int32_t id, float* out) { const cv::Mat d = ((SensorI3d*)h)->depth(id); if (d.empty()) return 0; if (out) std::memcpy(out, d.data, (size_t)d.rows * d.cols * 4); return (int64_t)d.rows * d.cols; }
int64_t ref_sensor_color(void* h, int32_t id, uint8_t* out) { const cv::Mat c = ((SensorI3d*)h)->color(id); if (c.empty()) return 0; if (out) std::memcpy(out, c.data, (size_t)c.rows * c.cols * 3); return (int64_t)c.rows * c.cols; }
void ref_sensor_set_pose(void* h, int32_t id, const float* m16) { Mat4f p; for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) p(r, c) = m16[4 * r + c]; ((SensorI3d*)h)->setPose(id, p); }
int32_t ref_sensor_save_poses(void* h, const char* path) { return ((SensorI3d*)h)->savePoses(path) ? 1 : 0; }
/* Sensor::loadPoses (static): TUM trajectory file -> timestamps + camera-to-world matrices; returns the count (-1: file not opened) */
int64_t ref_load_poses(const char* path, int32_t first_is_identity, int64_t cap, double* timestamps, float* m16) {
    std::vector<Mat4f> poses; std::vector<double> ts;
    if (!Sensor::loadPoses(path, poses, ts, first_is_identity != 0)) return -1;
    for (size_t i = 0; i < poses.size() && (int64_t)i < cap; ++i) { timestamps[i] = ts[i]; for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m16[16 * i + 4 * r + c] = poses[i](r, c); }
    return (int64_t)poses.size();
}
void ref_sensor_free(void* h) { delete (SensorI3d*)h; }

/* AppFusion::fuseSDF (apps/src/app_fusion.cpp:107-200) on a dataset folder: the reference's own fusion application loop — keyframe filter, erosion, normals,
 * integrate, correctSDF, clearInvalidVoxels, .tsdf and mesh output — over its own SensorI3d.  cfg: n (key, value) string pairs for nv::Settings. */
int32_t ref_app_fusion(const char* folder, int32_t max_frames, float depth_min, float depth_max, int32_t n, const char* const* keys, const char* const* values) {
    SensorI3d* s = new SensorI3d; s->setNumFramesMax(max_frames); s->setDepthMin(depth_min); s->setDepthMax(depth_max);
    const bool quiet = std::getenv("I3D_REF_VERBOSE") == nullptr;
    std::streambuf* o1 = std::cout.rdbuf(); std::streambuf* o2 = std::cerr.rdbuf();
    if (quiet) { std::cout.rdbuf(nullptr); std::cerr.rdbuf(nullptr); }
    bool ok = s->init(folder);
    if (ok) {
        Settings cfg; for (int i = 0; i < n; ++i) cfg.set<std::string>(keys[i], values[i]);
        AppFusion app; app.sensor_ = s;              // (the destructor deletes the sensor)
        ok = app.fuseSDF(cfg);
    } else delete s;
    std::cout.rdbuf(o1); std::cerr.rdbuf(o2);
    return ok ? 1 : 0;
}

/* AppIntrinsic3D::run (apps/src/app_intrinsic3d.cpp:72-155) without its command line and yml reading, on a dataset folder: its own SensorI3d, KeyframeSelection
 * ::load, SparseVoxelGrid<Voxel>::create(input_sdf), the two Config::load, Intrinsic3D (with its own init), refine, and onSDFRefined per level — the meshes of
 * the enabled colour modes, the pose and the intrinsics files.  cfg: (key, value) strings of intrinsic3d.yml. */
int32_t ref_app_intrinsic3d(const char* folder, int32_t max_frames, float depth_min, float depth_max, int32_t n, const char* const* keys, const char* const* values) {
    const bool quiet = std::getenv("I3D_REF_VERBOSE") == nullptr;
    std::streambuf* o1 = std::cout.rdbuf(); std::streambuf* o2 = std::cerr.rdbuf();
    if (quiet) { std::cout.rdbuf(nullptr); std::cerr.rdbuf(nullptr); }
    SensorI3d* s = new SensorI3d; s->setNumFramesMax(max_frames); s->setDepthMin(depth_min); s->setDepthMax(depth_max);
    bool ok = s->init(folder);
    if (!ok) delete s;
    else {
        AppIntrinsic3D app;                                              // (its destructor deletes sensor, keyframe selection and Intrinsic3D)
        for (int i = 0; i < n; ++i) app.i3d_cfg_.set<std::string>(keys[i], values[i]);
        app.sensor_ = s;
        app.keyframe_selection_ = new KeyframeSelection();
        (void)app.keyframe_selection_->load(app.i3d_cfg_.get<std::string>("keyframes"));
        SparseVoxelGrid<Voxel>* grid = SparseVoxelGrid<Voxel>::create(app.i3d_cfg_.get<std::string>("input_sdf"), s->depthMin(), s->depthMax());
        ok = grid != nullptr;
        if (ok) {
            Intrinsic3D::Config a; a.load(app.i3d_cfg_); Optimizer::Config b; b.load(app.i3d_cfg_);
            Intrinsic3DInit* i3d = new Intrinsic3DInit(a, b, s, app.keyframe_selection_);
            app.intrinsic3d_ = i3d; i3d->addRefinementCallback(&app);
            ok = i3d->refine(grid);
            delete grid;
        }
    }
    std::cout.rdbuf(o1); std::cerr.rdbuf(o2);
    return ok ? 1 : 0;
}

/* Intrinsic3D::init (intrinsic3d.cpp:151-203) on a dataset folder: SensorI3d + keyframe flags -> the image formation model the optimisation starts from.
 * Returns a handle (NULL: the sensor could not be initialised); the getters hand out keyframe ids, world-to-camera pose vectors, intrinsics and the
 * pyramid images.  init's own return value (the initial recolouring of a one-voxel stand-in grid) is not what is inspected here. */
struct RefInit { SensorI3d* sensor; KeyframeSelection* ks; Intrinsic3DInit* app; SparseVoxelGrid<VoxelSBR>* grid; };
void* ref_i3d_init(const char* folder, int32_t max_frames, float depth_min, float depth_max, int64_t n_flags, const uint8_t* is_kf, int32_t num_rgbd_levels) {
    const bool quiet = std::getenv("I3D_REF_VERBOSE") == nullptr;
    std::streambuf* o1 = std::cout.rdbuf(); std::streambuf* o2 = std::cerr.rdbuf();
    if (quiet) { std::cout.rdbuf(nullptr); std::cerr.rdbuf(nullptr); }
    SensorI3d* s = new SensorI3d; s->setNumFramesMax(max_frames); s->setDepthMin(depth_min); s->setDepthMax(depth_max);
    RefInit* R = nullptr;
    if (s->init(folder)) {
        KeyframeSelection* ks = new KeyframeSelection(1); ks->frame_scores_.assign((size_t)n_flags, 1.0); ks->is_keyframe_.resize((size_t)n_flags);
        for (int64_t i = 0; i < n_flags; ++i) ks->is_keyframe_[(size_t)i] = is_kf[i] != 0;
        Intrinsic3D::Config cfg; cfg.num_grid_levels = 1; cfg.num_rgbd_levels = num_rgbd_levels; cfg.thres_shell_factor = 2.0; cfg.thres_shell_factor_final = 1.0; cfg.clear_distant_voxels = false;
        cfg.occlusions_distance = 0.02f; cfg.num_observations = 5; cfg.subvolume_size_sh = 0.2f; cfg.sh_est_lambda_reg = 10.0;
        Optimizer::Config oc; oc.iterations = 1; oc.lm_steps = 1; oc.lambda_g = 0.2; oc.lambda_r0 = oc.lambda_r1 = oc.lambda_s0 = oc.lambda_s1 = 1.0; oc.lambda_a = 0.1; oc.fix_poses = oc.fix_intrinsics = oc.fix_distortion = false;
        R = new RefInit; R->sensor = s; R->ks = ks; R->app = new Intrinsic3DInit(cfg, oc, s, ks);
        R->grid = new SparseVoxelGrid<VoxelSBR>(0.004f, depth_min, depth_max);
        { VoxelSBR v; v.sdf = 0.0; v.sdf_refined = 0.0; v.albedo = 0.6; v.weight = 1.0f; v.color = Vec3b(128, 128, 128); R->grid->setVoxel(Vec3i(0, 0, 0), v); }
        R->app->opt_data_.grid = R->grid;
        (void)R->app->init_reference();
    } else delete s;
    std::cout.rdbuf(o1); std::cerr.rdbuf(o2);
    return R;
}
int32_t ref_i3d_init_count(void* h) { return (int32_t)((RefInit*)h)->app->image_model_.frame_ids.size(); }
void ref_i3d_init_model(void* h, int32_t* frame_ids, double* poses6, double* intr4, double* dist5) {
    auto& m = ((RefInit*)h)->app->image_model_;
    for (size_t i = 0; i < m.frame_ids.size(); ++i) { frame_ids[i] = m.frame_ids[i]; for (int k = 0; k < 6; ++k) poses6[6 * i + k] = m.poses[i][k]; }
    for (int k = 0; k < 4; ++k) intr4[k] = m.intrinsics[k];
    for (int k = 0; k < 5; ++k) dist5[k] = m.distortion_coeffs[(size_t)k];
}
/* image of keyframe k at pyramid level lvl: kind 0 intensity (float), 1 depth (float), 2 colour (3 x u8); returns rows * cols, dims in wh */
int64_t ref_i3d_init_image(void* h, int32_t k, int32_t lvl, int32_t kind, int32_t* wh, void* out) {
    Pyramid& p = ((RefInit*)h)->app->image_model_.rgbd_pyr[(size_t)k];
    const cv::Mat m = kind == 0 ? p.intensity(lvl) : kind == 1 ? p.depth(lvl) : p.color(lvl);
    if (m.empty()) return 0;
    wh[0] = m.cols; wh[1] = m.rows;
    if (out) std::memcpy(out, m.data, (size_t)m.rows * m.cols * cv::Mat::elem(m.type()));
    return (int64_t)m.rows * m.cols;
}
void ref_i3d_init_free(void* h) { RefInit* R = (RefInit*)h; if (!R) return; R->app->opt_data_.grid = nullptr; delete R->app; delete R->grid; delete R->ks; delete R->sensor; delete R; }

/* Intrinsic3D::Config::load + Optimizer::Config::load (intrinsic3d.cpp:58-80, optimizer.cpp:52-72) from (key, value) strings.
 * out[20]: num_grid_levels, num_rgbd_levels, thres_shell_factor, thres_shell_factor_final, clear_distant_voxels, occlusions_distance, num_observations,
 * subvolume_size_sh, sh_est_lambda_reg | iterations, lm_steps, lambda_g, lambda_r0, lambda_r1, lambda_s0, lambda_s1, lambda_a, fix_poses, fix_intrinsics, fix_distortion */
void ref_config_load(int32_t n, const char* const* keys, const char* const* values, double* out) {
    Settings cfg; for (int i = 0; i < n; ++i) cfg.set<std::string>(keys[i], values[i]);
    std::streambuf* o2 = std::cerr.rdbuf(nullptr);                  // (missing keys warn on stderr)
    Intrinsic3D::Config a; a.load(cfg); Optimizer::Config b; b.load(cfg);
    std::cerr.rdbuf(o2);
    const double v[20] = {(double)a.num_grid_levels, (double)a.num_rgbd_levels, a.thres_shell_factor, a.thres_shell_factor_final, (double)a.clear_distant_voxels, (double)a.occlusions_distance,
                          (double)a.num_observations, (double)a.subvolume_size_sh, a.sh_est_lambda_reg,
                          (double)b.iterations, (double)b.lm_steps, b.lambda_g, b.lambda_r0, b.lambda_r1, b.lambda_s0, b.lambda_s1, b.lambda_a, (double)b.fix_poses, (double)b.fix_intrinsics, (double)b.fix_distortion};
    for (int i = 0; i < 20; ++i) out[i] = v[i];
}

/* AppKeyframes::selectKeyframes (apps/src/app_keyframes.cpp:101-144) on a dataset folder: blur score of every frame, window selection, keyframes.txt */
int32_t ref_app_keyframes(const char* folder, int32_t max_frames, float depth_min, float depth_max, int32_t n, const char* const* keys, const char* const* values) {
    SensorI3d* s = new SensorI3d; s->setNumFramesMax(max_frames); s->setDepthMin(depth_min); s->setDepthMax(depth_max);
    const bool quiet = std::getenv("I3D_REF_VERBOSE") == nullptr;
    std::streambuf* o1 = std::cout.rdbuf(); std::streambuf* o2 = std::cerr.rdbuf();
    if (quiet) { std::cout.rdbuf(nullptr); std::cerr.rdbuf(nullptr); }
    bool ok = s->init(folder);
    if (ok) { Settings cfg; for (int i = 0; i < n; ++i) cfg.set<std::string>(keys[i], values[i]); AppKeyframes app; app.sensor_ = s; ok = app.selectKeyframes(cfg); } else delete s;
    std::cout.rdbuf(o1); std::cerr.rdbuf(o2);
    return ok ? 1 : 0;
}

/* KeyframeSelection (keyframe_selection.cpp:46-126, 139-310): the reference's class on caller data */
double ref_blur_score(const uint8_t* image, int32_t w, int32_t h, int32_t channels) {
    KeyframeSelection ks; return ks.estimateBlur(cv::Mat::wrap(h, w, channels == 3 ? CV_8UC3 : CV_8UC1, image));
}
void ref_keyframes_select(int32_t window, int64_t n, const double* scores, uint8_t* is_kf) {
    KeyframeSelection ks(window); ks.frame_scores_.assign(scores, scores + n); ks.selectKeyframes();
    for (int64_t i = 0; i < n; ++i) is_kf[i] = ks.isKeyframe((int)i) ? 1 : 0;
}
int32_t ref_keyframes_save(const char* path, int32_t window, int64_t n, const double* scores, const uint8_t* is_kf) {
    KeyframeSelection ks(window); ks.frame_scores_.assign(scores, scores + n); ks.is_keyframe_.resize((size_t)n); for (int64_t i = 0; i < n; ++i) ks.is_keyframe_[(size_t)i] = is_kf[i] != 0;
    return ks.save(path) ? 1 : 0;
}
int64_t ref_keyframes_load(const char* path, int32_t* window, int64_t cap, double* scores, uint8_t* is_kf) {
    KeyframeSelection ks(0); if (!ks.load(path)) return -1;
    *window = ks.window_size_; const int64_t n = (int64_t)ks.frame_scores_.size();
    for (int64_t i = 0; i < n && i < cap; ++i) { scores[i] = ks.frame_scores_[(size_t)i]; is_kf[i] = ks.is_keyframe_[(size_t)i] ? 1 : 0; }
    return n;
}
void ref_fusion_free(void* fp) { auto* f = (RefFusion*)fp; delete f->grid; delete f; }
void ref_erode_discontinuities(int32_t w, int32_t h, const float* in, int32_t window, float max_diff, float* out) {
    const cv::Mat r = erodeDiscontinuities(cv::Mat::wrap(h, w, CV_32FC1, in), window, max_diff); std::memcpy(out, r.data, (size_t)w * h * 4);
}
void ref_compute_normals(int32_t w, int32_t h, const float* c, const float* depth, float thr, float* normals) {
    Mat3f K = Mat3f::Identity(); K(0, 0) = c[0]; K(1, 1) = c[1]; K(0, 2) = c[2]; K(1, 2) = c[3];
    const cv::Mat r = computeNormals(K, cv::Mat::wrap(h, w, CV_32FC1, depth), thr); std::memcpy(normals, r.data, (size_t)w * h * 12);
}
void ref_resize_depth(int32_t iw, int32_t ih, const float* din, const float* in_intr, int32_t ow, int32_t oh, const float* out_intr, float* dout) {
    auto mk = [](const float* k, int w, int h) { Mat3f K = Mat3f::Identity(); K(0, 0) = k[0]; K(1, 1) = k[1]; K(0, 2) = k[2]; K(1, 2) = k[3]; return Camera(K, w, h); };
    const cv::Mat r = resizeDepth(mk(in_intr, iw, ih), cv::Mat::wrap(ih, iw, CV_32FC1, din), mk(out_intr, ow, oh)); std::memcpy(dout, r.data, (size_t)ow * oh * 4);
}
void ref_depth_down(int32_t w, int32_t h, const float* src, float* dst) {
    Pyramid p; const cv::Mat r = p.downsampleDepth(cv::Mat::wrap(h, w, CV_32FC1, src)); std::memcpy(dst, r.data, (size_t)(w / 2) * (h / 2) * 4);
}
void ref_pose_to_mat(const double* pose6, float* R9, float* t3) {
    Vec6 p; for (int i = 0; i < 6; ++i) p[i] = pose6[i];
    const Mat4f m = math::poseVecAAToMat(p).cast<float>();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R9[3 * r + c] = m(r, c); t3[r] = m(r, 3); }
}
void ref_interpolation_weights(const float* pos3, int32_t* coords24, float* weights8) {
    Vec3i c[8]; math::interpolationWeights(Vec3f(pos3[0], pos3[1], pos3[2]), c, weights8);
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 3; ++k) coords24[3 * i + k] = c[i][k];
}
void ref_surface_normal(void* g, const int32_t* key3, float* n3) {
    const Vec3f n = SDFOperators::computeSurfaceNormal((SparseVoxelGrid<VoxelSBR>*)g, Vec3i(key3[0], key3[1], key3[2])); n3[0] = n[0]; n3[1] = n[1]; n3[2] = n[2];
}

}  // extern "C"
