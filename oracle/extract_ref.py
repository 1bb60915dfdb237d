#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.

Recipe that builds oracle/_ref/libref_i3d.so FROM THE REFERENCE'S OWN SOURCES where they lie under /root/reference.

The reference library cannot be built (Ceres 2.1.0, Eigen, OpenCV and Boost are absent from this image), but the bodies below
only need scalar arithmetic plus a handful of Eigen / Ceres / OpenCV names.  This script cuts those bodies out of the reference
files BY FILE AND LINE RANGE into oracle/_ref/gen/*.inc (git-ignored: no reference source is ever committed) and compiles them
together with oracle/ref_shim/ (our stand-ins for the absent third-party names + a C ABI) into oracle/_ref/libref_i3d.so.
tests/test_oracle_vs_ref.py then checks the restated oracle (and, on the GPU box, the prebuilt library travels with the
snapshot) against code the reference authors wrote.

Every chunk is guarded: the first and last line of the range must contain the expected tokens, so a reference checkout whose
lines moved fails here instead of compiling the wrong text.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("I3D_REFERENCE", "/root/reference")
LIB = os.path.join(REF, "libintrinsic3d")
OUT = os.path.join(HERE, "_ref")
GEN = os.path.join(OUT, "gen")

# name, file (relative to libintrinsic3d/), first line, last line, token expected in the first line, token expected in the last line
CHUNKS = [
    # --- scalar helpers, functors, templates (round 2)
    ("mat_round",            "include/nv/mat.h", 88, 93, "inline Vec2i round(const Vec2f", "inline Vec4i round(const Vec4 "),
    ("mat_floor_ceil",       "include/nv/mat.h", 95, 107, "inline Vec2i floor(const Vec2f", "inline Vec4i ceil(const Vec4 "),
    ("mat_hash",             "include/nv/mat.h", 114, 125, "template <>", "};"),
    ("grid_voxels",          "include/nv/sparse_voxel_grid.h", 56, 77, "struct Voxel", "};"),
    ("operators_templates",  "include/nv/sdf/operators.h", 49, 109, "template <typename T>", "}"),
    ("operators_sdf_weight", "src/sdf/operators.cpp", 142, 147, "double sdfToWeight", "}"),
    ("shading_basis",        "include/nv/shading.h", 53, 67, "template <typename T>", "}"),
    ("shading_compute",      "include/nv/shading.h", 73, 112, "template <typename T>", "}"),
    ("shading_graddiff",     "include/nv/shading.h", 128, 148, "template <typename T>", "}"),
    ("camera_t",             "include/nv/camera.h", 92, 126, "template <typename T>", "};"),
    ("cost_helpers",         "include/nv/refinement/cost.h", 73, 150, "template <typename T>", "}"),
    ("shading_cost_data",    "include/nv/refinement/shading_cost.h", 52, 73, "class ShadingCostData", "};"),
    ("albedo_chroma",        "src/refinement/albedo_regularizer.cpp", 61, 70, "Vec3f c = v.color", "double w ="),
    ("invalid_residual",     "include/nv/refinement/cost.h", 45, 45, "#define NV_INVALID_RESIDUAL", "#define NV_INVALID_RESIDUAL"),
    ("vertex_observation",   "include/nv/sdf/colorization.h", 57, 78, "struct VertexObservation", "};"),
    ("vertex_observation_lt","src/sdf/colorization.cpp", 46, 49, "bool VertexObservation::operator<", "}"),
    ("mesh_struct",          "include/nv/mesh.h", 45, 59, "struct Mesh", "};"),
    ("mesh_save",            "src/mesh.cpp", 41, 100, "bool Mesh::save", "}"),
    ("mesh_degenerate",      "src/mesh/util.cpp", 174, 200, "bool removeDegenerateFaces", "}"),
    ("mesh_loose",           "src/mesh/util.cpp", 47, 101, "void removeLooseComponents", "}"),
    ("mesh_unused",          "src/mesh/util.cpp", 104, 171, "void removeUnusedVertices", "}"),
    ("mc_class",             "include/nv/mesh/marching_cubes.h", 50, 85, "template <class T>", "};"),
    ("mc_extract_mesh",      "src/mesh/marching_cubes.cpp", 43, 94, "template <class T>", "}"),
    ("mc_body",              "src/mesh/marching_cubes.cpp", 97, 317, "template <class T>", "}"),
    ("mc_tables",            "src/mesh/marching_cubes.cpp", 330, 623, "template <class T>", "};"),
    # --- classes and whole implementation files of the hot path (round 3): grid container incl. integrate / alloc, camera, math,
    #     level transitions, colourisation, the cost-function factories, NLSSolver, Optimizer, Subvolumes, LightingSVSH
    ("grid_class",           "include/nv/sparse_voxel_grid.h", 84, 161, "template <class T>", "};"),
    ("grid_impl",            "src/sparse_voxel_grid.cpp", 43, 467, "template <class T>", "}"),
    ("grid_frustum",         "src/sparse_voxel_grid.cpp", 572, 602, "template <class T>", "}"),
    ("grid_print_info",      "src/sparse_voxel_grid.cpp", 470, 480, "template <class T>", "}"),
    ("grid_save",            "src/sparse_voxel_grid.cpp", 483, 516, "template <class T>", "}"),
    ("grid_clone",           "src/sparse_voxel_grid.cpp", 519, 528, "template <class T>", "}"),
    ("grid_load",            "src/sparse_voxel_grid.cpp", 531, 569, "template <class T>", "}"),
    ("camera_class",         "include/nv/camera.h", 47, 89, "class Camera", "};"),
    ("camera_impl",          "src/camera.cpp", 41, 199, "Camera::Camera() :", "}"),
    ("camera_convert",       "src/camera.cpp", 277, 311, "void Camera::print", "}"),
    ("camera_load_save",     "src/camera.cpp", 202, 274, "bool Camera::load", "}"),
    ("settings_class",       "include/nv/settings.h", 48, 74, "class Settings", "};"),
    ("settings_impl",        "src/settings.cpp", 41, 138, "Settings::Settings()", "}"),
    ("kfs_draw",             "src/keyframe_selection.cpp", 129, 136, "void KeyframeSelection::drawScore", "}"),
    ("app_keyframes_class",  "../apps/include/nv/app_keyframes.h", 46, 58, "class AppKeyframes", "};"),
    ("app_keyframes_ctor",   "../apps/src/app_keyframes.cpp", 46, 55, "AppKeyframes::AppKeyframes() :", "}"),
    ("app_keyframes_select", "../apps/src/app_keyframes.cpp", 101, 144, "bool AppKeyframes::selectKeyframes", "}"),
    ("vis_class",            "include/nv/sdf/visualization.h", 56, 101, "class SDFVisualization", "};"),
    ("vis_impl",             "src/sdf/visualization.cpp", 60, 416, "SDFVisualization::SDFVisualization(SparseVoxelGrid<VoxelSBR>* grid", "}"),
    ("app_fusion_class",     "../apps/include/nv/app_fusion.h", 46, 58, "class AppFusion", "};"),
    ("app_i3d_class",        "../apps/include/nv/app_intrinsic3d.h", 53, 68, "class AppIntrinsic3D : public Intrinsic3D::RefinementCallback", "};"),
    ("app_i3d_ctor",         "../apps/src/app_intrinsic3d.cpp", 56, 69, "AppIntrinsic3D::AppIntrinsic3D() :", "}"),
    ("app_i3d_on_refined",   "../apps/src/app_intrinsic3d.cpp", 159, 210, "void AppIntrinsic3D::onSDFRefined", "}"),
    ("app_fusion_ctor",      "../apps/src/app_fusion.cpp", 52, 61, "AppFusion::AppFusion() :", "}"),
    ("app_fusion_fuse",      "../apps/src/app_fusion.cpp", 107, 200, "bool AppFusion::fuseSDF", "}"),
    ("sensor_class",         "include/nv/rgbd/sensor.h", 49, 112, "class Sensor", "};"),
    ("sensor_ctor",          "src/rgbd/sensor.cpp", 50, 63, "Sensor::Sensor() :", "}"),
    ("sensor_create",        "src/rgbd/sensor.cpp", 66, 118, "Sensor* Sensor::create(const std::string &dataset, Settings &cfg)", "}"),
    ("sensor_access",        "src/rgbd/sensor.cpp", 121, 220, "const Camera& Sensor::depthCamera() const", "}"),
    ("sensor_poses",         "src/rgbd/sensor.cpp", 235, 347, "bool Sensor::loadPoses", "}"),
    ("sensor_i3d_class",     "include/nv/rgbd/sensor_i3d.h", 50, 88, "class SensorI3d : public Sensor", "};"),
    ("sensor_i3d_impl",      "src/rgbd/sensor_i3d.cpp", 48, 345, "SensorI3d::SensorI3d() :", "}"),
    ("kfs_class",            "include/nv/keyframe_selection.h", 47, 75, "class KeyframeSelection", "};"),
    ("kfs_impl_a",           "src/keyframe_selection.cpp", 46, 126, "KeyframeSelection::KeyframeSelection(int window_size)", "}"),
    ("kfs_impl_b",           "src/keyframe_selection.cpp", 139, 310, "bool KeyframeSelection::load", "}"),
    ("math_decl",            "include/nv/math.h", 44, 65, "namespace math", "} // namespace math"),
    ("math_impl",            "src/math.cpp", 43, 163, "float robustKernel", "}"),
    ("operators_impl",       "src/sdf/operators.cpp", 45, 77, "Vec3f voxelCenterToIso(const SparseVoxelGrid", "}"),
    ("operators_lap_grad",   "src/sdf/operators.cpp", 80, 139, "float laplacian(const SparseVoxelGrid<VoxelSBR>* grid", "}"),
    ("algorithms_decl",      "include/nv/sdf/algorithms.h", 44, 69, "namespace SDFAlgorithms", "} // namespace SDFAlgorithms"),
    ("algorithms_impl",      "src/sdf/algorithms.cpp", 47, 458, "SparseVoxelGrid<VoxelSBR>* convert", "}"),
    ("color_util_decl",      "include/nv/color_util.h", 46, 57, "float intensity(unsigned char r", "Vec3b randomColor();"),
    ("color_intensity",      "src/color_util.cpp", 41, 58, "float intensity(unsigned char r", "}"),
    ("color_chroma",         "src/color_util.cpp", 61, 67, "Vec3f chromacity(const Vec3b &color)", "}"),
    ("color_scalar",         "src/color_util.cpp", 70, 80, "template <typename T>", "template Vec3b scalarToColor(const double val, const double scale);"),
    ("color_random",         "src/color_util.cpp", 83, 114, "Vec3f checkRange", "}"),
    ("processing_decl",      "include/nv/rgbd/processing.h", 50, 62, "cv::Mat computeVertexMap", "Vec3b interpolateRGB"),
    ("processing_impl",      "src/rgbd/processing.cpp", 49, 301, "cv::Mat computeVertexMap", "}"),
    ("pyramid_class",        "include/nv/rgbd/pyramid.h", 47, 69, "class Pyramid", "};"),
    ("pyramid_ctor",         "src/rgbd/pyramid.cpp", 43, 45, "Pyramid::Pyramid()", "}"),
    ("pyramid_ctor2",        "src/rgbd/pyramid.cpp", 48, 51, "Pyramid::Pyramid(int num_levels", "}"),
    ("pyramid_dtor",         "src/rgbd/pyramid.cpp", 54, 56, "Pyramid::~Pyramid()", "}"),
    ("pyramid_create",       "src/rgbd/pyramid.cpp", 59, 79, "bool Pyramid::create", "}"),
    ("pyramid_downsample",   "src/rgbd/pyramid.cpp", 108, 113, "cv::Mat Pyramid::downsample(const cv::Mat &img)", "}"),
    ("pyramid_create_pyr",   "src/rgbd/pyramid.cpp", 144, 152, "std::vector<cv::Mat> Pyramid::createPyramid", "}"),
    ("pyramid_access",       "src/rgbd/pyramid.cpp", 81, 105, "cv::Mat Pyramid::color", "}"),
    ("pyramid_depth_down",   "src/rgbd/pyramid.cpp", 116, 141, "cv::Mat Pyramid::downsampleDepth", "}"),
    ("pyramid_depth_pyr",    "src/rgbd/pyramid.cpp", 155, 166, "std::vector<cv::Mat> Pyramid::createDepthPyramid", "}"),
    ("colorization_class",   "include/nv/sdf/colorization.h", 87, 136, "class SDFColorization", "};"),
    ("colorization_impl",    "src/sdf/colorization.cpp", 52, 370, "SDFColorization::SDFColorization(const Camera", "}"),
    ("shading_decl",         "include/nv/shading.h", 49, 51, "static const int NUM_SPHERICAL_HARMONICS", "Eigen::VectorXf shBasisFunctions"),
    ("shading_basis_f",      "src/shading.cpp", 43, 58, "Eigen::VectorXf shBasisFunctions", "}"),
    ("shading_compute_f",    "src/shading.cpp", 61, 73, "float computeShading(const Vec3f &normal", "}"),
    ("voxel_residual",       "include/nv/refinement/cost.h", 59, 70, "struct VoxelResidual", "};"),
    ("shading_cost_class",   "include/nv/refinement/shading_cost.h", 76, 204, "class ShadingCost", "};"),
    ("shading_cost_impl",    "src/refinement/shading_cost.cpp", 46, 150, "ShadingCost::ShadingCost(const Vec3i", "}"),
    ("volreg_class",         "include/nv/refinement/volumetric_regularizer.h", 51, 76, "class VolumetricRegularizer", "};"),
    ("volreg_impl",          "src/refinement/volumetric_regularizer.cpp", 42, 78, "VolumetricRegularizer::VolumetricRegularizer()", "}"),
    ("stab_class",           "include/nv/refinement/surface_stab_regularizer.h", 51, 69, "class SurfaceStabRegularizer", "};"),
    ("stab_impl",            "src/refinement/surface_stab_regularizer.cpp", 40, 61, "SurfaceStabRegularizer::SurfaceStabRegularizer(double", "}"),
    ("albedo_class",         "include/nv/refinement/albedo_regularizer.h", 51, 69, "class AlbedoRegularizer", "};"),
    ("albedo_impl",          "src/refinement/albedo_regularizer.cpp", 40, 84, "AlbedoRegularizer::AlbedoRegularizer()", "}"),
    ("timer_class",          "include/nv/timer.h", 45, 80, "class Timer", "};"),
    ("nls_class",            "include/nv/refinement/nls_solver.h", 53, 125, "class NLSSolver", "};"),
    ("nls_impl",             "src/refinement/nls_solver.cpp", 45, 394, "NLSSolver::ProblemInfo::ProblemInfo()", "}"),
    ("optimizer_class",      "include/nv/refinement/optimizer.h", 59, 141, "class Optimizer", "};"),
    ("optimizer_impl",       "src/refinement/optimizer.cpp", 92, 361, "Optimizer::Optimizer(Config cfg)", "}"),
    ("subvolumes_class",     "include/nv/lighting/subvolumes.h", 47, 91, "class Subvolumes", "};"),
    ("subvolumes_impl",      "src/lighting/subvolumes.cpp", 43, 304, "Subvolumes::Subvolumes(float size)", "}"),
    ("svsh_class",           "include/nv/lighting/lighting_svsh.h", 46, 71, "class LightingSVSH", "};"),
    ("svsh_impl",            "src/lighting/lighting_svsh.cpp", 54, 346, "LightingSVSH::LightingSVSH(const SparseVoxelGrid", "}"),
    # --- the level schedule itself: Intrinsic3D::refine / prepare* / finish* / recomputeColors (init() is ours: it needs Sensor + OpenCV)
    ("i3d_class",            "include/nv/refinement/intrinsic3d.h", 59, 155, "class Intrinsic3D", "};"),
    ("i3d_callback_dtor",    "src/refinement/intrinsic3d.cpp", 53, 55, "Intrinsic3D::RefinementCallback::~RefinementCallback", "}"),
    ("i3d_cfg_load",         "src/refinement/intrinsic3d.cpp", 58, 80, "void Intrinsic3D::Config::load", "}"),
    ("opt_cfg_load",         "src/refinement/optimizer.cpp", 52, 72, "void Optimizer::Config::load", "}"),
    ("math_pose_mat_to_vec", "src/math.cpp", 166, 179, "Vec6 poseMatToVecAA", "}"),
    ("i3d_init",             "src/refinement/intrinsic3d.cpp", 151, 203, "bool Intrinsic3D::init()", "}"),
    ("i3d_ctor",             "src/refinement/intrinsic3d.cpp", 98, 148, "Intrinsic3D::Intrinsic3D(Config cfg", "}"),
    ("i3d_refine",           "src/refinement/intrinsic3d.cpp", 206, 409, "bool Intrinsic3D::refine", "}"),
]


def extract() -> None:
    os.makedirs(GEN, exist_ok=True)
    for name, rel, first, last, tok0, tok1 in CHUNKS:
        path = os.path.join(LIB, rel)
        with open(path, "r", encoding="utf-8", errors="replace") as f:
            lines = f.read().split("\n")
        body = lines[first - 1:last]
        if tok0 not in body[0] or tok1 not in body[-1]:
            raise SystemExit(f"extract_ref: {rel}:{first}-{last} does not start/end as expected "
                             f"({body[0].strip()!r} ... {body[-1].strip()!r}); the reference checkout differs from the one this recipe was written for")
        with open(os.path.join(GEN, name + ".inc"), "w", encoding="utf-8") as f:
            f.write(f"// generated from {rel}:{first}-{last} — NOT committed\n#line {first} \"{path}\"\n" + "\n".join(body) + "\n")


def build() -> str:
    so = os.path.join(OUT, "libref_i3d.so")
    src = os.path.join(HERE, "ref_shim", "ref_capi.cpp")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++14", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wno-deprecated-declarations",
           "-I", os.path.join(HERE, "ref_shim"), "-I", OUT, "-o", so, src]
    subprocess.check_call(cmd)
    return so


def main() -> int:
    if not os.path.isdir(LIB):
        print(f"extract_ref: {LIB} not present — keeping the prebuilt oracle/_ref (if any)")
        return 0
    extract()
    print("built", build())
    return 0


if __name__ == "__main__":
    sys.exit(main())
