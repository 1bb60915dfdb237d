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
    ("mat_round",            "include/nv/mat.h", 88, 93, "inline Vec2i round(const Vec2f", "inline Vec4i round(const Vec4 "),
    ("mat_hash",             "include/nv/mat.h", 114, 125, "template <>", "};"),
    ("grid_voxels",          "include/nv/sparse_voxel_grid.h", 56, 77, "struct Voxel", "};"),
    ("grid_ctor",            "src/sparse_voxel_grid.cpp", 43, 54, "template <class T>", "}"),
    ("grid_access",          "src/sparse_voxel_grid.cpp", 165, 297, "template <class T>", "}"),
    ("operators_templates",  "include/nv/sdf/operators.h", 49, 109, "template <typename T>", "}"),
    ("operators_sdf_weight", "src/sdf/operators.cpp", 142, 147, "double sdfToWeight", "}"),
    ("math_robust_kernel",   "src/math.cpp", 43, 47, "float robustKernel", "}"),
    ("shading_basis",        "include/nv/shading.h", 53, 67, "template <typename T>", "}"),
    ("shading_compute",      "include/nv/shading.h", 73, 112, "template <typename T>", "}"),
    ("shading_graddiff",     "include/nv/shading.h", 128, 148, "template <typename T>", "}"),
    ("camera_t",             "include/nv/camera.h", 92, 126, "template <typename T>", "};"),
    ("camera_project_f",     "src/camera.cpp", 124, 154, "bool Camera::project(const Vec3f", "}"),
    ("cost_helpers",         "include/nv/refinement/cost.h", 73, 150, "template <typename T>", "}"),
    ("shading_cost_data",    "include/nv/refinement/shading_cost.h", 52, 73, "class ShadingCostData", "};"),
    ("shading_cost_functor", "include/nv/refinement/shading_cost.h", 85, 198, "template <typename T>", "}"),
    ("volreg_functor",       "include/nv/refinement/volumetric_regularizer.h", 59, 72, "template <typename T>", "}"),
    ("stab_functor",         "include/nv/refinement/surface_stab_regularizer.h", 59, 66, "template <typename T>", "}"),
    ("albedo_functor",       "include/nv/refinement/albedo_regularizer.h", 59, 66, "template <typename T>", "}"),
    ("albedo_chroma",        "src/refinement/albedo_regularizer.cpp", 61, 70, "Vec3f c = v.color", "double w ="),
    ("color_intensity",      "src/color_util.cpp", 41, 52, "float intensity(unsigned char r", "}"),
    ("sh_costs",             "src/lighting/lighting_svsh.cpp", 113, 163, "class SHDataCost", "};"),
    ("invalid_residual",     "include/nv/refinement/cost.h", 45, 45, "#define NV_INVALID_RESIDUAL", "#define NV_INVALID_RESIDUAL"),
    ("colorization_config",  "include/nv/sdf/colorization.h", 91, 98, "struct Config", "};"),
    ("vertex_observation",   "include/nv/sdf/colorization.h", 57, 78, "struct VertexObservation", "};"),
    ("vertex_observation_lt","src/sdf/colorization.cpp", 46, 49, "bool VertexObservation::operator<", "}"),
    ("colorization_weights", "src/sdf/colorization.cpp", 254, 370, "bool SDFColorization::isVoxelVisible", "}"),
    ("mesh_struct",          "include/nv/mesh.h", 45, 59, "struct Mesh", "};"),
    ("mesh_save",            "src/mesh.cpp", 41, 100, "bool Mesh::save", "}"),
    ("mesh_degenerate",      "src/mesh/util.cpp", 174, 200, "bool removeDegenerateFaces", "}"),
    ("mc_class",             "include/nv/mesh/marching_cubes.h", 50, 85, "template <class T>", "};"),
    ("mc_extract_mesh",      "src/mesh/marching_cubes.cpp", 43, 94, "template <class T>", "}"),
    ("mc_body",              "src/mesh/marching_cubes.cpp", 97, 317, "template <class T>", "}"),
    ("mc_tables",            "src/mesh/marching_cubes.cpp", 330, 623, "template <class T>", "};"),
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
