"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_ref/libref_i3d.so: bodies of the reference itself (cut out of /root/reference by oracle/extract_ref.py
at build time, compiled over stand-ins for the absent Eigen / Ceres / OpenCV).  Only tests/ and __graft_entry__.build() use it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_i3d.so")
_lib = None


def build() -> str | None:
    """(Re)build from /root/reference when it is present; otherwise keep whatever prebuilt library travelled with the tree."""
    subprocess.check_call([sys.executable, os.path.join(_HERE, "extract_ref.py")])
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
        L.ref_hash.restype = C.c_uint64; L.ref_hash.argtypes = [i32, i32, i32]
        L.ref_round3f.argtypes = [vp, vp]; L.ref_round3d.argtypes = [vp, vp]
        L.ref_sdf_to_weight.restype = f64; L.ref_sdf_to_weight.argtypes = [f64, f64]
        L.ref_robust_kernel.restype = f32; L.ref_robust_kernel.argtypes = [f32]
        L.ref_varying_lambda.restype = f64; L.ref_varying_lambda.argtypes = [i32, i32, f64, f64]
        L.ref_pyramid_scale.restype = f64; L.ref_pyramid_scale.argtypes = [i32]
        L.ref_shading_row.restype = f64; L.ref_shading_row.argtypes = [i32, i32, i32, vp, i32, f64, i32, i32, vp, vp, vp, vp]
        L.ref_project_t.restype = i32; L.ref_project_t.argtypes = [vp, vp, i32, i32, vp, vp]
        L.ref_project_f.restype = i32; L.ref_project_f.argtypes = [vp, vp, i32, i32, vp, vp, vp]
        L.ref_bicubic.argtypes = [vp, i32, i32, f64, f64, vp, vp, vp]
        L.ref_transform_voxel_iso.argtypes = [f64, vp, vp, f64, vp, vp]
        L.ref_compute_normal.argtypes = [f64, f64, f64, f64, vp]
        L.ref_volumetric.argtypes = [vp, vp, vp]; L.ref_surface_stab.argtypes = [f64, f64, vp, vp]; L.ref_albedo_reg.argtypes = [f64, f64, vp, vp]
        L.ref_chroma_weight.restype = f64; L.ref_chroma_weight.argtypes = [vp, vp]
        L.ref_sh_data_cost.argtypes = [f64, vp, f64, vp, vp, vp]; L.ref_sh_reg_cost.argtypes = [vp, vp, vp]
        L.ref_voxel_visible.restype = i32; L.ref_voxel_visible.argtypes = [f32, vp, i32, i32, vp, i32, i32]
        L.ref_observation_weight.restype = f32; L.ref_observation_weight.argtypes = [i32, i32, vp, vp, i32, i32, vp]
        L.ref_compute_color.argtypes = [i32, vp, vp, vp]; L.ref_filter.argtypes = [i32, vp, i32, vp]
        L.ref_grid_visit_order.argtypes = [f32, i64, vp, vp]; L.ref_world_to_voxel.argtypes = [f32, vp, vp]
        L.ref_truncation.restype = f32; L.ref_truncation.argtypes = [f32]
        L.ref_mc_extract.restype = vp; L.ref_mc_extract.argtypes = [f32, i64, vp, vp, vp, vp]
        L.ref_mesh_counts.argtypes = [vp, vp, vp]; L.ref_mesh_get.argtypes = [vp, vp, vp, vp]
        L.ref_mesh_save.restype = i32; L.ref_mesh_save.argtypes = [vp, C.c_char_p]; L.ref_mesh_free.argtypes = [vp]
        L.ref_mc_tables.argtypes = [vp, vp]
        _lib = L
    return _lib


_pipeline = None


def pipeline():
    """The reference's own pipeline (Optimizer / NLSSolver / SDFColorization / SDFAlgorithms / Subvolumes / LightingSVSH / SparseVoxelGrid::integrate /
    Intrinsic3D::refine compiled from /root/reference, ceres::Solve = oracle/ref_shim/mini_ceres_solver.hpp) behind the SAME Python classes as the oracle:
    a second instance of oracle_py whose library handle maps orc_* onto this library's ref_* exports.  `R = ref_py.pipeline(); R.Grid.from_voxels(...)`."""
    global _pipeline
    if _pipeline is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("oracle._ref_pipeline", os.path.join(_HERE, "oracle_py.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        mod._lib = mod._configure(mod._Prefixed(C.CDLL(LIB_PATH), "ref_"))
        mod.build = lambda force=False: LIB_PATH
        _pipeline = mod
    return _pipeline


def shading_row(v, sh9, rgbd_level, voxel_size, lum, params29):
    """ShadingCost::operator() of the reference: (residual via Jets, 29 partials, residual via T = double)."""
    lum = np.ascontiguousarray(lum, np.float32); sh9 = np.ascontiguousarray(sh9, np.float64); prm = np.ascontiguousarray(params29, np.float64)
    J = np.zeros(29); val = C.c_double()
    r = lib().ref_shading_row(int(v[0]), int(v[1]), int(v[2]), _p(sh9), int(rgbd_level), float(voxel_size), lum.shape[1], lum.shape[0], _p(lum), _p(prm), _p(J), C.byref(val))
    return r, J, val.value


def marching_cubes(voxel_size, keys, sdf, weight, color, save_path=None):
    """MarchingCubes<VoxelSBR>::extractSurface of a grid filled by inserting the records in the given order."""
    keys = np.ascontiguousarray(keys, np.int32); sdf = np.ascontiguousarray(sdf, np.float64)
    weight = np.ascontiguousarray(weight, np.float32); color = np.ascontiguousarray(color, np.uint8)
    L = lib(); m = L.ref_mc_extract(float(voxel_size), keys.shape[0], _p(keys), _p(sdf), _p(weight), _p(color))
    nv = C.c_int64(); nf = C.c_int64(); L.ref_mesh_counts(m, C.byref(nv), C.byref(nf))
    v = np.zeros((nv.value, 3), np.float32); c = np.zeros((nv.value, 3), np.uint8); f = np.zeros((nf.value, 3), np.int32)
    if m:
        L.ref_mesh_get(m, _p(v), _p(c), _p(f))
        if save_path is not None:
            assert L.ref_mesh_save(m, save_path.encode()) == 1
        L.ref_mesh_free(m)
    return v, c, f


def mc_tables():
    edge = np.zeros(256, np.int32); tri = np.zeros((256, 16), np.int32)
    lib().ref_mc_tables(_p(edge), _p(tri))
    return edge, tri


# --- on-disk formats: the reference's own writers / readers (sparse_voxel_grid.cpp:484-569, camera.cpp:202-274)
def _raw():
    L = C.CDLL(LIB_PATH)
    L.ref_fusion_load.restype = C.c_void_p; L.ref_grid_load.restype = C.c_void_p
    L.ref_fusion_size.restype = C.c_int64; L.ref_grid_size.restype = C.c_int64
    vp = C.c_void_p
    L.ref_fusion_save.argtypes = [vp, C.c_char_p]; L.ref_grid_save.argtypes = [vp, C.c_char_p]; L.ref_fusion_size.argtypes = [vp]; L.ref_fusion_free.argtypes = [vp]
    L.ref_fusion_export.argtypes = [vp] * 5; L.ref_fusion_header.argtypes = [vp] * 4
    return L


def tsdf_save(fusion, path):
    """SparseVoxelGrid<Voxel>::save of a `pipeline().Fusion` object's grid"""
    assert _raw().ref_fusion_save(fusion.h, str(path).encode()) == 1


def tsdf_load(path):
    """SparseVoxelGrid<Voxel>::load, then the records in the loaded container's iteration order + the three header floats; None if the file does not open"""
    L = _raw(); h = L.ref_fusion_load(str(path).encode())
    if not h:
        return None
    h = C.c_void_p(h); n = int(L.ref_fusion_size(h))
    keys = np.zeros((n, 3), np.int32); sdf = np.zeros(n, np.float32); w = np.zeros(n, np.float32); col = np.zeros((n, 3), np.uint8)
    L.ref_fusion_export(h, _p(keys), _p(sdf), _p(w), _p(col))
    vs = C.c_float(); tr = C.c_float(); iws = C.c_float(); L.ref_fusion_header(h, C.byref(vs), C.byref(tr), C.byref(iws))
    L.ref_fusion_free(h)
    return dict(keys=keys, sdf=sdf, weight=w, color=col, voxel_size=np.float32(vs.value), truncation=np.float32(tr.value), integration_weight_sample=np.float32(iws.value))


def sbr_save(grid, path):
    """SparseVoxelGrid<VoxelSBR>::save of a `pipeline().Grid`"""
    assert _raw().ref_grid_save(grid.h, str(path).encode()) == 1


def sbr_load(path):
    """SparseVoxelGrid<VoxelSBR>::load -> `pipeline().Grid` (None if the file does not open)"""
    h = _raw().ref_grid_load(str(path).encode())
    return pipeline().Grid(C.c_void_p(h)) if h else None


def camera_save(path, w, h, k4, dist5):
    k = np.ascontiguousarray(k4, np.float32); d = np.ascontiguousarray(dist5, np.float32)
    return _raw().ref_camera_save(str(path).encode(), int(w), int(h), _p(k), _p(d)) == 1


def camera_load(path):
    w = C.c_int32(); h = C.c_int32(); k = np.zeros(4, np.float32); d = np.zeros(5, np.float32)
    ok = _raw().ref_camera_load(str(path).encode(), C.byref(w), C.byref(h), _p(k), _p(d)) == 1
    return ok, w.value, h.value, k, d


# --- KeyframeSelection (keyframe_selection.cpp): the reference's class on caller data
def blur_score(image):
    a = np.ascontiguousarray(image, np.uint8); L = _raw(); L.ref_blur_score.restype = C.c_double
    return float(L.ref_blur_score(_p(a), C.c_int32(a.shape[1]), C.c_int32(a.shape[0]), C.c_int32(1 if a.ndim == 2 else a.shape[2])))


def keyframes_select(window, scores):
    s = np.ascontiguousarray(scores, np.float64); out = np.zeros(len(s), np.uint8)
    _raw().ref_keyframes_select(C.c_int32(window), C.c_int64(len(s)), _p(s), _p(out))
    return out.astype(bool)


def keyframes_save(path, window, scores, is_kf):
    s = np.ascontiguousarray(scores, np.float64); k = np.ascontiguousarray(is_kf, np.uint8)
    return _raw().ref_keyframes_save(str(path).encode(), C.c_int32(window), C.c_int64(len(s)), _p(s), _p(k)) == 1


def keyframes_load(path, cap=1 << 16):
    L = _raw(); L.ref_keyframes_load.restype = C.c_int64
    w = C.c_int32(); s = np.zeros(cap); k = np.zeros(cap, np.uint8)
    n = int(L.ref_keyframes_load(str(path).encode(), C.byref(w), C.c_int64(cap), _p(s), _p(k)))
    return (None if n < 0 else (w.value, s[:n], k[:n].astype(bool)))


def mesh_remove_loose_components(vertices, colors, faces):
    """MeshUtil::removeLooseComponents (+ removeUnusedVertices) of the reference on arrays -> (vertices, colors or None, faces)"""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3); f = np.ascontiguousarray(faces, np.int32).reshape(-1, 3)
    c = None if colors is None else np.ascontiguousarray(colors, np.uint8).reshape(-1, 3)
    L = _raw(); L.ref_mesh_from_arrays.restype = C.c_void_p
    m = C.c_void_p(L.ref_mesh_from_arrays(C.c_int64(len(v)), _p(v), None if c is None else _p(c), C.c_int64(len(f)), _p(f)))
    L.ref_mesh_remove_loose.argtypes = [C.c_void_p]; L.ref_mesh_counts.argtypes = [C.c_void_p] * 3; L.ref_mesh_get.argtypes = [C.c_void_p] * 4; L.ref_mesh_free.argtypes = [C.c_void_p]
    L.ref_mesh_remove_loose(m)
    nv = C.c_int64(); nf = C.c_int64(); L.ref_mesh_counts(m, C.byref(nv), C.byref(nf))
    vo = np.zeros((nv.value, 3), np.float32); co = None if c is None else np.zeros((nv.value, 3), np.uint8); fo = np.zeros((nf.value, 3), np.int32)
    L.ref_mesh_get(m, _p(vo), None if co is None else _p(co), _p(fo)); L.ref_mesh_free(m)
    return vo, co, fo


# --- SensorI3d (rgbd/sensor_i3d.cpp, rgbd/sensor.cpp): the reference's dataset-folder sensor; PNG decoding supplied by Pillow through cv::imdecode's hook
_HOOK_T = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_ubyte), C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_ubyte))
_hook_keep = None


def _install_pillow_decoder(L):
    global _hook_keep
    if _hook_keep is not None:
        return
    import io
    from PIL import Image

    def decode(buf, size, rows, cols, typ, out):
        try:
            im = Image.open(io.BytesIO(bytes(bytearray(buf[:size])))); im.load()
        except Exception:
            return 0
        a = np.asarray(im)
        if a.ndim == 3:
            a = np.ascontiguousarray(a[:, :, 2::-1][:, :, :3], np.uint8); t = 16           # cv::imdecode hands out B, G, R
        elif a.dtype == np.uint8:
            a = np.ascontiguousarray(a, np.uint8); t = 0
        else:
            a = np.ascontiguousarray(a, np.uint16); t = 2
        rows[0], cols[0], typ[0] = a.shape[0], a.shape[1], t
        if out:
            C.memmove(out, a.ctypes.data, a.nbytes)
        return 1
    _hook_keep = _HOOK_T(decode)
    L.ref_set_imdecode_hook(_hook_keep)


class Sensor:
    """SensorI3d::init on a dataset folder + Sensor::depth / color / pose"""

    def __init__(self, folder=None, max_frames=0, min_depth=0.0, max_depth=0.0, cfg=None):
        self.L = C.CDLL(LIB_PATH); _install_pillow_decoder(self.L)
        self.L.ref_sensor_open.restype = C.c_void_p; self.L.ref_sensor_create.restype = C.c_void_p; self.L.ref_sensor_depth.restype = C.c_int64; self.L.ref_sensor_color.restype = C.c_int64
        for f in ("ref_sensor_info", "ref_sensor_pose", "ref_sensor_depth", "ref_sensor_color", "ref_sensor_free"):
            getattr(self.L, f).argtypes = None
        self.depth_range = (float(min_depth), float(max_depth)); self.max_frames = max_frames
        if cfg is not None:                                           # Sensor::create(Settings&) with the strings of a sensor.yml
            ks = [str(k).encode() for k in cfg]; vs = [str(v).encode() for v in cfg.values()]
            r2 = np.zeros(2, np.float32); mf = C.c_int32()
            h = self.L.ref_sensor_create(C.c_int32(len(ks)), (C.c_char_p * len(ks))(*ks), (C.c_char_p * len(vs))(*vs), _p(r2), C.byref(mf))
            self.depth_range = (float(r2[0]), float(r2[1])); self.max_frames = mf.value
        else:
            h = self.L.ref_sensor_open(str(folder).encode(), C.c_int32(max_frames), C.c_float(min_depth), C.c_float(max_depth))
        self.h = C.c_void_p(h) if h else None
        if self.h is None:
            raise RuntimeError("SensorI3d::init failed")
        nf = C.c_int32(); ns = C.c_int32(); cwh = np.zeros(2, np.int32); dwh = np.zeros(2, np.int32); ci = np.zeros(4, np.float32); di = np.zeros(4, np.float32)
        self.L.ref_sensor_info(self.h, C.byref(nf), C.byref(ns), _p(cwh), _p(dwh), _p(ci), _p(di))
        self.num_frames, self.num_stored = nf.value, ns.value
        self.color_size, self.depth_size, self.color_intrinsics, self.depth_intrinsics = tuple(int(x) for x in cwh), tuple(int(x) for x in dwh), ci, di

    def pose(self, i):
        m = np.zeros((4, 4), np.float32); self.L.ref_sensor_pose(self.h, C.c_int32(i), _p(m)); return m

    def depth(self, i):
        out = np.zeros((self.depth_size[1], self.depth_size[0]), np.float32)
        return out if self.L.ref_sensor_depth(self.h, C.c_int32(i), _p(out)) else None

    def color(self, i):
        out = np.zeros((self.color_size[1], self.color_size[0], 3), np.uint8)
        return out if self.L.ref_sensor_color(self.h, C.c_int32(i), _p(out)) else None

    def set_pose(self, i, cam_to_world):
        m = np.ascontiguousarray(cam_to_world, np.float32); self.L.ref_sensor_set_pose(self.h, C.c_int32(i), _p(m))

    def save_poses(self, path):
        return self.L.ref_sensor_save_poses(self.h, str(path).encode()) == 1

    def close(self):
        if self.h:
            self.L.ref_sensor_free(self.h); self.h = None


def load_poses(path, first_is_identity=False, cap=4096):
    """Sensor::loadPoses (static): TUM trajectory lines -> (timestamps, camera-to-world 4x4 float matrices); None if the file does not open"""
    L = _raw(); L.ref_load_poses.restype = C.c_int64
    ts = np.zeros(cap); m = np.zeros((cap, 4, 4), np.float32)
    n = int(L.ref_load_poses(str(path).encode(), C.c_int32(1 if first_is_identity else 0), C.c_int64(cap), _p(ts), _p(m)))
    return None if n < 0 else (ts[:n], m[:n])


def app_fusion(folder, cfg, max_frames=0, min_depth=0.0, max_depth=0.0):
    """AppFusion::fuseSDF of the reference on a dataset folder (its own SensorI3d, KeyframeSelection, grid, marching cubes); cfg: dict for nv::Settings
    (keyframes, voxel_size, clip_x0..clip_z1, discont_window_size, output_sdf, output_mesh)."""
    L = C.CDLL(LIB_PATH); _install_pillow_decoder(L)
    ks = [str(k).encode() for k in cfg]; vs = [str(v).encode() for v in cfg.values()]
    K = (C.c_char_p * len(ks))(*ks); V = (C.c_char_p * len(vs))(*vs)
    return L.ref_app_fusion(str(folder).encode(), C.c_int32(max_frames), C.c_float(min_depth), C.c_float(max_depth), C.c_int32(len(ks)), K, V) == 1


def app_keyframes(folder, cfg, max_frames=0, min_depth=0.0, max_depth=0.0):
    """AppKeyframes::selectKeyframes of the reference on a dataset folder; cfg: dict for nv::Settings (filename, window_size, show_keyframes)"""
    L = C.CDLL(LIB_PATH); _install_pillow_decoder(L)
    ks = [str(k).encode() for k in cfg]; vs = [str(v).encode() for v in cfg.values()]
    K = (C.c_char_p * len(ks))(*ks); V = (C.c_char_p * len(vs))(*vs)
    return L.ref_app_keyframes(str(folder).encode(), C.c_int32(max_frames), C.c_float(min_depth), C.c_float(max_depth), C.c_int32(len(ks)), K, V) == 1


def config_load(cfg):
    """Intrinsic3D::Config::load + Optimizer::Config::load of the reference from a dict of strings -> dict of the 20 loaded values"""
    L = _raw(); ks = [str(k).encode() for k in cfg]; vs = [str(v).encode() for v in cfg.values()]
    K = (C.c_char_p * len(ks))(*ks); V = (C.c_char_p * len(vs))(*vs); out = np.zeros(20)
    L.ref_config_load(C.c_int32(len(ks)), K, V, _p(out))
    names = ["num_grid_levels", "num_rgbd_levels", "thin_shell_factor", "thin_shell_factor_final", "clear_distant_voxels", "occlusion_distance", "num_observations",
             "subvolume_size_sh", "sh_lambda_reg", "iterations", "lm_steps", "lambda_g", "lambda_r0", "lambda_r1", "lambda_s0", "lambda_s1", "lambda_a",
             "fix_poses", "fix_intrinsics", "fix_distortion"]
    return dict(zip(names, out.tolist()))


def albedo_colors(albedo):
    """scalarToColor(albedo, 255.0) per voxel: the grey SDFVisualization::applyColorAlbedo paints before the "albedo" mesh is extracted"""
    a = np.ascontiguousarray(albedo, np.float64); out = np.zeros((len(a), 3), np.uint8)
    _raw().ref_albedo_colors(C.c_int64(len(a)), _p(a), _p(out)); return out


def visualization_colors(mode, voxel_size, keys, sdf_refined, albedo, weight, color, subvolume_size=0.0, sub_index=None, sub_sh=None):
    """SDFVisualization::applyColor<mode> of the reference on caller arrays -> (colours [n, 3], position of every voxel in the reference's walk over its grid)"""
    k = np.ascontiguousarray(keys, np.int32); n = len(k); out = np.zeros((n, 3), np.uint8)
    s = np.ascontiguousarray(sdf_refined, np.float64); a = np.ascontiguousarray(albedo, np.float64); w = np.ascontiguousarray(weight, np.float32); c = np.ascontiguousarray(color, np.uint8)
    si = np.ascontiguousarray(sub_index if sub_index is not None else np.zeros((0, 3)), np.int32); ss = np.ascontiguousarray(sub_sh if sub_sh is not None else np.zeros((0, 9)), np.float64)
    L = _raw(); L.ref_visualization_colors.restype = C.c_int32
    L.ref_visualization_colors.argtypes = [C.c_char_p, C.c_float, C.c_int64] + [C.c_void_p] * 5 + [C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rank = np.zeros(n, np.int64)
    rc = L.ref_visualization_colors(mode.encode(), float(voxel_size), n, _p(k), _p(s), _p(a), _p(w), _p(c), float(subvolume_size), len(si), _p(si), _p(ss), _p(out), _p(rank))
    if rc < 0:
        raise ValueError(mode)
    return out, rank


class InitModel:
    """Intrinsic3D::init of the reference on a dataset folder: keyframe ids, world-to-camera pose vectors, intrinsics, and the keyframe pyramids"""

    def __init__(self, folder, is_keyframe, num_rgbd_levels, max_frames=0, min_depth=0.0, max_depth=0.0):
        self.L = C.CDLL(LIB_PATH); _install_pillow_decoder(self.L)
        self.L.ref_i3d_init.restype = C.c_void_p; self.L.ref_i3d_init_image.restype = C.c_int64
        kf = np.ascontiguousarray(is_keyframe, np.uint8)
        h = self.L.ref_i3d_init(str(folder).encode(), C.c_int32(max_frames), C.c_float(min_depth), C.c_float(max_depth), C.c_int64(len(kf)), _p(kf), C.c_int32(num_rgbd_levels))
        if not h:
            raise RuntimeError("SensorI3d::init failed")
        self.h = C.c_void_p(h)
        n = int(self.L.ref_i3d_init_count(self.h))
        self.frame_ids = np.zeros(n, np.int32); self.poses = np.zeros((n, 6)); self.intrinsics = np.zeros(4); self.distortion = np.zeros(5)
        self.L.ref_i3d_init_model(self.h, _p(self.frame_ids), _p(self.poses), _p(self.intrinsics), _p(self.distortion))

    def image(self, k, level, kind):
        """kind: 'lum' | 'depth' | 'bgr'"""
        code = {"lum": 0, "depth": 1, "bgr": 2}[kind]; wh = np.zeros(2, np.int32)
        if not self.L.ref_i3d_init_image(self.h, C.c_int32(k), C.c_int32(level), C.c_int32(code), _p(wh), None):
            return None
        out = np.zeros((wh[1], wh[0], 3), np.uint8) if code == 2 else np.zeros((wh[1], wh[0]), np.float32)
        self.L.ref_i3d_init_image(self.h, C.c_int32(k), C.c_int32(level), C.c_int32(code), _p(wh), _p(out)); return out

    def close(self):
        if self.h:
            self.L.ref_i3d_init_free(self.h); self.h = None


def app_intrinsic3d(folder, cfg, max_frames=0, min_depth=0.0, max_depth=0.0):
    """AppIntrinsic3D::run of the reference without its command line / yml reading (see ref_app_intrinsic3d); cfg: dict of intrinsic3d.yml.  Paths in cfg are
    used as given (the reference changes into the sensor config's directory first: hand in absolute paths or chdir)."""
    L = C.CDLL(LIB_PATH); _install_pillow_decoder(L)
    ks = [str(k).encode() for k in cfg]; vs = [str(v).encode() for v in cfg.values()]
    K = (C.c_char_p * len(ks))(*ks); V = (C.c_char_p * len(vs))(*vs)
    return L.ref_app_intrinsic3d(str(folder).encode(), C.c_int32(max_frames), C.c_float(min_depth), C.c_float(max_depth), C.c_int32(len(ks)), K, V) == 1
