"""ctypes binding of libintrinsic3d_hip.so (include/intrinsic3d_hip.h).

Python is only the test / bench harness here: the host side of the product (Optimizer::optimize mirror, LM/PCG
control, lighting solve) is C++ inside the library.  Loading fails loudly when the HIP library is missing — there is
no CPU fallback anywhere on the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# I3D_LIB: load another build of the SAME C ABI instead (same-box A/B of kernel variants in one GPU session; never a CPU substitute)
LIB_PATH = os.environ.get("I3D_LIB") or os.path.join(_HERE, "libintrinsic3d_hip.so")

K_NAMES = ["classify", "observe", "build", "eg_pass", "gather", "cost", "vector", "sh", "eg_aux", "comm", "eg_mr2", "eg_mr3"]


class OptimizerConfig(C.Structure):
    """Optimizer::Config (optimizer.h:67-84) + the Intrinsic3D::Config / Optimizer::Data fields the path reads."""
    _fields_ = [("iterations", C.c_int32), ("lm_steps", C.c_int32),
                ("lambda_g", C.c_double), ("lambda_r0", C.c_double), ("lambda_r1", C.c_double),
                ("lambda_s0", C.c_double), ("lambda_s1", C.c_double), ("lambda_a", C.c_double),
                ("fix_poses", C.c_int32), ("fix_intrinsics", C.c_int32), ("fix_distortion", C.c_int32),
                ("occlusion_distance", C.c_float), ("num_observations", C.c_int32),
                ("thres_shell", C.c_double), ("grid_level", C.c_int32), ("rgbd_level", C.c_int32),
                ("pcg_fixed_iterations", C.c_int32), ("verbose", C.c_int32), ("carry_trust_radius", C.c_int32), ("fix_sdf", C.c_int32)]


class IterationStats(C.Structure):
    _fields_ = [("rows", C.c_int64 * 4), ("weight_sum", C.c_double * 4), ("type_weight", C.c_double * 4),
                ("valid_voxels", C.c_int64), ("free_parameters", C.c_int64),
                ("cost_initial", C.c_double), ("cost_final", C.c_double),
                ("lm_iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
                ("pcg_iterations", C.c_int32 * 50), ("step_accepted", C.c_int32 * 50), ("num_attempts", C.c_int32),
                ("final_radius", C.c_double), ("time_add", C.c_double), ("time_build", C.c_double), ("time_solve", C.c_double)]


class ShStats(C.Structure):
    _fields_ = [("data_rows", C.c_int64), ("reg_rows", C.c_int64), ("subvolumes", C.c_int32), ("lm_iterations", C.c_int32),
                ("termination", C.c_int32), ("cost_initial", C.c_double), ("cost_final", C.c_double)]


class RefineConfig(C.Structure):
    """i3d_refine_config (include/intrinsic3d_hip.h)."""
    _fields_ = [("num_grid_levels", C.c_int32), ("num_rgbd_levels", C.c_int32),
                ("thin_shell_factor", C.c_double), ("thin_shell_factor_final", C.c_double),
                ("clear_distant_voxels", C.c_int32), ("occlusion_distance", C.c_float), ("num_observations", C.c_int32),
                ("subvolume_size_sh", C.c_float), ("sh_lambda_reg", C.c_double)]


REFINE_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32)


class GridView(C.Structure):
    _fields_ = [("num_voxels", C.c_int64), ("voxel_size", C.c_float), ("truncation", C.c_float),
                ("keys", C.c_void_p), ("sdf", C.c_void_p), ("sdf_refined", C.c_void_p), ("albedo", C.c_void_p),
                ("weight", C.c_void_p), ("color", C.c_void_p)]


EXPORTS = ["i3d_create", "i3d_destroy", "i3d_last_error", "i3d_version", "i3d_set_grid", "i3d_get_grid", "i3d_update_grid",
           "i3d_set_frames", "i3d_set_frames_rgbd", "i3d_get_frame_image", "i3d_resize_depth", "i3d_set_camera", "i3d_get_camera", "i3d_set_voxel_sh", "i3d_get_voxel_sh",
           "i3d_optimizer_config_default", "i3d_optimize", "i3d_optimize_host", "i3d_estimate_sh",
           "i3d_set_grid_from_tsdf_records", "i3d_recompute_colors", "i3d_clear_outside_thin_shell", "i3d_upsample", "i3d_grid_info",
           "i3d_export_grid", "i3d_refine",
           "i3d_tsdf_read_header", "i3d_tsdf_read_records", "i3d_tsdf_write", "i3d_sbr_write", "i3d_sbr_read", "i3d_write_poses",
           "i3d_write_intrinsics", "i3d_read_intrinsics", "i3d_config_load_yaml", "i3d_yaml_get",
           "i3d_extract_mesh", "i3d_get_mesh", "i3d_export_mesh_ply", "i3d_write_ply", "i3d_mc_tables", "i3d_visualization_colors",
           "i3d_png_info", "i3d_png_decode", "i3d_pose_mat_to_vec6", "i3d_sensor_open", "i3d_sensor_open_yaml", "i3d_sensor_close", "i3d_sensor_info", "i3d_sensor_color",
           "i3d_sensor_depth", "i3d_sensor_pose", "i3d_sensor_set_pose", "i3d_sensor_set_pose_vec6", "i3d_sensor_save_poses",
           "i3d_mesh_remove_loose_components", "i3d_keyframes_load", "i3d_keyframes_save", "i3d_keyframes_select", "i3d_blur_score", "i3d_init_frames_from_sensor",
           "i3d_fusion_create", "i3d_fusion_destroy", "i3d_fusion_last_error", "i3d_fusion_integrate", "i3d_fusion_finish", "i3d_fusion_info", "i3d_fusion_get",
           "i3d_fusion_save", "i3d_shard_need", "i3d_comm_stats",
           "i3d_comm_unique_id", "i3d_comm_init", "i3d_comm_sim_create", "i3d_comm_sim_destroy", "i3d_comm_init_sim", "i3d_shard_plan", "i3d_shard_vec_index",
           "i3d_comm_transport", "i3d_timing_enable", "i3d_timing_select", "i3d_timing_get", "i3d_timing_get_work", "i3d_timing_get_work_ex", "i3d_kernel_name", "i3d_problem_sizes",
           "i3d_debug_assemble", "i3d_debug_map_order", "i3d_debug_flags", "i3d_debug_eg_rows", "i3d_debug_reg_rows", "i3d_debug_neighbors",
           "i3d_debug_normal_eq", "i3d_debug_jtj_apply", "i3d_debug_counters", "i3d_debug_cull_stats", "i3d_debug_ladder_stats"]

_lib = None


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C intrinsic3d_amd/csrc` "
                           "(or __graft_entry__.build()); the product path has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
    L.i3d_create.restype = i32; L.i3d_create.argtypes = [i32, C.POINTER(vp)]
    L.i3d_destroy.argtypes = [vp]
    L.i3d_last_error.restype = C.c_char_p; L.i3d_last_error.argtypes = [vp]
    L.i3d_version.restype = C.c_char_p
    L.i3d_set_grid.restype = i32; L.i3d_set_grid.argtypes = [vp, C.POINTER(GridView)]
    L.i3d_get_grid.restype = i32; L.i3d_get_grid.argtypes = [vp, vp, vp]
    L.i3d_update_grid.restype = i32; L.i3d_update_grid.argtypes = [vp, vp, vp, vp]
    L.i3d_set_frames.restype = i32; L.i3d_set_frames.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.i3d_set_frames_rgbd.restype = i32; L.i3d_set_frames_rgbd.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    L.i3d_resize_depth.restype = i32; L.i3d_resize_depth.argtypes = [i32, i32, i32, vp, vp, i32, i32, vp, vp]
    L.i3d_get_frame_image.restype = i32; L.i3d_get_frame_image.argtypes = [vp, i32, i32, vp, vp]
    L.i3d_set_camera.restype = i32; L.i3d_set_camera.argtypes = [vp, vp, vp, vp]
    L.i3d_get_camera.restype = i32; L.i3d_get_camera.argtypes = [vp, vp, vp, vp]
    L.i3d_set_voxel_sh.restype = i32; L.i3d_set_voxel_sh.argtypes = [vp, vp]
    L.i3d_get_voxel_sh.restype = i32; L.i3d_get_voxel_sh.argtypes = [vp, vp]
    L.i3d_optimizer_config_default.argtypes = [C.POINTER(OptimizerConfig)]
    L.i3d_optimize.restype = i32; L.i3d_optimize.argtypes = [vp, C.POINTER(OptimizerConfig), vp]
    L.i3d_optimize_host.restype = i32
    L.i3d_optimize_host.argtypes = [i32, C.POINTER(OptimizerConfig), C.POINTER(GridView), vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.i3d_estimate_sh.restype = i32
    L.i3d_estimate_sh.argtypes = [vp, f32, f64, f64, C.POINTER(i32), vp, vp, i32, C.POINTER(ShStats)]
    L.i3d_comm_unique_id.restype = i32; L.i3d_comm_unique_id.argtypes = [vp, C.POINTER(i32)]
    L.i3d_comm_init.restype = i32; L.i3d_comm_init.argtypes = [vp, i32, i32, vp, i32]
    L.i3d_comm_sim_create.restype = vp; L.i3d_comm_sim_create.argtypes = [i32]
    L.i3d_comm_sim_destroy.argtypes = [vp]
    L.i3d_comm_init_sim.restype = i32; L.i3d_comm_init_sim.argtypes = [vp, vp, i32]
    L.i3d_shard_plan.restype = i32; L.i3d_shard_plan.argtypes = [i32, i32, i32, vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), vp]
    L.i3d_shard_vec_index.restype = i32; L.i3d_shard_vec_index.argtypes = [i32, i32, i32]
    L.i3d_shard_need.restype = i32; L.i3d_shard_need.argtypes = [i32, i32, vp, vp, vp]
    L.i3d_comm_stats.restype = i32; L.i3d_comm_stats.argtypes = [vp] * 9
    L.i3d_timing_enable.restype = i32; L.i3d_timing_enable.argtypes = [vp, i32]
    L.i3d_timing_select.restype = i32; L.i3d_timing_select.argtypes = [vp, C.c_uint32]
    L.i3d_timing_get.restype = i32; L.i3d_timing_get.argtypes = [vp, vp, vp, i32]
    L.i3d_timing_get_work.restype = i32; L.i3d_timing_get_work.argtypes = [vp, vp, vp]
    L.i3d_timing_get_work_ex.restype = i32; L.i3d_timing_get_work_ex.argtypes = [vp, vp, vp, vp, vp]
    L.i3d_kernel_name.restype = C.c_char_p; L.i3d_kernel_name.argtypes = [i32]
    L.i3d_comm_transport.restype = C.c_char_p; L.i3d_comm_transport.argtypes = [vp]
    L.i3d_problem_sizes.restype = i32; L.i3d_problem_sizes.argtypes = [vp, vp]
    L.i3d_debug_assemble.restype = i32; L.i3d_debug_assemble.argtypes = [vp, C.POINTER(OptimizerConfig), i32, C.POINTER(i32)]
    L.i3d_debug_flags.restype = i32; L.i3d_debug_flags.argtypes = [vp, vp]
    L.i3d_debug_eg_rows.restype = i32; L.i3d_debug_eg_rows.argtypes = [vp, vp, vp, vp, vp]
    L.i3d_debug_reg_rows.restype = i32; L.i3d_debug_reg_rows.argtypes = [vp, vp, vp, vp]
    L.i3d_debug_neighbors.restype = i32; L.i3d_debug_neighbors.argtypes = [vp, vp]
    L.i3d_debug_normal_eq.restype = i32; L.i3d_debug_normal_eq.argtypes = [vp, vp, vp, C.POINTER(f64)]
    L.i3d_debug_jtj_apply.restype = i32; L.i3d_debug_jtj_apply.argtypes = [vp, vp, vp]
    L.i3d_debug_counters.restype = i32; L.i3d_debug_counters.argtypes = [vp, vp]
    L.i3d_debug_ladder_stats.restype = i32; L.i3d_debug_ladder_stats.argtypes = [vp, vp]
    L.i3d_debug_cull_stats.restype = i32; L.i3d_debug_cull_stats.argtypes = [vp, vp, vp]
    L.i3d_set_grid_from_tsdf_records.restype = i32; L.i3d_set_grid_from_tsdf_records.argtypes = [vp, f32, i64, vp, vp, vp, vp]
    L.i3d_recompute_colors.restype = i32; L.i3d_recompute_colors.argtypes = [vp, f32, i32]
    L.i3d_clear_outside_thin_shell.restype = i32; L.i3d_clear_outside_thin_shell.argtypes = [vp, f64, C.POINTER(i64)]
    L.i3d_upsample.restype = i32; L.i3d_upsample.argtypes = [vp, C.POINTER(i64)]
    L.i3d_grid_info.restype = i32; L.i3d_grid_info.argtypes = [vp, C.POINTER(i64), C.POINTER(f32), C.POINTER(f32)]
    L.i3d_export_grid.restype = i32; L.i3d_export_grid.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.i3d_refine.restype = i32; L.i3d_refine.argtypes = [vp, C.POINTER(RefineConfig), C.POINTER(OptimizerConfig), REFINE_CALLBACK, vp]
    u64 = C.c_uint64; cp = C.c_char_p
    L.i3d_tsdf_read_header.restype = i32; L.i3d_tsdf_read_header.argtypes = [cp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(u64), C.POINTER(f32)]
    L.i3d_tsdf_read_records.restype = i32; L.i3d_tsdf_read_records.argtypes = [cp, u64, vp, vp, vp, vp]
    L.i3d_tsdf_write.restype = i32; L.i3d_tsdf_write.argtypes = [cp, f32, f32, f32, f32, u64, vp, vp, vp, vp]
    L.i3d_sbr_write.restype = i32; L.i3d_sbr_write.argtypes = [cp, f32, f32, f32, f32, u64, vp, vp, vp, vp, vp, vp]
    L.i3d_sbr_read.restype = i32; L.i3d_sbr_read.argtypes = [cp, u64, vp, vp, vp, vp, vp, vp]
    L.i3d_write_poses.restype = i32; L.i3d_write_poses.argtypes = [cp, i32, vp, vp]
    L.i3d_write_intrinsics.restype = i32; L.i3d_write_intrinsics.argtypes = [cp, i32, i32, vp, vp]
    L.i3d_read_intrinsics.restype = i32; L.i3d_read_intrinsics.argtypes = [cp, C.POINTER(i32), C.POINTER(i32), vp, vp]
    L.i3d_extract_mesh.restype = i32; L.i3d_extract_mesh.argtypes = [vp, i32, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.i3d_get_mesh.restype = i32; L.i3d_get_mesh.argtypes = [vp, vp, vp, vp]
    L.i3d_export_mesh_ply.restype = i32; L.i3d_export_mesh_ply.argtypes = [vp, cp, i32, i32, i32]
    L.i3d_write_ply.restype = i32; L.i3d_write_ply.argtypes = [cp, i64, vp, vp, i64, vp]
    L.i3d_mesh_remove_loose_components.restype = i32; L.i3d_mesh_remove_loose_components.argtypes = [vp, vp, vp, vp, vp]
    L.i3d_mc_tables.restype = i32; L.i3d_mc_tables.argtypes = [vp, vp]
    L.i3d_config_load_yaml.restype = i32; L.i3d_config_load_yaml.argtypes = [cp, C.POINTER(RefineConfig), C.POINTER(OptimizerConfig)]
    u64 = C.c_uint64; f32 = C.c_float
    L.i3d_fusion_create.restype = i32; L.i3d_fusion_create.argtypes = [i32, f32, f32, f32, vp, u64, C.POINTER(vp)]
    L.i3d_fusion_destroy.restype = None; L.i3d_fusion_destroy.argtypes = [vp]
    L.i3d_fusion_last_error.restype = C.c_char_p; L.i3d_fusion_last_error.argtypes = [vp]
    L.i3d_fusion_integrate.restype = i32; L.i3d_fusion_integrate.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, i32]
    L.i3d_fusion_finish.restype = i32; L.i3d_fusion_finish.argtypes = [vp, i32, vp]
    L.i3d_fusion_info.restype = i32; L.i3d_fusion_info.argtypes = [vp, vp, vp, vp, vp]
    L.i3d_fusion_get.restype = i32; L.i3d_fusion_get.argtypes = [vp, vp, vp, vp, vp]
    L.i3d_fusion_save.restype = i32; L.i3d_fusion_save.argtypes = [vp, cp]
    L.i3d_debug_map_order.restype = i64; L.i3d_debug_map_order.argtypes = [vp, i64, i32, vp]
    L.i3d_blur_score.restype = i32; L.i3d_blur_score.argtypes = [vp, i32, i32, i32, vp]
    L.i3d_yaml_get.restype = i32; L.i3d_yaml_get.argtypes = [cp, cp, vp, u64]
    L.i3d_png_info.restype = i32; L.i3d_png_info.argtypes = [vp, u64, vp, vp, vp, vp]
    L.i3d_png_decode.restype = i32; L.i3d_png_decode.argtypes = [vp, u64, vp, u64]
    L.i3d_pose_mat_to_vec6.restype = i32; L.i3d_pose_mat_to_vec6.argtypes = [vp, vp]
    L.i3d_sensor_open.restype = i32; L.i3d_sensor_open.argtypes = [cp, i32, f32, f32, C.POINTER(vp)]
    L.i3d_sensor_close.restype = None; L.i3d_sensor_close.argtypes = [vp]
    L.i3d_sensor_info.restype = i32; L.i3d_sensor_info.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.i3d_sensor_color.restype = i32; L.i3d_sensor_color.argtypes = [vp, i32, vp]
    L.i3d_sensor_depth.restype = i32; L.i3d_sensor_depth.argtypes = [vp, i32, vp]
    L.i3d_sensor_pose.restype = i32; L.i3d_sensor_pose.argtypes = [vp, i32, vp]
    L.i3d_sensor_set_pose.restype = i32; L.i3d_sensor_set_pose.argtypes = [vp, i32, vp]
    L.i3d_sensor_set_pose_vec6.restype = i32; L.i3d_sensor_set_pose_vec6.argtypes = [vp, i32, vp]
    L.i3d_sensor_save_poses.restype = i32; L.i3d_sensor_save_poses.argtypes = [vp, cp]
    L.i3d_keyframes_load.restype = i32; L.i3d_keyframes_load.argtypes = [cp, vp, u64, vp, vp, vp]
    L.i3d_keyframes_save.restype = i32; L.i3d_keyframes_save.argtypes = [cp, i32, u64, vp, vp]
    L.i3d_keyframes_select.restype = i32; L.i3d_keyframes_select.argtypes = [i32, u64, vp, vp]
    L.i3d_init_frames_from_sensor.restype = i32; L.i3d_init_frames_from_sensor.argtypes = [vp, i32, vp, u64, vp, i32, i32, vp, vp]
    _lib = L
    return L


def default_config(**kw) -> OptimizerConfig:
    cfg = OptimizerConfig()
    load().i3d_optimizer_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class I3DError(RuntimeError):
    pass


class Context:
    """One device context (i3d_context)."""

    def __init__(self, device: int = 0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.i3d_create(int(device), C.byref(h))
        if rc != 0:
            raise I3DError(f"i3d_create failed ({rc}): {self.L.i3d_last_error(None).decode()}")
        self.h = h
        self.N = 0
        self.K = 0
        self._keep = []

    def _check(self, rc, what):
        if rc != 0:
            raise I3DError(f"{what} failed ({rc}): {self.L.i3d_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.i3d_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- uploads -------------------------------------------------------------------------------------------
    def set_grid(self, voxel_size, keys, sdf, sdf_refined, albedo, weight, color, truncation=None):
        keys = np.ascontiguousarray(keys, np.int32); sdf = np.ascontiguousarray(sdf, np.float64)
        sr = np.ascontiguousarray(sdf_refined, np.float64); al = np.ascontiguousarray(albedo, np.float64)
        w = np.ascontiguousarray(weight, np.float32); col = np.ascontiguousarray(color, np.uint8)
        gv = GridView(keys.shape[0], float(voxel_size), float(np.float32(voxel_size) * np.float32(5.0)) if truncation is None else float(truncation),
                      _p(keys), _p(sdf), _p(sr), _p(al), _p(w), _p(col))
        self._check(self.L.i3d_set_grid(self.h, C.byref(gv)), "i3d_set_grid")
        self.N = keys.shape[0]

    def set_frames(self, frames, levels):
        K = len(frames); self.K = K
        ws = np.array([frames[0]["lum"][l].shape[1] for l in range(levels)], np.int32)
        hs = np.array([frames[0]["lum"][l].shape[0] for l in range(levels)], np.int32)
        arr_t = C.c_void_p * (K * levels)
        lum = arr_t(); dep = arr_t(); bgr = arr_t(); keep = []
        for f in range(K):
            for l in range(levels):
                a = np.ascontiguousarray(frames[f]["lum"][l], np.float32); b = np.ascontiguousarray(frames[f]["depth"][l], np.float32)
                c = frames[f].get("bgr")
                c = np.ascontiguousarray(c[l], np.uint8) if c is not None else None
                keep += [a, b, c]
                lum[f * levels + l] = a.ctypes.data; dep[f * levels + l] = b.ctypes.data
                bgr[f * levels + l] = c.ctypes.data if c is not None else None
        self._check(self.L.i3d_set_frames(self.h, K, levels, _p(ws), _p(hs), C.cast(lum, C.c_void_p), C.cast(dep, C.c_void_p), C.cast(bgr, C.c_void_p)), "i3d_set_frames")

    def set_frames_rgbd(self, bgr_list, depth_list, levels):
        """level-0 colour (uint8 HxWx3, BGR) + depth (float32 HxW) per keyframe; the pyramids are built on the device"""
        K = len(bgr_list); h, w = depth_list[0].shape
        self._keep = [np.ascontiguousarray(b, np.uint8) for b in bgr_list] + [np.ascontiguousarray(d, np.float32) for d in depth_list]
        pb = (C.c_void_p * K)(*[a.ctypes.data for a in self._keep[:K]]); pd = (C.c_void_p * K)(*[a.ctypes.data for a in self._keep[K:]])
        self._check(self.L.i3d_set_frames_rgbd(self.h, K, int(levels), int(w), int(h), pb, pd), "i3d_set_frames_rgbd")
        self.K = K

    def get_frame_image(self, frame, level, w, h):
        lum = np.zeros((h, w), np.float32); dep = np.zeros((h, w), np.float32)
        self._check(self.L.i3d_get_frame_image(self.h, int(frame), int(level), _p(lum), _p(dep)), "i3d_get_frame_image")
        return lum, dep

    def set_camera(self, intr, dist, poses):
        a = np.ascontiguousarray(intr, np.float64); b = np.ascontiguousarray(dist, np.float64); c = np.ascontiguousarray(poses, np.float64)
        self._check(self.L.i3d_set_camera(self.h, _p(a), _p(b), _p(c)), "i3d_set_camera")

    def get_camera(self):
        a = np.zeros(4); b = np.zeros(5); c = np.zeros((self.K, 6))
        self._check(self.L.i3d_get_camera(self.h, _p(a), _p(b), _p(c)), "i3d_get_camera")
        return a, b, c

    def set_voxel_sh(self, sh):
        s = np.ascontiguousarray(sh, np.float64)
        self._check(self.L.i3d_set_voxel_sh(self.h, _p(s)), "i3d_set_voxel_sh")

    def get_voxel_sh(self):
        s = np.zeros((self.N, 9))
        self._check(self.L.i3d_get_voxel_sh(self.h, _p(s)), "i3d_get_voxel_sh")
        return s

    def get_grid(self):
        a = np.zeros(self.N); b = np.zeros(self.N)
        self._check(self.L.i3d_get_grid(self.h, _p(a), _p(b)), "i3d_get_grid")
        return a, b

    def update_grid(self, sdf_refined=None, albedo=None, color=None):
        a = None if sdf_refined is None else np.ascontiguousarray(sdf_refined, np.float64)
        b = None if albedo is None else np.ascontiguousarray(albedo, np.float64)
        c = None if color is None else np.ascontiguousarray(color, np.uint8)
        self._check(self.L.i3d_update_grid(self.h, _p(a), _p(b), _p(c)), "i3d_update_grid")

    # ---- level transitions / refine schedule ------------------------------------------------------------
    def set_grid_from_tsdf_records(self, voxel_size, keys, sdf, weight, color):
        keys = np.ascontiguousarray(keys, np.int32); sdf = np.ascontiguousarray(sdf, np.float32)
        weight = np.ascontiguousarray(weight, np.float32); color = np.ascontiguousarray(color, np.uint8)
        self._check(self.L.i3d_set_grid_from_tsdf_records(self.h, float(voxel_size), len(sdf), _p(keys), _p(sdf), _p(weight), _p(color)),
                    "i3d_set_grid_from_tsdf_records")
        self.grid_info()

    def grid_info(self):
        n = C.c_int64(0); vs = C.c_float(0); tr = C.c_float(0)
        self._check(self.L.i3d_grid_info(self.h, C.byref(n), C.byref(vs), C.byref(tr)), "i3d_grid_info")
        self.N = int(n.value)
        return self.N, float(vs.value), float(tr.value)

    def export_grid(self):
        N = self.grid_info()[0]
        out = dict(keys=np.zeros((N, 3), np.int32), sdf=np.zeros(N), sdf_refined=np.zeros(N), albedo=np.zeros(N),
                   weight=np.zeros(N, np.float32), color=np.zeros((N, 3), np.uint8))
        self._check(self.L.i3d_export_grid(self.h, _p(out["keys"]), _p(out["sdf"]), _p(out["sdf_refined"]), _p(out["albedo"]), _p(out["weight"]),
                                           _p(out["color"])), "i3d_export_grid")
        return out

    def recompute_colors(self, occlusion_distance, num_observations):
        self._check(self.L.i3d_recompute_colors(self.h, float(occlusion_distance), int(num_observations)), "i3d_recompute_colors")

    def clear_outside_thin_shell(self, thres_shell):
        n = C.c_int64(0)
        self._check(self.L.i3d_clear_outside_thin_shell(self.h, float(thres_shell), C.byref(n)), "i3d_clear_outside_thin_shell")
        self.N = int(n.value)
        return self.N

    def upsample(self):
        n = C.c_int64(0)
        self._check(self.L.i3d_upsample(self.h, C.byref(n)), "i3d_upsample")
        self.N = int(n.value)
        return self.N

    def refine(self, rcfg: "RefineConfig", ocfg: OptimizerConfig, callback=None):
        """Intrinsic3D::refine; callback(grid_level, num_grid_levels, pyramid_level, num_pyramid_levels) may call export_grid()."""
        cb = REFINE_CALLBACK((lambda user, a, b, c_, d: callback(a, b, c_, d)) if callback else (lambda *a: None))
        self._check(self.L.i3d_refine(self.h, C.byref(rcfg), C.byref(ocfg), cb, None), "i3d_refine")
        self.grid_info()

    # ---- mesh export ------------------------------------------------------------------------------------------
    def extract_mesh(self, use_refined_sdf=True, color_mode=0, largest_component_only=False):
        nv, nf = C.c_int64(0), C.c_int64(0)
        self._check(self.L.i3d_extract_mesh(self.h, int(bool(use_refined_sdf)), int(color_mode), int(bool(largest_component_only)), C.byref(nv), C.byref(nf)), "i3d_extract_mesh")
        v = np.zeros((nv.value, 3), np.float32); c = np.zeros((nv.value, 3), np.uint8); f = np.zeros((nf.value, 3), np.int32)
        self._check(self.L.i3d_get_mesh(self.h, _p(v), _p(c), _p(f)), "i3d_get_mesh")
        return v, c, f

    def export_mesh_ply(self, path, use_refined_sdf=True, color_mode=0, largest_component_only=False):
        self._check(self.L.i3d_export_mesh_ply(self.h, str(path).encode(), int(bool(use_refined_sdf)), int(color_mode), int(bool(largest_component_only))), "i3d_export_mesh_ply")

    # ---- sharding -----------------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(256); n = C.c_int32(0)
        rc = load().i3d_comm_unique_id(buf, C.byref(n))
        if rc != 0:
            raise I3DError(f"i3d_comm_unique_id failed ({rc})")
        return buf.raw[:n.value]

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        self._check(self.L.i3d_comm_init(self.h, int(rank), int(world), unique_id, len(unique_id)), "i3d_comm_init")

    def comm_stats(self):
        """traffic log of the sharded path: dict(halo_calls, halo_bytes_sent, reduce_calls, reduce_bytes, halo_send, halo_recv, ghost_tiles, compute_list)"""
        a = [C.c_int64() for _ in range(4)]; b = [C.c_int32() for _ in range(4)]
        self._check(self.L.i3d_comm_stats(self.h, *[C.byref(x) for x in a], *[C.byref(x) for x in b]), "i3d_comm_stats")
        return dict(zip(["halo_calls", "halo_bytes_sent", "reduce_calls", "reduce_bytes", "halo_send", "halo_recv", "ghost_tiles", "compute_list"], [x.value for x in a + b]))

    def comm_transport(self) -> str:
        return (self.L.i3d_comm_transport(self.h) or b"").decode()

    def comm_init_sim(self, shared, rank: int):
        self._check(self.L.i3d_comm_init_sim(self.h, shared, int(rank)), "i3d_comm_init_sim")

    # ---- the path ------------------------------------------------------------------------------------------
    def optimize(self, cfg: OptimizerConfig):
        stats = (IterationStats * cfg.iterations)()
        self._check(self.L.i3d_optimize(self.h, C.byref(cfg), C.cast(stats, C.c_void_p)), "i3d_optimize")
        return list(stats)

    def estimate_sh(self, subvolume_size, lambda_reg, thres_shell, cap=8192):
        S = C.c_int32(0); sh = np.zeros((cap, 9)); idx = np.zeros((cap, 3), np.int32); st = ShStats()
        self._check(self.L.i3d_estimate_sh(self.h, float(subvolume_size), float(lambda_reg), float(thres_shell), C.byref(S), _p(sh), _p(idx), cap, C.byref(st)), "i3d_estimate_sh")
        return sh[:S.value].copy(), idx[:S.value].copy(), st

    # ---- measurement ---------------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._check(self.L.i3d_timing_enable(self.h, 1 if on else 0), "i3d_timing_enable")

    def timing_select(self, names):
        """HIP events only around the launches of these categories (K_NAMES)"""
        mask = 0
        for n in names:
            mask |= 1 << K_NAMES.index(n)
        self._check(self.L.i3d_timing_select(self.h, mask), "i3d_timing_select")

    def timing_get_work(self):
        """like timing_get (no reset), restricted to launches that did work (>= 25 % of the category's longest launch)"""
        ms = np.zeros(len(K_NAMES)); n = np.zeros(len(K_NAMES), np.int64)
        self._check(self.L.i3d_timing_get_work(self.h, _p(ms), _p(n)), "i3d_timing_get_work")
        return {k: (ms[i], int(n[i])) for i, k in enumerate(K_NAMES)}

    def timing_get_work_ex(self):
        """timing_get_work plus the launches its upper cut-off (> 4x the 90th percentile) removed: {category: (ms, launches, slow_ms, slow_launches)}"""
        ms = np.zeros(len(K_NAMES)); n = np.zeros(len(K_NAMES), np.int64); sms = np.zeros(len(K_NAMES)); sn = np.zeros(len(K_NAMES), np.int64)
        self._check(self.L.i3d_timing_get_work_ex(self.h, _p(ms), _p(n), _p(sms), _p(sn)), "i3d_timing_get_work_ex")
        return {k: (ms[i], int(n[i]), sms[i], int(sn[i])) for i, k in enumerate(K_NAMES)}

    def timing_get(self, reset=True):
        ms = np.zeros(len(K_NAMES)); n = np.zeros(len(K_NAMES), np.int64)
        self._check(self.L.i3d_timing_get(self.h, _p(ms), _p(n), 1 if reset else 0), "i3d_timing_get")
        return {k: (ms[i], int(n[i])) for i, k in enumerate(K_NAMES)}

    def problem_sizes(self):
        o = np.zeros(6, np.int64)
        self._check(self.L.i3d_problem_sizes(self.h, _p(o)), "i3d_problem_sizes")
        return dict(zip(["active", "eg", "er", "es", "ea", "free"], [int(x) for x in o]))

    # ---- parity probes -------------------------------------------------------------------------------------
    def debug_assemble(self, cfg, iteration=0):
        s = C.c_int32(0)
        self._check(self.L.i3d_debug_assemble(self.h, C.byref(cfg), int(iteration), C.byref(s)), "i3d_debug_assemble")
        self.slots = s.value
        return s.value

    def debug_flags(self):
        f = np.zeros(self.N, np.uint8)
        self._check(self.L.i3d_debug_flags(self.h, _p(f)), "i3d_debug_flags")
        return f

    def debug_eg_rows(self, jac=True):
        S = self.slots
        fr = np.zeros((self.N, S), np.int32); w = np.zeros((self.N, S), np.float32); r = np.zeros((self.N, S), np.float32)
        J = np.zeros((self.N, S, 29), np.float32) if jac else None
        self._check(self.L.i3d_debug_eg_rows(self.h, _p(fr), _p(w), _p(r), _p(J)), "i3d_debug_eg_rows")
        return fr, w, r, J

    def debug_reg_rows(self):
        er = np.zeros(self.N, np.uint8); es = np.zeros(self.N, np.uint8); ea = np.zeros((self.N, 6), np.float32)
        self._check(self.L.i3d_debug_reg_rows(self.h, _p(er), _p(es), _p(ea)), "i3d_debug_reg_rows")
        return er, es, ea

    def debug_neighbors(self):
        nb = np.zeros((self.N, 18), np.int32)
        self._check(self.L.i3d_debug_neighbors(self.h, _p(nb)), "i3d_debug_neighbors")
        return nb

    def debug_normal_eq(self):
        NP = 2 * self.N + 6 * self.K + 9
        g = np.zeros(NP); d = np.zeros(NP); cost = C.c_double(0)
        self._check(self.L.i3d_debug_normal_eq(self.h, _p(g), _p(d), C.byref(cost)), "i3d_debug_normal_eq")
        return g, d, cost.value

    def debug_counters(self):
        """stream synchronisations of the solver path since the context was created"""
        n = C.c_int64(0)
        self._check(self.L.i3d_debug_counters(self.h, C.byref(n)), "i3d_debug_counters")
        return {"stream_syncs": int(n.value)}

    def debug_ladder_stats(self):
        """the damping ladder since the context was created: batches, row streams, system passes (= the streams of the serial loop), re-solved batches, unused systems, depth"""
        a = (C.c_int64 * 6)()
        self._check(self.L.i3d_debug_ladder_stats(self.h, a), "i3d_debug_ladder_stats")
        return {"batches": int(a[0]), "row_streams": int(a[1]), "system_passes": int(a[2]), "resyncs": int(a[3]), "unused_systems": int(a[4]), "depth": int(a[5])}

    def debug_cull_stats(self):
        """(group, keyframe) pairs of the last assemble and how many the observation pass skipped (-1: culling off)."""
        a = C.c_int64(0); b = C.c_int64(0)
        self._check(self.L.i3d_debug_cull_stats(self.h, C.byref(a), C.byref(b)), "i3d_debug_cull_stats")
        return int(a.value), int(b.value)

    def debug_jtj_apply(self, x):
        x = np.ascontiguousarray(x, np.float64); y = np.zeros_like(x)
        self._check(self.L.i3d_debug_jtj_apply(self.h, _p(x), _p(y)), "i3d_debug_jtj_apply")
        return y


def shard_need(A, world, anbr, active):
    """need[e] bit k: rank k's rows read entry e, which it does not own (host statement of the device plan)."""
    anbr = np.ascontiguousarray(anbr, np.int32); active = np.ascontiguousarray(active, np.uint8); need = np.zeros(A, np.uint64)
    rc = load().i3d_shard_need(int(A), int(world), _p(anbr), _p(active), _p(need))
    if rc != 0:
        raise I3DError(f"i3d_shard_need failed ({rc})")
    return need


def shard_plan(A, world, rank, anbr, active):
    """Host-side sharding plan (no GPU): returns chunk, own0, own1 and the compute-list mask of `rank`."""
    anbr = np.ascontiguousarray(anbr, np.int32); active = np.ascontiguousarray(active, np.uint8)
    ch = C.c_int32(); o0 = C.c_int32(); o1 = C.c_int32(); mask = np.zeros(A, np.uint8)
    rc = load().i3d_shard_plan(int(A), int(world), int(rank), _p(anbr), _p(active), C.byref(ch), C.byref(o0), C.byref(o1), _p(mask))
    if rc != 0:
        raise I3DError(f"i3d_shard_plan failed ({rc})")
    return ch.value, o0.value, o1.value, mask.astype(bool)


# ---- on-disk formats (host-only entry points) -----------------------------------------------------------------------------
def _io_check(rc, what):
    if rc != 0:
        raise I3DError(f"{what} failed ({rc})")


def tsdf_read(path):
    """-> dict(voxel_size, truncation, integration_weight_sample, max_load_factor, keys, sdf, weight, color) in FILE order."""
    L = load(); p = str(path).encode()
    vs, tr, iw, ml = C.c_float(), C.c_float(), C.c_float(), C.c_float(); n = C.c_uint64()
    _io_check(L.i3d_tsdf_read_header(p, C.byref(vs), C.byref(tr), C.byref(iw), C.byref(n), C.byref(ml)), "i3d_tsdf_read_header")
    N = int(n.value)
    out = dict(voxel_size=np.float32(vs.value), truncation=np.float32(tr.value), integration_weight_sample=np.float32(iw.value),
               max_load_factor=np.float32(ml.value), keys=np.zeros((N, 3), np.int32), sdf=np.zeros(N, np.float32),
               weight=np.zeros(N, np.float32), color=np.zeros((N, 3), np.uint8))
    _io_check(L.i3d_tsdf_read_records(p, N, _p(out["keys"]), _p(out["sdf"]), _p(out["weight"]), _p(out["color"])), "i3d_tsdf_read_records")
    return out


def tsdf_write(path, voxel_size, keys, sdf, weight, color, truncation=None, integration_weight_sample=0.0, max_load_factor=0.6):
    keys = np.ascontiguousarray(keys, np.int32); sdf = np.ascontiguousarray(sdf, np.float32)
    weight = np.ascontiguousarray(weight, np.float32); color = np.ascontiguousarray(color, np.uint8)
    tr = float(np.float32(voxel_size) * np.float32(5.0)) if truncation is None else float(truncation)
    _io_check(load().i3d_tsdf_write(str(path).encode(), float(voxel_size), tr, float(integration_weight_sample), float(max_load_factor), len(sdf),
                                    _p(keys), _p(sdf), _p(weight), _p(color)), "i3d_tsdf_write")


def sbr_write(path, voxel_size, grid, truncation=None, integration_weight_sample=0.0, max_load_factor=0.6):
    """grid: the dict of Context.export_grid() (visit order)."""
    tr = float(np.float32(voxel_size) * np.float32(5.0)) if truncation is None else float(truncation)
    g = {k: np.ascontiguousarray(v) for k, v in grid.items() if isinstance(v, np.ndarray)}
    _io_check(load().i3d_sbr_write(str(path).encode(), float(voxel_size), tr, float(integration_weight_sample), float(max_load_factor), len(g["sdf"]),
                                   _p(g["keys"]), _p(g["sdf"]), _p(g["sdf_refined"]), _p(g["albedo"]), _p(g["weight"]), _p(g["color"])), "i3d_sbr_write")


def sbr_read(path):
    L = load(); p = str(path).encode()
    vs = C.c_float(); n = C.c_uint64()
    _io_check(L.i3d_tsdf_read_header(p, C.byref(vs), None, None, C.byref(n), None), "i3d_tsdf_read_header")
    N = int(n.value)
    out = dict(voxel_size=np.float32(vs.value), keys=np.zeros((N, 3), np.int32), sdf=np.zeros(N), sdf_refined=np.zeros(N), albedo=np.zeros(N),
               weight=np.zeros(N, np.float32), color=np.zeros((N, 3), np.uint8))
    _io_check(L.i3d_sbr_read(p, N, _p(out["keys"]), _p(out["sdf"]), _p(out["sdf_refined"]), _p(out["albedo"]), _p(out["weight"]), _p(out["color"])), "i3d_sbr_read")
    return out


def write_poses(path, timestamps, poses_world_to_cam):
    t = np.ascontiguousarray(timestamps, np.float64); p = np.ascontiguousarray(poses_world_to_cam, np.float64).reshape(-1, 6)
    _io_check(load().i3d_write_poses(str(path).encode(), len(t), _p(t), _p(p)), "i3d_write_poses")


def write_intrinsics(path, width, height, intr, dist):
    a = np.ascontiguousarray(intr, np.float64); d = np.ascontiguousarray(dist, np.float64)
    _io_check(load().i3d_write_intrinsics(str(path).encode(), int(width), int(height), _p(a), _p(d)), "i3d_write_intrinsics")


def read_intrinsics(path):
    w, h = C.c_int32(), C.c_int32(); a = np.zeros(4); d = np.zeros(5)
    rc = load().i3d_read_intrinsics(str(path).encode(), C.byref(w), C.byref(h), _p(a), _p(d))
    return rc == 0, int(w.value), int(h.value), a, d


def load_yaml_config(path):
    rc = RefineConfig(); oc = default_config()
    _io_check(load().i3d_config_load_yaml(str(path).encode(), C.byref(rc), C.byref(oc)), "i3d_config_load_yaml")
    return rc, oc


def yaml_get(path, key, default=None):
    buf = C.create_string_buffer(4096)
    rc = load().i3d_yaml_get(str(path).encode(), key.encode(), buf, 4096)
    if rc == 1 and default is not None:
        return default
    _io_check(rc, f"i3d_yaml_get({key})")
    return buf.value.decode()


def mesh_remove_loose_components(vertices, colors, faces):
    """MeshUtil::removeLooseComponents on arrays -> (vertices, colors or None, faces) of the largest connected component"""
    v = np.array(vertices, np.float32, copy=True).reshape(-1, 3); f = np.array(faces, np.int32, copy=True).reshape(-1, 3)
    c = None if colors is None else np.array(colors, np.uint8, copy=True).reshape(-1, 3)
    nv, nf = C.c_int64(len(v)), C.c_int64(len(f))
    _io_check(load().i3d_mesh_remove_loose_components(C.byref(nv), _p(v), None if c is None else _p(c), C.byref(nf), _p(f)), "i3d_mesh_remove_loose_components")
    return v[:nv.value].copy(), (None if c is None else c[:nv.value].copy()), f[:nf.value].copy()


COLOR_MODES = {"": 0, "albedo": 1, "normals": 2, "lap": 3, "lum": 4, "lum_grad": 5, "shading_sv": 6, "shading_sv_const": 7, "chroma": 8}   # SDFVisualization::getOutputModes' names


def visualization_colors(mode, voxel_size, keys, sdf_refined, albedo, weight, color, subvolume_size=0.0, sub_index=None, sub_sh=None, visit_rank=None):
    """SDFVisualization::applyColor<mode> on arrays (host instantiation of the export kernel's function) -> colours [n, 3]"""
    k = np.ascontiguousarray(keys, np.int32); n = len(k); out = np.zeros((n, 3), np.uint8)
    s = np.ascontiguousarray(sdf_refined, np.float64); a = np.ascontiguousarray(albedo, np.float64); w = np.ascontiguousarray(weight, np.float32); c = np.ascontiguousarray(color, np.uint8)
    si = None if sub_index is None else np.ascontiguousarray(sub_index, np.int32); ss = None if sub_sh is None else np.ascontiguousarray(sub_sh, np.float64)
    L = load(); L.i3d_visualization_colors.restype = C.c_int32
    L.i3d_visualization_colors.argtypes = [C.c_int32, C.c_float, C.c_int64] + [C.c_void_p] * 6 + [C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    vr = None if visit_rank is None else np.ascontiguousarray(visit_rank, np.int64)
    _io_check(L.i3d_visualization_colors(COLOR_MODES[mode] if isinstance(mode, str) else int(mode), float(voxel_size), n, _p(k), _p(s), _p(a), _p(w), _p(c), _p(vr), float(subvolume_size),
                                         0 if si is None else len(si), _p(si), _p(ss), _p(out)), "i3d_visualization_colors")
    return out


def write_ply(path, vertices, colors, faces):
    v = np.ascontiguousarray(vertices, np.float32); f = np.ascontiguousarray(faces, np.int32)
    c = None if colors is None else np.ascontiguousarray(colors, np.uint8)
    _io_check(load().i3d_write_ply(str(path).encode(), len(v), _p(v), _p(c), len(f), _p(f)), "i3d_write_ply")


def mc_tables():
    ntri = np.zeros(256, np.uint8); tri = np.zeros((256, 16), np.int8)
    mx = load().i3d_mc_tables(_p(ntri), _p(tri))
    return ntri, tri, mx


def optimize_host(cfg, voxel_size, keys, sdf, sdf_refined, albedo, weight, color, frames, levels, intr, dist, poses, voxel_sh, device=0):
    """i3d_optimize_host: the one-call form of Optimizer::optimize (host arrays in, unknowns written back in place).  Returns
    (sdf_refined, albedo, intr, dist, poses, stats)."""
    L = load()
    keys = np.ascontiguousarray(keys, np.int32); sdf = np.ascontiguousarray(sdf, np.float64)
    sr = np.array(sdf_refined, np.float64); al = np.array(albedo, np.float64)
    w = np.ascontiguousarray(weight, np.float32); col = np.ascontiguousarray(color, np.uint8)
    gv = GridView(keys.shape[0], float(voxel_size), float(np.float32(voxel_size) * np.float32(5.0)), _p(keys), _p(sdf), _p(sr), _p(al), _p(w), _p(col))
    K = len(frames)
    ws = np.array([frames[0]["lum"][l].shape[1] for l in range(levels)], np.int32); hs = np.array([frames[0]["lum"][l].shape[0] for l in range(levels)], np.int32)
    arr_t = C.c_void_p * (K * levels); lum = arr_t(); dep = arr_t(); keep = []
    for f in range(K):
        for l in range(levels):
            a = np.ascontiguousarray(frames[f]["lum"][l], np.float32); b = np.ascontiguousarray(frames[f]["depth"][l], np.float32); keep += [a, b]
            lum[f * levels + l] = a.ctypes.data; dep[f * levels + l] = b.ctypes.data
    i4 = np.array(intr, np.float64); d5 = np.array(dist, np.float64); p6 = np.array(poses, np.float64); sh = np.ascontiguousarray(voxel_sh, np.float64)
    stats = (IterationStats * cfg.iterations)()
    rc = L.i3d_optimize_host(int(device), C.byref(cfg), C.byref(gv), _p(sr), _p(al), K, int(levels), _p(ws), _p(hs), C.cast(lum, C.c_void_p), C.cast(dep, C.c_void_p),
                             _p(i4), _p(d5), _p(p6), _p(sh), C.cast(stats, C.c_void_p))
    if rc != 0:
        raise I3DError(f"i3d_optimize_host failed ({rc})")
    return sr, al, i4, d5, p6, list(stats)


def resize_depth(depth, in_intr, out_w, out_h, out_intr, device=0):
    d = np.ascontiguousarray(depth, np.float32); a = np.ascontiguousarray(in_intr, np.float32); b = np.ascontiguousarray(out_intr, np.float32)
    out = np.zeros((out_h, out_w), np.float32)
    _io_check(load().i3d_resize_depth(int(device), d.shape[1], d.shape[0], _p(d), _p(a), int(out_w), int(out_h), _p(b), _p(out)), "i3d_resize_depth")
    return out


def png_decode(data: bytes):
    """cv::imdecode(buf, IMREAD_UNCHANGED) for PNG: HxW or HxWxC array (uint8 / uint16), colour in B,G,R[,A] order"""
    L = load(); buf = np.frombuffer(data, np.uint8)
    w = C.c_int32(); h = C.c_int32(); ch = C.c_int32(); bd = C.c_int32()
    _io_check(L.i3d_png_info(_p(buf), buf.size, C.byref(w), C.byref(h), C.byref(ch), C.byref(bd)), "i3d_png_info")
    out = np.zeros((h.value, w.value, ch.value), np.uint16 if bd.value == 16 else np.uint8)
    _io_check(L.i3d_png_decode(_p(buf), buf.size, _p(out), out.nbytes), "i3d_png_decode")
    return out[:, :, 0] if ch.value == 1 else out


def pose_mat_to_vec6(cam_to_world):
    m = np.ascontiguousarray(cam_to_world, np.float32).reshape(16); out = np.zeros(6)
    _io_check(load().i3d_pose_mat_to_vec6(_p(m), _p(out)), "i3d_pose_mat_to_vec6")
    return out


def keyframes_load(path):
    """KeyframeSelection::load -> (window_size, scores, is_keyframe)"""
    L = load(); win = C.c_int32(0); n = C.c_uint64(0)
    _io_check(L.i3d_keyframes_load(path.encode(), C.byref(win), 0, None, None, C.byref(n)), "i3d_keyframes_load")
    scores = np.zeros(n.value); kf = np.zeros(n.value, np.uint8)
    _io_check(L.i3d_keyframes_load(path.encode(), C.byref(win), n.value, _p(scores), _p(kf), C.byref(n)), "i3d_keyframes_load")
    return win.value, scores, kf.astype(bool)


def keyframes_save(path, window_size, scores, is_keyframe):
    s = np.ascontiguousarray(scores, np.float64); k = np.ascontiguousarray(is_keyframe, np.uint8)
    _io_check(load().i3d_keyframes_save(path.encode(), int(window_size), s.size, _p(s), _p(k)), "i3d_keyframes_save")


def keyframes_select(window_size, scores):
    s = np.ascontiguousarray(scores, np.float64); k = np.zeros(s.size, np.uint8)
    _io_check(load().i3d_keyframes_select(int(window_size), s.size, _p(s), _p(k)), "i3d_keyframes_select")
    return k.astype(bool)


class Sensor:
    """Sensor::create on an Intrinsic3D dataset folder (rgbd/sensor_i3d.cpp); decoding happens on demand, like the reference"""

    def __init__(self, folder=None, max_frames=0, min_depth=0.0, max_depth=0.0, yml=None):
        self.L = load(); self.h = C.c_void_p(); self.depth_range = (float(min_depth), float(max_depth))
        if yml is not None:                                           # Sensor::create(Settings(sensor.yml))
            lo = C.c_float(); hi = C.c_float()
            self.L.i3d_sensor_open_yaml.restype = C.c_int32; self.L.i3d_sensor_open_yaml.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.POINTER(C.c_float)]
            _io_check(self.L.i3d_sensor_open_yaml(str(yml).encode(), C.byref(self.h), C.byref(lo), C.byref(hi)), "i3d_sensor_open_yaml")
            self.depth_range = (lo.value, hi.value)
        else:
            _io_check(self.L.i3d_sensor_open(str(folder).encode(), int(max_frames), float(min_depth), float(max_depth), C.byref(self.h)), "i3d_sensor_open")
        nf = C.c_int32(); nl = C.c_int32(); cwh = np.zeros(2, np.int32); dwh = np.zeros(2, np.int32); ci = np.zeros(4, np.float32); di = np.zeros(4, np.float32)
        _io_check(self.L.i3d_sensor_info(self.h, C.byref(nf), C.byref(nl), _p(cwh), _p(dwh), _p(ci), _p(di)), "i3d_sensor_info")
        self.num_frames, self.num_loaded = nf.value, nl.value
        self.color_size, self.depth_size, self.color_intrinsics, self.depth_intrinsics = tuple(cwh), tuple(dwh), ci, di

    def close(self):
        if self.h:
            self.L.i3d_sensor_close(self.h); self.h = None

    def __del__(self):
        self.close()

    def color(self, i):
        out = np.zeros((self.color_size[1], self.color_size[0], 3), np.uint8)
        _io_check(self.L.i3d_sensor_color(self.h, int(i), _p(out)), "i3d_sensor_color"); return out

    def depth(self, i):
        out = np.zeros((self.depth_size[1], self.depth_size[0]), np.float32)
        _io_check(self.L.i3d_sensor_depth(self.h, int(i), _p(out)), "i3d_sensor_depth"); return out

    def pose(self, i):
        out = np.zeros((4, 4), np.float32)
        _io_check(self.L.i3d_sensor_pose(self.h, int(i), _p(out)), "i3d_sensor_pose"); return out

    def set_pose(self, i, cam_to_world):
        m = np.ascontiguousarray(cam_to_world, np.float32)
        _io_check(self.L.i3d_sensor_set_pose(self.h, int(i), _p(m)), "i3d_sensor_set_pose")

    def set_pose_vec6(self, i, pose_world_to_cam):
        p = np.ascontiguousarray(pose_world_to_cam, np.float64)
        _io_check(self.L.i3d_sensor_set_pose_vec6(self.h, int(i), _p(p)), "i3d_sensor_set_pose_vec6")

    def save_poses(self, path):
        _io_check(self.L.i3d_sensor_save_poses(self.h, str(path).encode()), "i3d_sensor_save_poses")


def init_frames_from_sensor(ctx: "Context", sensor: Sensor, is_keyframe, levels, device=0):
    """Intrinsic3D::init's keyframe loop; returns the frame ids of the keyframes (ImageFormationModel::frame_ids)"""
    kf = np.ascontiguousarray(is_keyframe, np.uint8); ids = np.zeros(max(1, int(kf.sum())), np.int32); nk = C.c_int32(0)
    rc = ctx.L.i3d_init_frames_from_sensor(ctx.h, int(device), sensor.h, kf.size, _p(kf), int(levels), ids.size, _p(ids), C.byref(nk))
    ctx._check(rc, "i3d_init_frames_from_sensor")
    ctx.K = nk.value
    return ids[:nk.value]


class Fusion:
    """AppFusion::fuseSDF's volume on the device: integrate() per frame, then finish() = correctSDF + clearInvalidVoxels"""

    def __init__(self, voxel_size, depth_min, depth_max, clip=None, initial_capacity=1 << 20, device=0):
        self.L = load(); self.h = C.c_void_p()
        c = None if clip is None else np.ascontiguousarray(clip, np.float32)
        rc = self.L.i3d_fusion_create(int(device), float(voxel_size), float(depth_min), float(depth_max), _p(c) if c is not None else None, int(initial_capacity), C.byref(self.h))
        if rc != 0:
            raise I3DError(f"i3d_fusion_create failed ({rc})")

    def _check(self, rc, what):
        if rc != 0:
            raise I3DError(f"{what} failed ({rc}): {self.L.i3d_fusion_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.i3d_fusion_destroy(self.h); self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def integrate(self, depth, dcam, bgr, ccam, pose_c2w, erode_window=2):
        d = np.ascontiguousarray(depth, np.float32); b = np.ascontiguousarray(bgr, np.uint8)
        dc = np.ascontiguousarray(dcam, np.float32); cc = np.ascontiguousarray(ccam, np.float32); T = np.ascontiguousarray(pose_c2w, np.float32)
        self._check(self.L.i3d_fusion_integrate(self.h, d.shape[1], d.shape[0], _p(dc), b.shape[1], b.shape[0], _p(cc), _p(d), _p(b), _p(T), int(erode_window)),
                    "i3d_fusion_integrate")

    def finish(self, correct_iterations=10):
        n = C.c_uint64(0)
        self._check(self.L.i3d_fusion_finish(self.h, int(correct_iterations), C.byref(n)), "i3d_fusion_finish")
        return n.value

    def info(self):
        fr = C.c_uint64(); al = C.c_uint64(); cap = C.c_uint64(); cl = C.c_int32()
        self._check(self.L.i3d_fusion_info(self.h, C.byref(fr), C.byref(al), C.byref(cap), C.byref(cl)), "i3d_fusion_info")
        return dict(frames=fr.value, allocated=al.value, capacity=cap.value, correct_launches=cl.value)

    def export(self):
        n = self.finish()
        keys = np.zeros((n, 3), np.int32); sdf = np.zeros(n, np.float32); w = np.zeros(n, np.float32); col = np.zeros((n, 3), np.uint8)
        self._check(self.L.i3d_fusion_get(self.h, _p(keys), _p(sdf), _p(w), _p(col)), "i3d_fusion_get")
        return dict(keys=keys, sdf=sdf, weight=w, color=col)

    def save(self, path):
        self._check(self.L.i3d_fusion_save(self.h, str(path).encode()), "i3d_fusion_save")


def debug_map_order(keys, mode=0):
    k = np.ascontiguousarray(keys, np.int32); out = np.zeros(k.shape[0], np.int32)
    n = load().i3d_debug_map_order(_p(k), k.shape[0], int(mode), _p(out))
    return out[:n]


def blur_score(image):
    """KeyframeSelection::estimateBlur of an HxW (grey) or HxWx3 (B,G,R) uint8 image"""
    a = np.ascontiguousarray(image, np.uint8); out = C.c_double(0)
    _io_check(load().i3d_blur_score(_p(a), a.shape[1], a.shape[0], 1 if a.ndim == 2 else a.shape[2], C.byref(out)), "i3d_blur_score")
    return out.value
