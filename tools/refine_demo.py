#!/usr/bin/env python3
"""End-to-end run of the resident path through the C ABI, shaped like AppIntrinsic3D (apps/src/app_intrinsic3d.cpp:96-210):

    .tsdf volume + keyframes (synthetic here) -> i3d_refine (coarse-to-fine) -> per level: mesh_g{L}_p{P}[_albedo].ply, poses_*.txt, intrinsics_*.txt

    python tools/refine_demo.py --out gpurun_out/demo --radius 100 --frames 30 --grid-levels 3 --rgbd-levels 2
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from intrinsic3d_amd import binding, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/demo"); ap.add_argument("--radius", type=int, default=60)
    ap.add_argument("--frames", type=int, default=20); ap.add_argument("--width", type=int, default=320); ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--grid-levels", type=int, default=3); ap.add_argument("--rgbd-levels", type=int, default=2); ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--voxel-size", type=float, default=0.004)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    t0 = time.time()
    sc = synthetic.make_scene(radius_vox=a.radius, voxel_size=a.voxel_size, K=a.frames, width=a.width, height=a.height, levels=1, seed=3,
                              pose_noise=(0.001, 0.002), lum_noise=0.003)
    tsdf = os.path.join(a.out, f"volume_{a.voxel_size:g}.tsdf")
    binding.tsdf_write(tsdf, sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])          # what AppFusion would have left
    print(f"[demo] scene {sc['keys'].shape[0]} voxels, {a.frames} keyframes in {time.time() - t0:.1f}s -> {tsdf}")

    vol = binding.tsdf_read(tsdf)
    ctx = binding.Context(0)
    ctx.set_grid_from_tsdf_records(vol["voxel_size"], vol["keys"], vol["sdf"], vol["weight"], vol["color"])
    ctx.set_frames_rgbd([f["bgr"][0] for f in sc["frames"]], [f["depth"][0] for f in sc["frames"]], a.rgbd_levels)
    ctx.set_camera(sc["intr"], np.zeros(5), sc["poses"])
    rc = binding.RefineConfig(num_grid_levels=a.grid_levels, num_rgbd_levels=a.rgbd_levels, thin_shell_factor=2.0, thin_shell_factor_final=1.0,
                              clear_distant_voxels=1, occlusion_distance=0.02, num_observations=5, subvolume_size_sh=0.2, sh_lambda_reg=10.0)
    oc = binding.default_config(iterations=a.iterations, lm_steps=50, lambda_g=0.2, lambda_r0=80.0, lambda_r1=10.0, lambda_s0=120.0, lambda_s1=10.0, lambda_a=0.1)
    stamps = np.arange(a.frames, dtype=np.float64)

    def on_refined(gl, ngl, pl, npl):                                  # AppIntrinsic3D::onSDFRefined
        post = f"_g{gl}_p{pl}"
        n, vs, _ = ctx.grid_info()
        ctx.export_mesh_ply(os.path.join(a.out, "mesh" + post + ".ply"), True, 0, True)
        ctx.export_mesh_ply(os.path.join(a.out, "mesh" + post + "_albedo.ply"), True, 1, True)
        intr, dist, poses = ctx.get_camera()
        binding.write_poses(os.path.join(a.out, "poses" + post + ".txt"), stamps, poses)
        binding.write_intrinsics(os.path.join(a.out, "intrinsics" + post + ".txt"), a.width, a.height, intr, dist)
        print(f"[demo] level g{gl} p{pl}: {n} voxels @ {vs * 1e3:.2f} mm, fx={intr[0]:.3f} ({time.time() - t1:.1f}s since start of refine)", flush=True)

    t1 = time.time()
    ctx.refine(rc, oc, on_refined)
    n, vs, _ = ctx.grid_info()
    g = ctx.export_grid()
    binding.sbr_write(os.path.join(a.out, "refined.sbr"), vs, g)
    print(f"[demo] refine done in {time.time() - t1:.1f}s: {n} voxels @ {vs * 1e3:.2f} mm; albedo mean {g['albedo'][g['weight'] > 0].mean():.4f}; "
          f"mean |sdf_refined - sdf| = {np.abs(g['sdf_refined'] - g['sdf'])[g['weight'] > 0].mean() / vs:.4f} voxels")
    ctx.close()


if __name__ == "__main__":
    main()
