#!/usr/bin/env python3
"""Write a synthetic scene in the reference's on-disk dataset layout, ready for apps/app_intrinsic3d:

    <out>/sensor.yml  <out>/intrinsic3d.yml  <out>/fusion.yml  <out>/keyframes.yml
    <out>/rgbd/frame-%06d.{color,depth}.png  .pose.txt  colorIntrinsics.txt  depthIntrinsics.txt      (rgbd/sensor_i3d.cpp:184-220)
    <out>/fusion/keyframes.txt  <out>/fusion/volume_<voxel size>.tsdf                                  (what AppKeyframes / AppFusion leave)

    python tools/make_dataset.py --out /tmp/ds --radius 40 --frames 12
    apps/app_keyframes -s /tmp/ds/sensor.yml -k /tmp/ds/keyframes.yml      # optional: re-selects the keyframes by blur score (host only)
    apps/app_fusion -s /tmp/ds/sensor.yml -f /tmp/ds/fusion.yml            # optional: replaces the analytic volume by one fused from the frames
    apps/app_intrinsic3d -s /tmp/ds/sensor.yml -i /tmp/ds/intrinsic3d.yml
"""
import argparse, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from intrinsic3d_amd import binding, synthetic


def pose_vec_to_cam_to_world(p):
    th = np.linalg.norm(p[:3]); k = p[:3] / th if th > 0 else np.zeros(3)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    T = np.eye(4); T[:3, :3] = R.T; T[:3, 3] = -R.T @ p[3:]
    return T


def write_dataset(out, sc, window=1, grid_levels=2, rgbd_levels=2, iterations=2, extra_frames=0, **cfg):
    """sc: synthetic.make_scene(..., levels=1).  Every `window`-th frame window holds one keyframe; extra_frames appends non-keyframes."""
    os.makedirs(os.path.join(out, "rgbd"), exist_ok=True); os.makedirs(os.path.join(out, "fusion"), exist_ok=True)
    h, w = sc["frames"][0]["depth"][0].shape
    K = np.eye(4); K[0, 0], K[1, 1], K[0, 2], K[1, 2] = sc["intr"]
    for name in ("colorIntrinsics.txt", "depthIntrinsics.txt"):
        np.savetxt(os.path.join(out, "rgbd", name), K, fmt="%.9g")
    n = len(sc["frames"]) + extra_frames
    for i in range(n):
        fr = sc["frames"][min(i, len(sc["frames"]) - 1)]; pose = sc["poses"][min(i, len(sc["frames"]) - 1)]
        base = os.path.join(out, "rgbd", f"frame-{i:06d}")
        Image.fromarray(np.ascontiguousarray(fr["bgr"][0][:, :, ::-1])).save(base + ".color.png")
        Image.fromarray(np.clip(np.rint(fr["depth"][0] * 1000.0), 0, 65535).astype(np.uint16)).save(base + ".depth.png")
        np.savetxt(base + ".pose.txt", pose_vec_to_cam_to_world(np.asarray(pose, np.float64)), fmt="%.9g")
    scores = np.linspace(0.3, 0.6, n); is_kf = np.zeros(n, bool); is_kf[:len(sc["frames"])] = True
    binding.keyframes_save(os.path.join(out, "fusion", "keyframes.txt"), window, scores, is_kf)
    tsdf = f"./fusion/volume_{float(sc['voxel_size']):g}.tsdf"
    binding.tsdf_write(os.path.join(out, tsdf), sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    with open(os.path.join(out, "sensor.yml"), "w") as f:
        f.write('%YAML:1.0\n\n# rgbd sensor config\ndataset: "./rgbd/"\nmax_frames: "0"\nmin_depth: "0.1"\nmax_depth: "10.0"\n')
    vals = dict(keyframes="./fusion/keyframes.txt", input_sdf=tsdf, num_grid_levels=grid_levels, num_rgbd_levels=rgbd_levels, thin_shell_factor=2.0,
                thin_shell_factor_final=1.0, subvolume_size_sh=0.2, subvolume_sh_lamda_reg=10.0, clear_distant_voxels=1, occlusion_distance=0.02,
                num_observations=5, lambda_g=0.2, lambda_r0=80.0, lambda_r1=10.0, lambda_s0=120.0, lambda_s1=10.0, lambda_a=0.1, iterations=iterations,
                lm_steps=50, fix_poses=0, fix_intrinsics=0, fix_distortion=0, output_mesh_prefix="./intrinsic3d/mesh", output_mesh_albedo=1,
                output_mesh_largest_comp_only=1, output_poses_prefix="./intrinsic3d/poses", output_intrinsics_prefix="./intrinsic3d/intrinsics")
    vals.update(cfg)
    with open(os.path.join(out, "intrinsic3d.yml"), "w") as f:
        f.write("%YAML:1.0\n\n# Intrinsic3D config\n" + "".join(f'{k}: "{v}"\n' for k, v in vals.items()))
    with open(os.path.join(out, "fusion.yml"), "w") as f:                  # data/fusion.yml of the reference; all-zero clip bounds = no clipping
        f.write('%YAML:1.0\n\n# sdf fusion config\nkeyframes: ""\n' + f'voxel_size: "{float(sc["voxel_size"]):g}"\ndiscont_window_size: "2"\n'
                + "".join(f'clip_{a}: "0.0"\n' for a in ("x0", "x1", "y0", "y1", "z0", "z1"))
                + f'output_mesh: "./fusion/mesh_{float(sc["voxel_size"]):g}.ply"\noutput_sdf: "{tsdf}"\n')
    with open(os.path.join(out, "keyframes.yml"), "w") as f:               # data/keyframes.yml of the reference
        f.write(f'%YAML:1.0\n\n# keyframe selection config\nwindow_size: "{window}"\nfilename: "./fusion/keyframes.txt"\nshow_keyframes: "0"\n')
    return os.path.join(out, "sensor.yml"), os.path.join(out, "intrinsic3d.yml")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True); ap.add_argument("--radius", type=int, default=40); ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--width", type=int, default=320); ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--grid-levels", type=int, default=3); ap.add_argument("--rgbd-levels", type=int, default=2); ap.add_argument("--iterations", type=int, default=3)
    a = ap.parse_args()
    sc = synthetic.make_scene(radius_vox=a.radius, K=a.frames, width=a.width, height=a.height, levels=1, seed=3, pose_noise=(0.001, 0.002), lum_noise=0.003)
    s, i = write_dataset(a.out, sc, grid_levels=a.grid_levels, rgbd_levels=a.rgbd_levels, iterations=a.iterations)
    print(f"{sc['keys'].shape[0]} voxels, {a.frames} frames -> {a.out}\nrun: apps/app_intrinsic3d -s {s} -i {i}")


if __name__ == "__main__":
    main()
