"""-m gpu: Intrinsic3D::init's keyframe loop from a dataset folder — PNG decode and pose conversion on the host,
depth resampling + pyramids on the device — against the oracle's restatement of resizeDepth / Pyramid fed with the same decoded frames."""
import numpy as np
import pytest

from test_loader_cpu import _make_dataset

pytestmark = pytest.mark.gpu


def test_init_frames_from_dataset_folder(oracle, tmp_path):
    from intrinsic3d_amd import binding as B
    rng = np.random.default_rng(11)
    _make_dataset(tmp_path / "rgbd", 7, rng, cw=128, ch=96, dw=64, dh=48)
    sensor = B.Sensor(tmp_path / "rgbd", 0, 0.1, 2.5)
    is_kf = np.array([1, 0, 0, 1, 1, 0, 1, 1, 1], np.uint8)                  # longer than the dataset: extra flags are ignored
    levels = 3
    with B.Context(0) as ctx:
        ids = B.init_frames_from_sensor(ctx, sensor, is_kf, levels)
        assert ids.tolist() == [0, 3, 4, 6]
        intr, dist, poses = ctx.get_camera()
        np.testing.assert_array_equal(intr, sensor.color_intrinsics.astype(np.float64)); assert not dist.any()
        for k, fid in enumerate(ids):
            np.testing.assert_array_equal(poses[k], B.pose_mat_to_vec6(sensor.pose(fid)))
            depth = oracle.resize_depth(sensor.depth(fid), sensor.depth_intrinsics, 128, 96, sensor.color_intrinsics)
            assert (depth > 0).mean() > 0.2
            lum = oracle.lum_from_bgr(sensor.color(fid))
            for lvl in range(levels):
                got_l, got_d = ctx.get_frame_image(k, lvl, 128 >> lvl, 96 >> lvl)
                assert np.array_equal(got_l, lum) and np.array_equal(got_d, depth), (k, lvl)
                if lvl + 1 < levels:
                    lum = oracle.pyr_down(lum); depth = oracle.depth_down(depth)
    # no keyframe selected -> error, like an empty image model
    with B.Context(0) as ctx, pytest.raises(B.I3DError):
        B.init_frames_from_sensor(ctx, sensor, np.zeros(7, np.uint8), 1)


def test_app_intrinsic3d_end_to_end(tmp_path):
    """apps/app_intrinsic3d (the AppIntrinsic3D equivalent, C++ over the C ABI) on a dataset folder in the reference's layout: it must write the
    reference's per-level outputs, and they must agree with the same flow driven in-process through the binding."""
    import os, subprocess, sys
    from intrinsic3d_amd import binding as B, synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_dataset
    app = os.path.join(root, "apps", "app_intrinsic3d")
    assert os.path.exists(app), "apps/app_intrinsic3d has not been built (run __graft_entry__.build())"
    sc = synthetic.make_scene(radius_vox=14, K=6, width=128, height=96, levels=1, seed=9, pose_noise=(0.001, 0.002), lum_noise=0.003)
    s_yml, i_yml = make_dataset.write_dataset(str(tmp_path), sc, grid_levels=2, rgbd_levels=2, iterations=2, extra_frames=2)
    r = subprocess.run([app, "-s", s_yml, "-i", i_yml], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = tmp_path / "intrinsic3d"
    assert sorted(p.name for p in out.glob("poses_*")) == ["poses_g0_p0.txt", "poses_g1_p0.txt", "poses_g1_p1.txt"]   # all pyramid levels only on the coarsest grid
    for gl, pl in ((1, 1), (1, 0), (0, 0)):
        if True:
            for name in (f"mesh_g{gl}_p{pl}.ply", f"mesh_g{gl}_p{pl}_albedo.ply", f"poses_g{gl}_p{pl}.txt", f"intrinsics_g{gl}_p{pl}.txt"):
                assert (out / name).stat().st_size > 0, name
    poses_app = np.loadtxt(out / "poses_g0_p0.txt")
    assert poses_app.shape == (8, 8) and np.array_equal(poses_app[:, 0], np.arange(8.0))        # one line per FRAME, keyframe or not
    ok, w, h, intr_app, dist_app = B.read_intrinsics(str(out / "intrinsics_g0_p0.txt"))
    assert ok and (w, h) == (128, 96)

    # the same flow in-process
    sensor = B.Sensor(tmp_path / "rgbd", 0, 0.1, 10.0)
    _, _, is_kf = B.keyframes_load(str(tmp_path / "fusion" / "keyframes.txt"))
    assert is_kf.tolist() == [True] * 6 + [False] * 2
    rc, oc = B.load_yaml_config(i_yml)
    vol = B.tsdf_read(str(tmp_path / "fusion" / f"volume_{float(sc['voxel_size']):g}.tsdf"))
    with B.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(vol["voxel_size"], vol["keys"], vol["sdf"], vol["weight"], vol["color"])
        ids = B.init_frames_from_sensor(ctx, sensor, is_kf, rc.num_rgbd_levels)
        ctx.refine(rc, oc)
        intr, dist, poses = ctx.get_camera()
        verts, _, faces = ctx.extract_mesh(True, 0, True)
    for k, fid in enumerate(ids):
        sensor.set_pose_vec6(fid, poses[k])
    sensor.save_poses(tmp_path / "poses_inproc.txt")
    poses_in = np.loadtxt(tmp_path / "poses_inproc.txt")
    assert np.allclose(poses_app, poses_in, atol=5e-5), np.abs(poses_app - poses_in).max()       # fp32 atomics order differs run to run
    assert np.allclose(intr_app, intr, rtol=1e-4) and np.allclose(dist_app, dist, rtol=2e-3, atol=1e-5)      # 6 significant digits in the file; distortion is weakly determined on this tiny scene
    # the refined poses moved (noise was added to the keyframe poses) and the non-keyframes kept their input pose
    p0 = np.array([np.r_[sensor.pose(i)[:3, 3]] for i in range(8)])
    assert np.allclose(poses_app[6:, 1:4], p0[6:], atol=1e-5)
    with open(out / "mesh_g0_p0.ply", "rb") as f:
        head = f.read(400)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n")
    nv_app = int(head.split(b"element vertex ")[1].split(b"\n")[0]); nf_app = int(head.split(b"element face ")[1].split(b"\n")[0])
    assert abs(nv_app - len(verts)) <= 0.01 * len(verts) and abs(nf_app - len(faces)) <= 0.01 * len(faces) and nv_app > 1000
