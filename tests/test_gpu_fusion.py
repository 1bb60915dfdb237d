"""-m gpu: TSDF fusion on the device through the C ABI against the oracle's restatement of
SparseVoxelGrid::integrate / alloc, correctSDF and clearInvalidVoxels.  Everything is held to bit-exactness INCLUDING the record order of
the saved volume (the reference's unordered_map iteration order, which depends on the order of first insertion of every voxel)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers  # noqa: E402


def _frames(seed=9, K=6, radius=14, w=128, h=96, noise=0.0):
    from intrinsic3d_amd import synthetic
    from make_dataset import pose_vec_to_cam_to_world
    sc = synthetic.make_scene(radius_vox=radius, K=K, width=w, height=h, levels=1, seed=seed)
    rng = np.random.default_rng(seed)
    out = []
    for fr, pose in zip(sc["frames"], sc["poses"]):
        d = fr["depth"][0].copy()
        if noise > 0:
            d[d > 0] += rng.normal(0, noise, int((d > 0).sum())).astype(np.float32)
        bgr = fr["bgr"][0].copy(); bgr[..., 0] = bgr[..., 0] // 2; bgr[..., 2] = 255 - bgr[..., 2] // 3          # three distinct channels
        out.append((d, bgr, pose_vec_to_cam_to_world(np.asarray(pose, np.float64)).astype(np.float32)))
    return sc, out


def _same(a, b):
    assert a["keys"].shape == b["keys"].shape, (a["keys"].shape, b["keys"].shape)
    for k in ("keys", "sdf", "weight", "color"):
        assert np.array_equal(a[k], b[k]), (k, int((a[k] != b[k]).sum()))


def test_fusion_matches_oracle_bit_exact(oracle):
    from intrinsic3d_amd import binding as B
    sc, frames = _frames(noise=0.0015)
    intr = sc["intr"].astype(np.float32)
    o = oracle.Fusion(sc["voxel_size"], 0.1, 10.0)
    with B.Fusion(sc["voxel_size"], 0.1, 10.0, initial_capacity=1 << 10) as f:          # 4096-slot table: it has to grow four times
        for d, bgr, T in frames:
            o.integrate(d, intr, bgr, intr, T, 2); f.integrate(d, intr, bgr, intr, T, 2)
        raw_o = o.export()
        o.finish(10); n = f.finish(10)
        ref = o.export(); got = f.export(); info = f.info()
    assert info["frames"] == len(frames) and info["allocated"] == len(raw_o["sdf"]) and info["capacity"] >= (1 << 16)
    assert n == len(ref["sdf"]) and 5000 < n < info["allocated"]                         # clearInvalidVoxels removed the never-seen blocks
    assert (ref["weight"] == 1.0).sum() > 100, "correctSDF did not touch anything: the sweep emulation is not exercised"
    _same(got, ref)
    assert info["correct_launches"] >= 2


def test_fusion_separate_cameras_clip_and_no_erosion(oracle, tmp_path):
    """depth camera at half resolution with its own intrinsics, clip bounds that cut the object, no erosion, 3 correction sweeps"""
    from intrinsic3d_amd import binding as B
    sc, frames = _frames(seed=4, K=5, radius=12)
    ci = sc["intr"].astype(np.float32); di = (ci * 0.5).astype(np.float32)
    c = np.asarray(sc["keys"], np.float64).mean(0) * sc["voxel_size"]
    clip = np.array([c[0] - 1.0, c[0] + 0.01, c[1] - 1.0, c[1] + 1.0, c[2] - 1.0, c[2] + 1.0], np.float32)
    o = oracle.Fusion(sc["voxel_size"], 0.2, 3.0, clip)
    with B.Fusion(sc["voxel_size"], 0.2, 3.0, clip) as f:
        for d, bgr, T in frames:
            dd = np.ascontiguousarray(d[::2, ::2])
            o.integrate(dd, di, bgr, ci, T, 0); f.integrate(dd, di, bgr, ci, T, 0)
        o.finish(3); f.finish(3)
        ref = o.export(); got = f.export()
        _same(got, ref)
        assert len(ref["sdf"]) > 1000 and (ref["keys"][:, 0] * sc["voxel_size"]).max() <= clip[1] + 1.5 * sc["voxel_size"]
        f.save(tmp_path / "vol.tsdf")
    vol = B.tsdf_read(str(tmp_path / "vol.tsdf"))
    assert np.array_equal(vol["keys"], ref["keys"]) and np.array_equal(vol["sdf"], ref["sdf"]) and np.array_equal(vol["color"], ref["color"])
    assert abs(vol["voxel_size"] - sc["voxel_size"]) < 1e-9


def test_fusion_then_refine_round_trip(tmp_path):
    """the fused volume feeds the path: .tsdf -> i3d_set_grid_from_tsdf_records -> a mesh with the sphere's size"""
    from intrinsic3d_amd import binding as B
    sc, frames = _frames(seed=2, K=8, radius=14)
    intr = sc["intr"].astype(np.float32)
    with B.Fusion(sc["voxel_size"], 0.1, 10.0) as f:
        for d, bgr, T in frames:
            f.integrate(d, intr, bgr, intr, T, 2)
        vol = f.export()
    truth = {tuple(k): s for k, s in zip(sc["keys"], sc["sdf"])}
    err = np.array([abs(truth[tuple(k)] - s) for k, s in zip(vol["keys"], vol["sdf"]) if tuple(k) in truth and abs(truth[tuple(k)]) < 2 * sc["voxel_size"]])
    assert len(err) > 3000 and err.mean() < 0.6 * sc["voxel_size"]                       # projective TSDF vs true distance
    with B.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], vol["keys"], vol["sdf"], vol["weight"], vol["color"])
        v, c, faces = ctx.extract_mesh(False, 0, True)
    centre = v.mean(0); r = np.linalg.norm(v - centre, axis=1)
    assert len(faces) > 2000 and abs(r.mean() - 14 * sc["voxel_size"]) < 1.0 * sc["voxel_size"]


def test_fusion_errors():
    from intrinsic3d_amd import binding as B
    with pytest.raises(B.I3DError):
        B.Fusion(0.0, 0.1, 2.0)                                                          # SparseVoxelGrid::create: voxel size <= 1e-5
    with B.Fusion(0.01, 0.1, 2.0) as f:
        assert f.finish() == 0 and f.export()["keys"].shape == (0, 3)                    # nothing integrated
        with pytest.raises(B.I3DError):
            f.integrate(np.zeros((4, 4), np.float32), [1, 1, 0, 0], np.zeros((4, 4, 3), np.uint8), [1, 1, 0, 0], np.eye(4, dtype=np.float32))


def test_app_fusion_then_app_intrinsic3d(oracle, tmp_path):
    """the three CLIs back to back (keyframes -> fusion -> refinement) on a dataset folder in the reference's layout: app_fusion's volume is byte-identical to the oracle's fusion of
    the same decoded frames, and app_intrinsic3d refines it"""
    import subprocess
    from intrinsic3d_amd import binding as B, synthetic
    import make_dataset
    sc = synthetic.make_scene(radius_vox=14, K=6, width=128, height=96, levels=1, seed=9, pose_noise=(0.0005, 0.001), lum_noise=0.003)
    s_yml, i_yml = make_dataset.write_dataset(str(tmp_path), sc, grid_levels=1, rgbd_levels=1, iterations=1)
    tsdf = tmp_path / "fusion" / f"volume_{float(sc['voxel_size']):g}.tsdf"
    tsdf.unlink()                                                                     # the analytic volume the writer leaves: app_fusion must produce its own
    for name in ("app_keyframes", "app_fusion", "app_intrinsic3d"):
        assert os.path.exists(os.path.join(ROOT, "apps", name)), f"apps/{name} has not been built (run __graft_entry__.build())"
    (tmp_path / "fusion" / "keyframes.txt").unlink()                                  # ... and app_keyframes the keyframe list (window 1: every frame)
    r = subprocess.run([os.path.join(ROOT, "apps", "app_keyframes"), "-s", s_yml, "-k", str(tmp_path / "keyframes.yml")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and B.keyframes_load(str(tmp_path / "fusion" / "keyframes.txt"))[2].all(), r.stdout + r.stderr
    r = subprocess.run([os.path.join(ROOT, "apps", "app_fusion"), "-s", s_yml, "-f", str(tmp_path / "fusion.yml")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "integrating frame 5" in r.stdout, r.stdout + r.stderr
    vol = B.tsdf_read(str(tsdf))
    sensor = B.Sensor(tmp_path / "rgbd", 0, 0.1, 10.0)
    o = oracle.Fusion(sc["voxel_size"], 0.1, 10.0, np.zeros(6, np.float32))
    for i in range(sensor.num_frames):
        o.integrate(sensor.depth(i), sensor.depth_intrinsics, sensor.color(i), sensor.color_intrinsics, sensor.pose(i), 2)
    o.finish(10); ref = o.export()
    for k in ("keys", "sdf", "weight", "color"):
        assert np.array_equal(vol[k], ref[k]), k
    assert (tmp_path / "fusion" / f"mesh_{float(sc['voxel_size']):g}.ply").stat().st_size > 10000
    r = subprocess.run([os.path.join(ROOT, "apps", "app_intrinsic3d"), "-s", s_yml, "-i", i_yml], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "intrinsic3d" / "mesh_g0_p0_albedo.ply").stat().st_size > 10000 and (tmp_path / "intrinsic3d" / "poses_g0_p0.txt").exists()


def test_app_fusion_equals_the_reference_application(tmp_path):
    """apps/app_fusion on a dataset folder against the reference's own AppFusion::fuseSDF (apps/src/app_fusion.cpp:107-200, compiled into oracle/_ref over its
    own SensorI3d / KeyframeSelection / SparseVoxelGrid / MarchingCubes; PNG decoding by Pillow): the .tsdf — header and every record, in file order — and
    the mesh file, byte for byte.  Cameras on the coordinate axes (exact pose inverses), with a keyframe file that drops two of the six frames."""
    import subprocess
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    if not ref_py.available():
        pytest.skip("oracle/_ref/libref_i3d.so not built")
    folder, vs, n = helpers.axis_camera_dataset(tmp_path)
    keep = [True, False, True, True, False, True]
    B.keyframes_save(str(tmp_path / "fusion" / "keyframes.txt"), 1, np.ones(n), keep)
    (tmp_path / "sensor.yml").write_text('%YAML:1.0\n\n# rgbd sensor config\ndataset: "./rgbd/"\nmax_frames: "0"\nmin_depth: "0.05"\nmax_depth: "10.0"\n')
    (tmp_path / "fusion.yml").write_text('%YAML:1.0\n\n# sdf fusion config\nkeyframes: "./fusion/keyframes.txt"\n' + f'voxel_size: "{vs:g}"\ndiscont_window_size: "2"\n'
                                         + "".join(f'clip_{a}: "0.0"\n' for a in ("x0", "x1", "y0", "y1", "z0", "z1")) + 'output_mesh: "./fusion/mesh.ply"\noutput_sdf: "./fusion/volume.tsdf"\n')
    r = subprocess.run([os.path.join(ROOT, "apps", "app_fusion"), "-s", str(tmp_path / "sensor.yml"), "-f", str(tmp_path / "fusion.yml")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    cfg = {"keyframes": str(tmp_path / "fusion" / "keyframes.txt"), "voxel_size": f"{vs:g}", "clip_x0": 0, "clip_x1": 0, "clip_y0": 0, "clip_y1": 0, "clip_z0": 0, "clip_z1": 0,
           "discont_window_size": 2, "output_sdf": str(tmp_path / "fusion" / "ref.tsdf"), "output_mesh": str(tmp_path / "fusion" / "ref.ply")}
    assert ref_py.app_fusion(folder, cfg, 0, 0.05, 10.0)
    a = np.frombuffer(open(tmp_path / "fusion" / "volume.tsdf", "rb").read(), np.uint8); b = np.frombuffer(open(tmp_path / "fusion" / "ref.tsdf", "rb").read(), np.uint8)
    nrec = (b.size - 24) // 24
    assert a.size == b.size and nrec > 2000 and np.array_equal(a[:24], b[:24])
    assert np.array_equal(a[24:].reshape(nrec, 24)[:, :23], b[24:].reshape(nrec, 24)[:, :23])          # byte 23 of a record is the struct's padding
    assert open(tmp_path / "fusion" / "mesh.ply", "rb").read() == open(tmp_path / "fusion" / "ref.ply", "rb").read()


def test_device_fusion_matches_committed_golden():
    """the device path alone against tests/golden/fusion_small.json (CRCs generated by make_golden.py FROM THE REFERENCE'S OWN integrate / alloc / correctSDF code, oracle/_ref):
    needs neither the oracle nor the reference at run time"""
    import json, zlib
    from intrinsic3d_amd import binding as B
    import golden.make_golden as mg
    gold = mg.strip_tags(json.load(open(os.path.join(ROOT, "tests", "golden", "fusion_small.json"))))
    crc = lambda a: int(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    frames, intr, vs = mg.fusion_frames()
    with B.Fusion(vs, 0.1, 10.0, initial_capacity=1 << 14) as f:
        for d, bgr, T in frames:
            f.integrate(d, intr, bgr, intr, T, 2)
        v = f.export(); info = f.info()
    got = {"allocated": info["allocated"], "saved": len(v["sdf"]), "keys": crc(v["keys"]), "sdf": crc(v["sdf"]), "weight": crc(v["weight"]), "color": crc(v["color"]),
           "corrected": int((v["weight"] == 1.0).sum())}
    assert got == gold


def test_fusion_degenerate_frames(oracle):
    """frames that allocate nothing (all-zero depth, everything outside the clip box), a frustum whose far plane lies in front of the object,
    and a 4096-slot table that has to grow in the middle of a frame's allocation (which must then be repeated without changing the order of
    first insertion): the volume must equal the oracle's"""
    from intrinsic3d_amd import binding as B
    sc, frames = _frames(seed=6, K=3, radius=10, w=96, h=72)
    intr = sc["intr"].astype(np.float32)
    d0, bgr0, T0 = frames[0]
    far_clip = np.array([50, 51, 50, 51, 50, 51], np.float32)
    for clip, dmax in ((None, 10.0), (far_clip, 10.0), (None, 0.2)):                  # dmax 0.2 m: the object lies outside the frustum bounds
        o = oracle.Fusion(sc["voxel_size"], 0.1, dmax, clip)
        with B.Fusion(sc["voxel_size"], 0.1, dmax, clip, initial_capacity=1 << 10) as f:
            for d, bgr, T in ((np.zeros_like(d0), bgr0, T0), (d0, bgr0, T0), (frames[1][0], frames[1][1], frames[1][2]), (np.zeros_like(d0), bgr0, T0)):
                o.integrate(d, intr, bgr, intr, T, 1); f.integrate(d, intr, bgr, intr, T, 1)
            o.finish(2); f.finish(2); ref = o.export(); got = f.export()
        _same(got, ref)
        assert (len(ref["sdf"]) > 1000) == (clip is None)                  # the frustum bounds are rounded to whole METRES (sparse_voxel_grid.cpp:587-588): dmax 0.2 cuts nothing here


@pytest.mark.parametrize("case", ["two_levels", "three_levels_half_res_depth_skipped_frames"])
def test_app_intrinsic3d_equals_the_reference_application(tmp_path, case):
    """apps/app_intrinsic3d on a dataset folder against the reference's own AppIntrinsic3D flow (apps/src/app_intrinsic3d.cpp:71-210: SensorI3d, KeyframeSelection::load,
    SparseVoxelGrid::create(tsdf), Intrinsic3D::init / refine, onSDFRefined -> SDFVisualization::colorize, savePoses, Camera::save — all compiled into oracle/_ref) with
    `iterations: "0"`: the optimiser declines (optimizer.cpp:113-114) and both sides go on, so every stage's files depend only on loading, initialisation, thin
    shell, lighting, recolouring, upsampling and export — and must agree byte for byte: three stages x (the mesh in nine colour modes — voxel colours, normals, Laplacian,
    intensity, intensity gradient (painted in place in the reference: depends on the walk over the grid), albedo, shading with the estimated / a constant albedo, chromacity —
    poses, intrinsics) = 33 files."""
    import shutil
    import subprocess
    from intrinsic3d_amd import synthetic
    from oracle import ref_py
    import make_dataset
    if not ref_py.available():
        pytest.skip("oracle/_ref/libref_i3d.so not built")
    sc = synthetic.make_scene(radius_vox=10, K=4, width=96, height=72, levels=1, seed=9, pose_noise=(0.0005, 0.001), lum_noise=0.003)
    GL, PL, extra = (2, 2, 0) if case == "two_levels" else (3, 3, 2)
    s_yml, i_yml = make_dataset.write_dataset(str(tmp_path), sc, grid_levels=GL, rgbd_levels=PL, iterations=0, extra_frames=extra)
    if case != "two_levels":                                                           # depth maps at half the colour resolution with their own intrinsics: resizeDepth proper
        from PIL import Image
        K = np.loadtxt(tmp_path / "rgbd" / "depthIntrinsics.txt"); K[:2, :3] *= 0.5; np.savetxt(tmp_path / "rgbd" / "depthIntrinsics.txt", K, fmt="%.9g")
        for f in sorted((tmp_path / "rgbd").glob("*.depth.png")):
            Image.fromarray(np.asarray(Image.open(f))[::2, ::2].copy()).save(f)
    import re
    txt = open(i_yml).read(); assert 'subvolume_size_sh: "0.2"' in txt
    open(i_yml, "w").write(txt.replace('subvolume_size_sh: "0.2"', 'subvolume_size_sh: "0.03"'))     # 3 cm subvolumes: the 8 cm object spans several (interpolated shading)
    with open(i_yml, "a") as f:                                                        # every debug view of SDFVisualization::getOutputModes that can be reproduced
        f.write("".join(f'output_mesh_{k}: "1"\n' for k in ("normals", "laplacian", "intensity", "intensity_grad", "shading_sv", "shading_sv_const", "chromacity")))
    cfg = dict(re.findall(r'^(\w+): "(.*)"$', open(i_yml).read(), re.M))
    out = tmp_path / "intrinsic3d"; out.mkdir(exist_ok=True)
    r = subprocess.run([os.path.join(ROOT, "apps", "app_intrinsic3d"), "-s", s_yml, "-i", i_yml], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    ours = tmp_path / "ours"; shutil.move(str(out), str(ours)); out.mkdir()
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        assert ref_py.app_intrinsic3d("./rgbd/", cfg, 0, 0.1, 10.0)
    finally:
        os.chdir(cwd)
    names = sorted(os.listdir(out))
    views = ("", "_normals", "_lap", "_lum", "_lum_grad", "_albedo", "_shading_sv", "_shading_sv_const", "_chroma")
    stages = [f"g{GL - 1}_p{p}" for p in range(PL - 1, -1, -1)] + [f"g{g}_p0" for g in range(GL - 2, -1, -1)]
    assert names == sorted(f"{p}_{s}{e}" for s in stages for p, e in [("intrinsics", ".txt"), ("poses", ".txt")] + [("mesh", v + ".ply") for v in views])
    report = {n: (os.path.exists(ours / n) and open(ours / n, "rb").read() == open(out / n, "rb").read()) for n in names}
    if not all(report.values()) or sorted(os.listdir(ours)) != names:
        keep = os.path.join(ROOT, "gpurun_out", "app_i3d_mismatch")                        # (for a look afterwards)
        shutil.rmtree(keep, ignore_errors=True); os.makedirs(keep)
        shutil.copytree(ours, os.path.join(keep, "ours")); shutil.copytree(out, os.path.join(keep, "ref"))
        open(os.path.join(keep, "app.log"), "w").write(r.stdout + r.stderr)
    assert sorted(os.listdir(ours)) == names, sorted(os.listdir(ours))
    assert all(report.values()), report
    assert (out / "mesh_g0_p0.ply").stat().st_size > 100000
