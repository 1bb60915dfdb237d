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


def test_device_fusion_matches_committed_golden():
    """the device path alone against tests/golden/fusion_small.json (CRCs of the ORACLE's fused volume on seeded frames, tests/golden/make_golden.py —
    regression vectors, not reference outputs): needs no oracle library at run time"""
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


def test_fusion_around_the_origin_truncating_round(oracle):
    """Row a2 of the scope table on the device: `worldToVoxel` is `(p / voxel_size + 0.5).cast<int>()` (sparse_voxel_grid.cpp:211-228 with mat.h:88-93) — truncation toward
    ZERO, not floor: -0.7 -> 0, so the voxels just below a coordinate plane are half a voxel 'wider' than the others.  The scene of the first test, shifted so that the object
    sits AROUND the origin (every frame's camera-to-world translation moved by the object's centre rounded to whole voxels): keys of every sign in all three coordinates,
    and the allocation (`alloc`: voxel of a back-projected point), the integration and the record order must still equal the oracle's bit for bit.  Against the unshifted
    run the key set must NOT be a pure translation (the planes x, y, z = 0 are where truncation and floor disagree) — otherwise the test would not see the quirk."""
    from intrinsic3d_amd import binding as B
    sc, frames = _frames(noise=0.0015)
    intr = sc["intr"].astype(np.float32); vs = float(sc["voxel_size"])
    shift_vox = np.round(np.asarray(sc["center"], np.float64) / vs).astype(np.int64)
    shift = (shift_vox * vs).astype(np.float32)
    moved = []
    for d, bgr, T in frames:
        T2 = T.copy(); T2[:3, 3] = T2[:3, 3] - shift; moved.append((d, bgr, T2))
    res = {}
    for name, fs in (("origin", moved), ("positive", frames)):
        o = oracle.Fusion(sc["voxel_size"], 0.1, 10.0)
        with B.Fusion(sc["voxel_size"], 0.1, 10.0, initial_capacity=1 << 12) as f:
            for d, bgr, T in fs:
                o.integrate(d, intr, bgr, intr, T, 2); f.integrate(d, intr, bgr, intr, T, 2)
            o.finish(10); f.finish(10)
            ref = o.export(); got = f.export()
        _same(got, ref)
        res[name] = ref
    k = res["origin"]["keys"]
    assert all((k[:, a] < 0).any() and (k[:, a] > 0).any() and (k[:, a] == 0).any() for a in range(3)), (k.min(0), k.max(0))
    a = set(map(tuple, k.tolist())); b = set(map(tuple, (res["positive"]["keys"] - shift_vox.astype(np.int32)).tolist()))
    assert len(a) > 5000 and a != b, "the shifted volume is a pure translation of the unshifted one: the truncating round made no difference on this scene"
