"""-m gpu: the configurations BASELINE.json names, at test size, through the refine schedule (i3d_refine / apps/app_intrinsic3d) against
oracle.refine (Intrinsic3D::refine, intrinsic3d.cpp:206-350).  configs[0] (C1) is tests/test_gpu_parity.py::test_config_c1_dense_albedo_only.

  C2  single level at 4 mm, fixed camera (poses, intrinsics, distortion), ONE global SH volume
  C3  3 grid levels (4 -> 2 -> 1 mm) x (3, 1, 1) pyramid levels, poses fixed, joint SDF + albedo + spatially varying SH, on a dataset folder
      in the reference's layout through apps/app_intrinsic3d and through the same flow in-process
  C5  the same schedule with EVERY group free (poses, intrinsics, distortion: the shipped data/intrinsic3d.yml), noisy input poses

Structure (keys, visit order, weights) must be identical; fields are held to 1e-4 on the 99.9 % quantile and to
max(1e-4, helpers.ENVELOPE_FACTOR x the oracle's own sensitivity to 1e-7 input perturbations) on the maximum."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_refine(O, sc, frames, levels, ocfg, rc, intr, dist, poses, pose_eps=0.0):
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    fr = O.Frames(frames, levels)
    rcode, ointr, odist, oposes, done = O.refine(g, fr, ocfg, rc.num_grid_levels, rc.num_rgbd_levels, rc.thin_shell_factor, rc.thin_shell_factor_final,
                                                 rc.clear_distant_voxels, rc.subvolume_size_sh, rc.sh_lambda_reg, intr, dist,
                                                 np.array(poses, np.float64) * (1.0 + pose_eps))
    assert rcode == 0
    ref = g.export(); g.free(); fr.free()
    return ref, ointr, oposes, done


def _check_fields(out, ref, env, ordered=True, free_camera=False):
    # voxels on one side only (a re-sparsification threshold within the two computations' difference): as many as the reference computation itself flips under
    # 1e-7 input perturbations allow, times the one envelope factor — 2e-4 of the voxels at least
    out, ref = helpers.align_by_key(out, ref, max_frac=max(2e-4, helpers.ENVELOPE_FACTOR * env.get("key_frac", 0.0)), ordered=ordered)
    assert (out["weight"] != ref["weight"]).mean() <= max(2e-4, helpers.ENVELOPE_FACTOR * env.get("key_frac", 0.0)), int((out["weight"] != ref["weight"]).sum())      # (a child next to a differing voxel interpolates other corners)
    d_sdf = np.abs(out["sdf_refined"] - ref["sdf_refined"]); d_alb = np.abs(out["albedo"] - ref["albedo"])
    smax = float(np.abs(ref["sdf_refined"]).max()); amax = float(np.abs(ref["albedo"]).max())
    if free_camera:      # free poses on a near-symmetric object leave a gauge direction: the bulk of the field inside the reference computation's own spread (as in
                         # test_gpu_levels.py::test_refine_two_levels_matches_oracle), the median far below it
        print(f"\n[schedule, free camera] sdf: median {np.median(d_sdf) / smax:.2e}, 99.9 % {np.quantile(d_sdf, 0.999) / smax:.2e}, max {d_sdf.max() / smax:.2e} of max |sdf|; "
              f"oracle's own spread under 1e-7 input perturbations {env['sdf_refined'] / smax:.2e}; albedo: median {np.median(d_alb):.2e}, 99.9 % {np.quantile(d_alb, 0.999):.2e}, "
              f"max {d_alb.max():.2e}, spread {env['albedo']:.2e}; one-sided voxels under the perturbations {env.get('key_frac', 0.0):.2e}")
        assert np.quantile(d_sdf, 0.999) <= max(1e-4 * smax, env["sdf_refined"]), (np.quantile(d_sdf, 0.999), smax, env)
        assert np.quantile(d_alb, 0.999) <= max(1e-4 * amax, env["albedo"]), (np.quantile(d_alb, 0.999), amax, env)
    else:
        assert np.quantile(d_sdf, 0.999) <= 1e-4 * smax, (np.quantile(d_sdf, 0.999), smax)
        assert np.quantile(d_alb, 0.999) <= 1e-4 * amax, (np.quantile(d_alb, 0.999), amax)
    assert d_sdf.max() <= max(1e-4 * smax, helpers.ENVELOPE_FACTOR * env["sdf_refined"]), (d_sdf.max(), smax, env)
    assert d_alb.max() <= max(1e-4 * amax, helpers.ENVELOPE_FACTOR * env["albedo"]), (d_alb.max(), env)
    cd = np.abs(out["color"].astype(int) - ref["color"].astype(int))
    # 8-bit truncation of colours computed from ~1e-7-different geometry; with a free camera the keyframe poses themselves differ inside the envelope and
    # the recolourisation samples the images elsewhere: as many components as the oracle's own perturbed runs change, times the one factor
    assert (cd > 1).mean() < max(1e-3, helpers.ENVELOPE_FACTOR * env.get("color_frac", 0.0)), ((cd > 1).mean(), env.get("color_frac"))


def _envelope(O, sc, frames, levels, ocfg, rc, intr, dist, poses, ref, eps_list):
    env = dict(sdf_refined=0.0, albedo=0.0, key_frac=0.0, poses=0.0, intr=0.0, color_frac=0.0)
    for eps in eps_list:
        per, pintr, pposes, _ = _oracle_refine(O, sc, frames, levels, ocfg, rc, intr, dist, poses, pose_eps=eps)
        env["poses"] = max(env["poses"], float(np.abs(pposes - ref["_poses"]).max())) if "_poses" in ref else env["poses"]
        env["intr"] = max(env["intr"], float(np.abs(pintr - ref["_intr"]).max() / np.abs(ref["_intr"]).max())) if "_intr" in ref else env["intr"]
        if per["keys"].shape == ref["keys"].shape and np.array_equal(per["keys"], ref["keys"]):
            for k in ("sdf_refined", "albedo"):
                env[k] = max(env[k], float(np.abs(per[k] - ref[k]).max()))
            env["color_frac"] = max(env["color_frac"], float((np.abs(per["color"].astype(int) - ref["color"].astype(int)) > 1).mean()))
        else:       # the perturbed reference keeps / drops other voxels at a re-sparsification threshold: how many, and the fields on the common ones
            so, sr = set(map(tuple, per["keys"].tolist())), set(map(tuple, ref["keys"].tolist()))
            env["key_frac"] = max(env["key_frac"], len(so ^ sr) / float(len(sr)))
            a, b = helpers.align_by_key({k: v for k, v in per.items()}, {k: v for k, v in ref.items() if not k.startswith("_")}, max_frac=1.0, ordered=False)
            for k in ("sdf_refined", "albedo"):
                env[k] = max(env[k], float(np.abs(a[k] - b[k]).max()))
            env["color_frac"] = max(env["color_frac"], float((np.abs(a["color"].astype(int) - b["color"].astype(int)) > 1).mean()))
    return env


def test_config_c2_single_level_fixed_camera_global_sh(oracle):
    """BASELINE.json configs[1]: one grid level at 4 mm, one pyramid level, camera fixed, one SH volume for the whole object (a subvolume
    size larger than the scene: intrinsic3d.yml's `subvolume_size_sh` with a single cell), the shipped lambda schedule."""
    from intrinsic3d_amd import binding
    sc = helpers.small_scene(seed=21, radius_vox=14, K=8, width=160, height=120, levels=1, pose_noise=(0.0, 0.0), lum_noise=0.003)
    ocfg = helpers.oracle_cfg(oracle, 0.0, iterations=3, lm_steps=50, fix_poses=1, fix_intrinsics=1, fix_distortion=1)
    rc = binding.RefineConfig(num_grid_levels=1, num_rgbd_levels=1, thin_shell_factor=2.0, thin_shell_factor_final=2.0, clear_distant_voxels=1,
                              occlusion_distance=0.02, num_observations=5, subvolume_size_sh=10.0, sh_lambda_reg=10.0)
    seen = []
    with binding.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        ctx.set_frames(sc["frames"], sc["levels"]); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        ctx.refine(rc, helpers.gpu_cfg(ocfg), callback=lambda gl, ng, pl, npl: seen.append((gl, pl)))
        out = ctx.export_grid(); intr, dist, poses = ctx.get_camera()
        sh, _, _ = ctx.estimate_sh(10.0, 10.0, 2.0 * float(sc["voxel_size"]))
    assert seen == [(0, 0)] and sh.shape[0] == 1                              # one level, ONE SH volume
    ref, ointr, oposes, done = _oracle_refine(oracle, sc, sc["frames"], 1, ocfg, rc, sc["intr"], sc["dist"], sc["poses"])
    assert done == 1
    np.testing.assert_array_equal(intr, ointr); np.testing.assert_array_equal(poses, oposes)          # fixed blocks come back untouched
    env = _envelope(oracle, sc, sc["frames"], 1, ocfg, rc, sc["intr"], sc["dist"], sc["poses"], ref, ())
    _check_fields(out, ref, env)                                             # a fixed camera leaves no gauge freedom: plain 1e-4
    assert np.abs(out["sdf_refined"] - out["sdf"]).max() > 1e-3 * float(sc["voxel_size"])             # the geometry did move


def _three_level_schedule(oracle, tmp_path, *, seed, fix_poses, fix_distortion, iterations, pose_noise, subvolume):
    """The reference's coarse-to-fine schedule on a dataset folder, through apps/app_intrinsic3d and in-process, against oracle.refine on the same decoded
    keyframes.  Returns what the callers assert on."""
    from intrinsic3d_amd import binding as B, synthetic
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_dataset
    app = os.path.join(ROOT, "apps", "app_intrinsic3d")
    assert os.path.exists(app), "apps/app_intrinsic3d has not been built (run __graft_entry__.build())"
    levels = 3
    sc = synthetic.make_scene(radius_vox=10, K=6, width=192, height=144, levels=1, seed=seed, pose_noise=pose_noise, lum_noise=0.003, cam_dist=0.2)      # surface beyond sensor.yml's min_depth 0.1 m
    s_yml, i_yml = make_dataset.write_dataset(str(tmp_path), sc, grid_levels=3, rgbd_levels=levels, iterations=iterations, fix_poses=fix_poses, fix_distortion=fix_distortion,
                                              subvolume_size_sh=subvolume)
    r = subprocess.run([app, "-s", s_yml, "-i", i_yml], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    outdir = tmp_path / "intrinsic3d"
    want = [(2, 2), (2, 1), (2, 0), (1, 0), (0, 0)]                           # all pyramid levels only on the coarsest grid
    assert sorted(p.name for p in outdir.glob("poses_*")) == sorted(f"poses_g{g}_p{p}.txt" for g, p in want)
    for g, p in want:
        for name in (f"mesh_g{g}_p{p}.ply", f"mesh_g{g}_p{p}_albedo.ply", f"intrinsics_g{g}_p{p}.txt"):
            assert (outdir / name).stat().st_size > 0, name
    ok, w, h, intr_app, dist_app = B.read_intrinsics(str(outdir / "intrinsics_g0_p0.txt"))
    assert ok and (w, h) == (192, 144)

    # the same flow in-process, and the oracle on the same decoded keyframes
    sensor = B.Sensor(tmp_path / "rgbd", 0, 0.1, 10.0)
    _, _, is_kf = B.keyframes_load(str(tmp_path / "fusion" / "keyframes.txt"))
    rc, oc = B.load_yaml_config(i_yml)
    assert (rc.num_grid_levels, rc.num_rgbd_levels, oc.fix_poses, oc.fix_intrinsics, oc.fix_distortion) == (3, 3, fix_poses, 0, fix_distortion)
    vol = B.tsdf_read(str(tmp_path / "fusion" / f"volume_{float(sc['voxel_size']):g}.tsdf"))
    seen = []
    with B.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(vol["voxel_size"], vol["keys"], vol["sdf"], vol["weight"], vol["color"])
        ids = B.init_frames_from_sensor(ctx, sensor, is_kf, levels)
        intr0, dist0, poses0 = ctx.get_camera()
        frames = []
        for k, fid in enumerate(ids):
            bgr0 = sensor.color(fid)
            lum, dep, bgr = [], [], []
            for lvl in range(levels):
                l, d = ctx.get_frame_image(k, lvl, 192 >> lvl, 144 >> lvl)
                lum.append(l); dep.append(d); bgr.append(np.ascontiguousarray(bgr0[::1 << lvl, ::1 << lvl]))      # colours are sampled at level 0 only
            frames.append({"lum": lum, "depth": dep, "bgr": bgr})
        ctx.refine(rc, oc, callback=lambda gl, ng, pl, npl: seen.append((gl, pl)))
        out = ctx.export_grid(); intr, dist, poses = ctx.get_camera()
    assert seen == want
    assert np.allclose(intr_app, intr, rtol=1e-4)                            # the app and the in-process flow agree (6 significant digits in the file)
    ocfg = helpers.oracle_cfg(oracle, 0.0, iterations=oc.iterations, lm_steps=oc.lm_steps, fix_poses=fix_poses, fix_intrinsics=0, fix_distortion=fix_distortion)
    vsc = dict(voxel_size=vol["voxel_size"], keys=vol["keys"], sdf=vol["sdf"], weight=vol["weight"], color=vol["color"])
    ref, ointr, oposes, done = _oracle_refine(oracle, vsc, frames, levels, ocfg, rc, intr0, dist0, poses0)
    assert done == 5
    ref["_poses"] = oposes; ref["_intr"] = ointr
    env = _envelope(oracle, vsc, frames, levels, ocfg, rc, intr0, dist0, poses0, ref, (1e-7, -1e-7))
    del ref["_poses"], ref["_intr"]
    assert float(vol["voxel_size"]) / 4.0 == pytest.approx(0.001)            # 4 mm -> 1 mm
    return dict(out=out, ref=ref, env=env, intr=intr, ointr=ointr, poses=poses, oposes=oposes, poses0=poses0, dist=dist, intr0=intr0,
                oracle_args=(vsc, frames, levels, ocfg, rc, intr0, dist0, poses0))


def test_config_c3_three_level_schedule_through_the_app(oracle, tmp_path):
    """BASELINE.json configs[2] at test size: the reference's coarse-to-fine schedule — 3 grid levels, 3 pyramid levels on the coarsest grid and
    the finest pyramid level on the others (intrinsic3d.cpp:233-247) — with poses fixed and SDF + albedo + SVSH + intrinsics joint.  The dataset
    folder is in the reference's layout (tools/make_dataset.py); apps/app_intrinsic3d runs it end to end; the same flow in-process is compared
    with oracle.refine fed with the decoded keyframes."""
    r = _three_level_schedule(oracle, tmp_path, seed=33, fix_poses=1, fix_distortion=1, iterations=2, pose_noise=(0.0, 0.0), subvolume=0.05)
    np.testing.assert_array_equal(r["poses"], r["poses0"])                   # fixed
    _check_fields(r["out"], r["ref"], r["env"])
    assert np.abs(r["intr"] - r["ointr"]).max() <= 1e-4 * np.abs(r["ointr"]).max(), (r["intr"], r["ointr"])


def test_config_c5_full_joint_refinement_with_free_camera(oracle, tmp_path, capsys, monkeypatch):
    """BASELINE.json configs[4] at test size: the FULL joint problem — SDF + albedo + spatially varying SH + poses + intrinsics + distortion, every switch of the
    shipped data/intrinsic3d.yml (fix_poses 0, fix_intrinsics 0, fix_distortion 0; 3 grid levels x (3, 1, 1) pyramid levels; subvolume_size_sh 0.2 m scaled to the
    8 cm object: 0.03 m) — from a dataset folder with noisy input poses (2 mm / 0.2 deg) through apps/app_intrinsic3d, against oracle.refine (Intrinsic3D::refine,
    refinement/intrinsic3d.cpp:229-290; Optimizer::optimize with the camera blocks free, optimizer.cpp:296-306).  Fields by key: 99.9 % quantile <= 1e-4, the
    maximum inside the reference computation's own sensitivity (free poses on a near-symmetric object leave a gauge direction); poses / intrinsics reported.
    Measured: on this 4 cm synthetic sphere the free-camera schedule WANDERS — the oracle's own poses move 5e-2 and its fields 4e-2 of max |sdf| when its input poses
    are perturbed by 1e-7 (every scene variant tried: more bumps, more keyframes, no pose noise) — so this test shows that the whole pipeline agrees with the oracle
    as far as the oracle agrees with itself (median 2e-4, 99.9 % 8e-3 of max |sdf|); the 1e-4 statements with free poses are the single-level tests."""
    # Run in the bit-reproducible mode (I3D_DETERMINISTIC=1, inherited by the application's process): on this ill-conditioned problem two runs of the DEFAULT mode
    # (fp32 LDS atomics inside the operator pass) end 1e-4 apart in the intrinsics — measured — which is the comparison between the application and the in-process
    # flow below, not the parity with the oracle
    monkeypatch.setenv("I3D_DETERMINISTIC", "1")
    r = _three_level_schedule(oracle, tmp_path, seed=35, fix_poses=0, fix_distortion=0, iterations=3, pose_noise=(0.002, 0.0035), subvolume=0.03)
    assert np.abs(r["poses"] - r["poses0"]).max() > 1e-5 and np.abs(r["intr"] - r["intr0"]).max() > 1e-4      # the camera did move
    with capsys.disabled():
        _check_fields(r["out"], r["ref"], r["env"], ordered=False, free_camera=True)
    # the camera against the oracle's: its own sensitivity to 1e-7 input perturbations (two perturbed re-runs) bounds what can be asked of poses on this object
    spread_p, spread_i = r["env"]["poses"], r["env"]["intr"]
    d_pose = float(np.abs(r["poses"] - r["oposes"]).max()); d_intr = float(np.abs(r["intr"] - r["ointr"]).max() / np.abs(r["ointr"]).max())
    with capsys.disabled():
        print(f"\n[C5] voxels on one side only under 1e-7 perturbations of the ORACLE's input: {r['env']['key_frac']:.2e} of the grid; poses: device vs oracle {d_pose:.3e} (oracle's own spread under 1e-7 input perturbations {spread_p:.3e}); intrinsics {d_intr:.3e} relative (spread {spread_i:.3e}); "
              f"pose update {np.abs(r['poses'] - r['poses0']).max():.3e}")
    assert d_intr <= max(1e-4, helpers.ENVELOPE_FACTOR * spread_i), (d_intr, spread_i)
    assert d_pose <= max(1e-4 * max(1.0, float(np.abs(r["oposes"]).max())), helpers.ENVELOPE_FACTOR * spread_p), (d_pose, spread_p)


def test_free_camera_schedule_in_the_lds_atomic_mode_against_the_bit_reproducible_one(tmp_path, monkeypatch):
    """The C5 schedule (every group free, 3 grid levels x (3, 1, 1) pyramid levels, dataset folder) in the LDS-ATOMIC mode (I3D_DETERMINISTIC=0: fp32 LDS atomics inside k_eg_tile,
    which the lone systems of the damping ladder go through; the default up to round 4 and still the default of a sharded run) against the same schedule in the
    bit-reproducible mode (the default on one rank), which the test above compares with the oracle.  The two differ by summation-order noise only, and this gauge-free schedule amplifies noise: the test above measures the ORACLE's own spread under 1e-7
    perturbations of its input at 4.0e-2 (sdf) / 1.8e-2 (albedo) of the field maximum, 4.8e-2 in the poses, 5.7e-4 in the intrinsics, and the bit-reproducible run
    against the oracle at 99.9 % 6.9e-3 / max 1.7e-2.  Measured here (MI355X, round 5): sdf 99.9 % 4.6e-3, max 1.7e-2; albedo 2.3e-3 / 1.1e-2; intrinsics 9.2e-5;
    poses 6.5e-4; 6 of 121 167 voxels on one side only.  The bars are those spreads: median 1e-3, 99.9 % 1e-2, max 4e-2 of the field maximum; intrinsics 1e-3
    relative, poses 5e-3; at most 2e-3 of the voxels kept on one side only (advisor finding of round 4: no free-camera schedule test covered the default mode)."""
    from intrinsic3d_amd import binding as B, synthetic
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_dataset
    levels = 3
    sc = synthetic.make_scene(radius_vox=10, K=6, width=192, height=144, levels=1, seed=35, pose_noise=(0.002, 0.0035), lum_noise=0.003, cam_dist=0.2)
    s_yml, i_yml = make_dataset.write_dataset(str(tmp_path), sc, grid_levels=3, rgbd_levels=levels, iterations=3, fix_poses=0, fix_distortion=0, subvolume_size_sh=0.03)
    sensor = B.Sensor(tmp_path / "rgbd", 0, 0.1, 10.0)
    _, _, is_kf = B.keyframes_load(str(tmp_path / "fusion" / "keyframes.txt"))
    rc, oc = B.load_yaml_config(i_yml)
    vol = B.tsdf_read(str(tmp_path / "fusion" / f"volume_{float(sc['voxel_size']):g}.tsdf"))

    def run():
        with B.Context(0) as ctx:
            ctx.set_grid_from_tsdf_records(vol["voxel_size"], vol["keys"], vol["sdf"], vol["weight"], vol["color"])
            B.init_frames_from_sensor(ctx, sensor, is_kf, levels)
            ctx.refine(rc, oc)
            return ctx.export_grid(), ctx.get_camera()
    monkeypatch.setenv("I3D_DETERMINISTIC", "1")
    det, (di, dd, dp) = run()
    monkeypatch.setenv("I3D_DETERMINISTIC", "0")
    dflt, (fi, fd, fp) = run()
    a, b = helpers.align_by_key(dflt, det, max_frac=2e-3, ordered=False)
    e_sdf = np.abs(a["sdf_refined"] - b["sdf_refined"]) / np.abs(b["sdf_refined"]).max(); e_alb = np.abs(a["albedo"] - b["albedo"]) / np.abs(b["albedo"]).max()
    d_intr = float(np.abs(fi - di).max() / np.abs(di).max()); d_pose = float(np.abs(fp - dp).max())
    print(f"\n[C5, LDS-atomic vs bit-reproducible mode] sdf: median {np.median(e_sdf):.2e}, 99.9 % {np.quantile(e_sdf, 0.999):.2e}, max {e_sdf.max():.2e}; albedo: median {np.median(e_alb):.2e}, 99.9 % {np.quantile(e_alb, 0.999):.2e}, max {e_alb.max():.2e}; "
          f"intrinsics {d_intr:.2e} relative, poses {d_pose:.2e}; voxels {len(dflt['keys'])} / {len(det['keys'])}")
    assert np.median(e_sdf) <= 1e-3 and np.median(e_alb) <= 1e-3
    assert np.quantile(e_sdf, 0.999) <= 1e-2 and np.quantile(e_alb, 0.999) <= 1e-2
    assert e_sdf.max() <= 4e-2 and e_alb.max() <= 4e-2
    assert d_intr <= 1e-3 and d_pose <= 5e-3
