"""-m gpu: the configurations BASELINE.json names, at test size, through the refine schedule (i3d_refine / apps/app_intrinsic3d) against
oracle.refine (Intrinsic3D::refine, intrinsic3d.cpp:206-350).  configs[0] (C1) is tests/test_gpu_parity.py::test_config_c1_dense_albedo_only.

  C2  single level at 4 mm, fixed camera (poses, intrinsics, distortion), ONE global SH volume
  C3  3 grid levels (4 -> 2 -> 1 mm) x (3, 1, 1) pyramid levels, poses fixed, joint SDF + albedo + spatially varying SH, on a dataset folder
      in the reference's layout through apps/app_intrinsic3d and through the same flow in-process
  C5  the same schedule with EVERY group free (poses, intrinsics, distortion: the shipped data/intrinsic3d.yml), noisy input poses, on a textured object:
      end to end (15 chained outer iterations) AND stage by stage from the oracle's state, both at 1e-4

Structure (keys, visit order, weights) must be identical; fields are held to 1e-4 on the 99.9 % quantile, and on the maximum to 1e-4 (C5) /
max(1e-4, helpers.ENVELOPE_FACTOR x the oracle's own sensitivity to 1e-7 input perturbations) (C2, C3: never needed so far, the numbers are printed)."""
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
    print(f"\n[schedule{', free camera' if free_camera else ''}] sdf: median {np.median(d_sdf) / smax:.2e}, 99.9 % {np.quantile(d_sdf, 0.999) / smax:.2e}, max {d_sdf.max() / smax:.2e} of max |sdf|; "
          f"oracle's own spread under 1e-7 input perturbations {env['sdf_refined'] / smax:.2e}; albedo: median {np.median(d_alb) / amax:.2e}, 99.9 % {np.quantile(d_alb, 0.999) / amax:.2e}, "
          f"max {d_alb.max() / amax:.2e}, spread {env['albedo'] / amax:.2e}; one-sided voxels under the perturbations {env.get('key_frac', 0.0):.2e}")
    assert np.quantile(d_sdf, 0.999) <= 1e-4 * smax, (np.quantile(d_sdf, 0.999), smax)
    assert np.quantile(d_alb, 0.999) <= 1e-4 * amax, (np.quantile(d_alb, 0.999), amax)
    if free_camera:      # C5 on the textured object: the plain bar on the maximum as well — no envelope (measured: sdf 5.3e-5, albedo 1.1e-5; the oracle's own spread
                         # under 1e-7 perturbations of its input is 7e-3 / 1e-3, i.e. the device is two orders closer to the oracle than the oracle is to its perturbed self)
        assert d_sdf.max() <= 1e-4 * smax, (d_sdf.max(), smax)
        assert d_alb.max() <= 1e-4 * amax, (d_alb.max(), amax)
    else:
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


def _three_level_schedule(oracle, tmp_path, *, seed, fix_poses, fix_distortion, iterations, pose_noise, subvolume, **scene_kw):
    """The reference's coarse-to-fine schedule on a dataset folder, through apps/app_intrinsic3d and in-process, against oracle.refine on the same decoded
    keyframes.  Returns what the callers assert on."""
    from intrinsic3d_amd import binding as B, synthetic
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_dataset
    app = os.path.join(ROOT, "apps", "app_intrinsic3d")
    assert os.path.exists(app), "apps/app_intrinsic3d has not been built (run __graft_entry__.build())"
    levels = 3
    sc = synthetic.make_scene(radius_vox=10, K=6, width=192, height=144, levels=1, seed=seed, pose_noise=pose_noise, lum_noise=0.003, cam_dist=0.2, **scene_kw)      # surface beyond sensor.yml's min_depth 0.1 m
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
    """BASELINE.json configs[4] at test size, END TO END: the FULL joint problem — SDF + albedo + spatially varying SH + poses + intrinsics + distortion, every switch of the
    shipped data/intrinsic3d.yml (fix_poses 0, fix_intrinsics 0, fix_distortion 0; 3 grid levels x (3, 1, 1) pyramid levels; subvolume_size_sh 0.2 m scaled to the
    8 cm object: 0.03 m; 3 outer iterations per stage) — from a dataset folder with noisy input poses (2 mm / 0.2 deg) through apps/app_intrinsic3d and in-process, against
    oracle.refine (Intrinsic3D::refine, refinement/intrinsic3d.cpp:229-290; Optimizer::optimize with the camera blocks free, optimizer.cpp:296-306), chained over all
    15 outer iterations WITHOUT re-seeding.  Bars: keys and visit order identical; sdf and albedo <= 1e-4 of the field maximum on the 99.9 % quantile AND on the maximum;
    poses <= 1e-4, intrinsics <= 1e-4 relative.  No envelope.  (Rounds 4-5 ran this on the untextured 4 cm sphere at fx 157, where the free poses are all but unconstrained and
    the oracle's own result moves 4e-2 under 1e-7 perturbations: only an envelope test was possible there.  On the textured object — C5_TEXTURE below — the problem is still
    sensitive (the oracle moves 7e-3 of max |sdf| under 1e-7 perturbations of its input poses, printed) but the device follows the oracle's trajectory: measured sdf
    median 5e-7 / 99.9 % 7e-6 / max 5.3e-5, albedo max 1.1e-5, poses 2.5e-7, intrinsics 4.3e-7.)  The test below holds every stage on its own, from the oracle's state."""
    monkeypatch.setenv("I3D_DETERMINISTIC", "1")     # (the default on one rank; inherited by the application's process)
    r = _three_level_schedule(oracle, tmp_path, seed=35, fix_poses=0, fix_distortion=0, iterations=3, pose_noise=(0.002, 0.0035), subvolume=0.03, **C5_TEXTURE)
    assert np.abs(r["poses"] - r["poses0"]).max() > 1e-3 and np.abs(r["intr"] - r["intr0"]).max() > 1e-2      # the camera did move
    with capsys.disabled():
        _check_fields(r["out"], r["ref"], r["env"], ordered=True, free_camera=True)
    d_pose = float(np.abs(r["poses"] - r["oposes"]).max()); d_intr = float(np.abs(r["intr"] - r["ointr"]).max() / np.abs(r["ointr"]).max())
    with capsys.disabled():
        print(f"\n[C5] poses: device vs oracle {d_pose:.3e} (oracle's own spread under 1e-7 input perturbations {r['env']['poses']:.3e}); intrinsics {d_intr:.3e} relative "
              f"(spread {r['env']['intr']:.3e}); pose update {np.abs(r['poses'] - r['poses0']).max():.3e}; one-sided voxels under the perturbations {r['env']['key_frac']:.2e}")
    assert d_intr <= 1e-4, d_intr
    assert d_pose <= 1e-4 * max(1.0, float(np.abs(r["oposes"]).max())), d_pose


# The C5 test object: the 4 cm sphere of the other schedule tests, but TEXTURED (albedo pattern of ~2.5 cm wavelength in three incommensurate directions, 1 cm bumps)
# and seen through a longer lens (fx 300: 60 px radius at level 0, 15 px on the coarsest pyramid level).  On the untextured sphere at fx 157 (8 px on the coarsest
# level) the free poses are all but unconstrained: the ORACLE moves them by 0.26 and 0.9 (rad / m) in the first two stages and its end result by 4e-2 of max |sdf|
# under 1e-7 input perturbations; on this one the first stage moves them 0.046, the later ones ~5e-3, and the end result 3e-3 (tools/c5_conditioning.py).
C5_TEXTURE = dict(albedo_freq=250.0, albedo_amp=0.3, bump_freq=120.0, fx=300.0)
C5_SCENE = dict(radius_vox=10, K=6, width=192, height=144, seed=35, pose_noise=(0.002, 0.0035), lum_noise=0.003, cam_dist=0.2, **C5_TEXTURE)


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


def test_config_c5_every_stage_of_the_free_camera_schedule_at_1e_4(oracle, monkeypatch, capsys):
    """BASELINE.json configs[4], the 1e-4 statement.  The free-camera schedule CHAINS 5 stages x 3 outer iterations, and the reference algorithm amplifies what it is
    handed: the oracle's own end result moves 2e-3 ... 4e-2 of max |sdf| when its input poses are perturbed by 1e-7 (tools/c5_conditioning.py, every scene tried), so two
    correct implementations cannot agree to 1e-4 at the END of the chain, and the end-to-end test above can only hold the device inside the oracle's own spread.  What
    CAN be held to 1e-4 is every link of the chain: the reference's schedule (Intrinsic3D::refine, intrinsic3d.cpp:229-290; thin shell :298-316; all pyramid levels on
    the coarsest grid only :236) is walked by the oracle, and every stage — (grid level, pyramid level) = (2,2) (2,1) (2,0) (1,0) (0,0), every group free
    (optimizer.cpp:296-306 with no flag set) — is ALSO run on the device from the oracle's state at the start of that stage:
      * thin shell on the device from the oracle's grid before it: keys in visit order and every field bit-exact;
      * lighting estimate + 2 chained outer iterations (Ceres' own PCG stop) + recolourisation: rows of every type, LM attempts, accept / reject sequence equal; PCG
        counts equal (+-1 on rejected attempts: fp32 vectors against the fp64 oracle at a stop threshold); sdf, albedo, poses, intrinsics <= 1e-4 in the max-norm
        convention of DESIGN.md section 6; colours within one count on >= 99.9 % of the components;
      * the first outer iteration of the stage again in the LDS-ATOMIC mode (I3D_DETERMINISTIC=0), with the damping ladder and through the serial loop (I3D_LADDER=1:
        every pass through k_eg_tile's fp32 LDS atomics — the path a sharded run takes): each against the oracle at 1e-4 and against the bit-reproducible default
        at <= 2e-6 (summation order only);
      * x2 upsampling on the device from the oracle's grid at the end of the level: keys in visit order and every field bit-exact."""
    from intrinsic3d_amd import binding as B, synthetic
    O = oracle
    GL = PL = 3
    sc = synthetic.make_scene(levels=PL, **C5_SCENE)
    G = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); Fr = O.Frames(sc["frames"], PL)
    intr, dist, poses = np.array(sc["intr"], np.float64), np.array(sc["dist"], np.float64), np.array(sc["poses"], np.float64)
    O.recompute_colors(G, Fr, intr, dist, poses, 0.02, 5)
    sub, lam_reg, iters = 0.03, 10.0, 2
    worst = dict(sdf=0.0, alb=0.0, pose=0.0, intr=0.0, mode=0.0); stages = []

    def seed(ctx, a, cam):
        ctx.set_grid(a["voxel_size"], a["keys"], a["sdf"], a["sdf_refined"], a["albedo"], a["weight"], a["color"]); ctx.set_camera(*cam)

    def same_grid(dev, ref, what):
        for k in ("keys", "weight", "color", "sdf", "sdf_refined", "albedo"):
            assert np.array_equal(dev[k], ref[k]), (what, k)

    with B.Context(0) as ctx:
        ctx.set_frames(sc["frames"], PL)
        for gl in range(GL - 1, -1, -1):
            vs = float(G.voxel_size)
            thres = float(O.lib().orc_varying_lambda(GL - 1 - gl, GL, 2.0, 1.0)) * vs
            before = dict(G.export(), voxel_size=vs)
            G.clear_outside_shell(thres)
            seed(ctx, before, (intr, dist, poses)); ctx.clear_outside_thin_shell(thres)
            same_grid(ctx.export_grid(), G.export(), f"thin shell at grid level {gl}")
            for pl in range(PL - 1, -1, -1):
                if pl > 0 and gl < GL - 1:
                    continue
                start = dict(G.export(), voxel_size=vs); cam0 = (intr.copy(), dist.copy(), poses.copy())
                rc_sh, _, _, vsh, _, _ = O.estimate_sh(G, sub, lam_reg, thres)
                assert rc_sh == 0
                ocfg = helpers.oracle_cfg(O, thres, iterations=iters, lm_steps=50, grid_level=gl, rgbd_level=pl)
                # the first outer iteration alone (for the mode comparison), then the stage proper
                o1 = helpers.oracle_cfg(O, thres, iterations=1, lm_steps=50, grid_level=gl, rgbd_level=pl,
                                        lambda_r0=ocfg.lambda_r0, lambda_s0=ocfg.lambda_s0, lambda_r1=ocfg.lambda_r0, lambda_s1=ocfg.lambda_s0)
                rc1, i1, d1, p1, st1 = O.optimize(G, Fr, o1, cam0[0], cam0[1], cam0[2], vsh); ref1 = G.export()
                assert rc1 == 0
                G.import_fields(sdf_refined=start["sdf_refined"], albedo=start["albedo"])                          # back to the start of the stage (optimize touches nothing else)
                first = {}
                for name, env in (("default", {}), ("lds_atomic", {"I3D_DETERMINISTIC": "0"}), ("lds_atomic_serial_loop", {"I3D_DETERMINISTIC": "0", "I3D_LADDER": "1"})):
                    with monkeypatch.context() as m:
                        for k, v in env.items():
                            m.setenv(k, v)
                        seed(ctx, start, cam0); ctx.estimate_sh(sub, lam_reg, thres)
                        g1 = ctx.optimize(helpers.gpu_cfg(o1))[0]
                    sdf, alb = ctx.get_grid(); ci, cd, cp = ctx.get_camera(); first[name] = (sdf, alb, ci, cp)
                    assert list(g1.rows) == list(st1[0].rows), (gl, pl, name)
                    assert list(g1.step_accepted[:g1.num_attempts]) == list(st1[0].accepted[:st1[0].n_attempts]), (gl, pl, name)
                    e = (_rel(sdf, ref1["sdf_refined"]), _rel(alb, ref1["albedo"]), _rel(ci, i1), float(np.abs(cp - p1).max() / max(1.0, np.abs(p1).max())))
                    assert max(e) <= 1e-4, (gl, pl, name, e)
                for name in ("lds_atomic", "lds_atomic_serial_loop"):
                    dm = max(_rel(first[name][0], first["default"][0]), _rel(first[name][1], first["default"][1]), _rel(first[name][2], first["default"][2]),
                             float(np.abs(first[name][3] - first["default"][3]).max()))
                    worst["mode"] = max(worst["mode"], dm)
                    assert dm <= 2e-6, (gl, pl, name, dm)
                # the stage: lighting + `iters` chained outer iterations + recolourisation, on both sides from the same state
                rc, intr, dist, poses, ost = O.optimize(G, Fr, ocfg, cam0[0], cam0[1], cam0[2], vsh)
                assert rc == 0
                O.recompute_colors(G, Fr, intr, dist, poses, 0.02, 5); ref = G.export()
                seed(ctx, start, cam0); ctx.estimate_sh(sub, lam_reg, thres)
                gst = ctx.optimize(helpers.gpu_cfg(ocfg)); ctx.recompute_colors(0.02, 5)
                out = ctx.export_grid(); ci, cd, cp = ctx.get_camera()
                for so, sg in zip(ost, gst):
                    assert list(so.rows) == list(sg.rows) and so.rows[0] > 0, (gl, pl)
                    assert list(so.accepted[:so.n_attempts]) == list(sg.step_accepted[:sg.num_attempts]), (gl, pl)
                    oc = list(so.cg_iters[:so.n_attempts]); gc = list(sg.pcg_iterations[:sg.num_attempts])
                    assert all(abs(x - y) <= 1 for x, y in zip(oc, gc)) and oc[-1] == gc[-1], (gl, pl, oc, gc)
                    assert abs(so.cost_final - sg.cost_final) <= 1e-4 * so.cost_final
                assert np.array_equal(out["keys"], ref["keys"])
                e = dict(sdf=_rel(out["sdf_refined"], ref["sdf_refined"]), alb=_rel(out["albedo"], ref["albedo"]), intr=_rel(ci, intr),
                         pose=float(np.abs(cp - poses).max() / max(1.0, np.abs(poses).max())))
                stages.append((gl, pl, len(ref["keys"]), [int(s.n_attempts) for s in ost], e))
                for k, v in e.items():
                    worst[k] = max(worst[k], v)
                assert max(e.values()) <= 1e-4, (gl, pl, e)
                cdiff = np.abs(out["color"].astype(int) - ref["color"].astype(int))
                assert (cdiff > 1).mean() <= 1e-3, (gl, pl, float((cdiff > 1).mean()))
                assert float(np.abs(poses - cam0[2]).max()) > 1e-6                  # the camera is free, and moves
            if gl > 0:
                end = dict(G.export(), voxel_size=vs)
                up = G.upsample(); G.free(); G = up
                seed(ctx, end, (intr, dist, poses)); ctx.upsample()
                same_grid(ctx.export_grid(), G.export(), f"upsampling from grid level {gl}")
    G.free(); Fr.free()
    with capsys.disabled():
        print("\n[C5, stage by stage] " + "; ".join(f"g{gl}p{pl}: {n} voxels, attempts {att}, sdf {e['sdf']:.1e} albedo {e['alb']:.1e} intrinsics {e['intr']:.1e} poses {e['pose']:.1e}" for gl, pl, n, att, e in stages)
              + f"; LDS-atomic modes against the default, one iteration: {worst['mode']:.1e}")
    assert [(gl, pl) for gl, pl, *_ in stages] == [(2, 2), (2, 1), (2, 0), (1, 0), (0, 0)]
