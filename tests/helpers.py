"""Shared scene set-up for the parity tests: synthetic scene -> oracle grid (reference visit order) -> flat arrays."""
import numpy as np

from intrinsic3d_amd import synthetic


# The refine schedule chains lighting estimates, joint solves and 8-bit recolourisations over several levels; where the joint geometry + camera
# problem is so ill-conditioned that the ORACLE ITSELF moves further than 1e-4 when its input poses are perturbed by a few 1e-7 (gauge
# freedom), fields are held to ENVELOPE_FACTOR x that measured sensitivity of the reference computation instead.  One factor, used by every
# schedule test (test_gpu_levels.py, test_gpu_configs.py) and quoted in DESIGN.md section 6: the device path adds run-to-run summation-order
# noise of its own (fp32 atomics), and an envelope from a handful of perturbed re-runs under-estimates the true spread.
ENVELOPE_FACTOR = 10.0


def small_scene(seed=1, radius_vox=16, K=6, width=160, height=120, levels=1, **kw):
    return synthetic.make_scene(radius_vox=radius_vox, voxel_size=0.004, K=K, width=width, height=height, levels=levels, seed=seed, **kw)


def oracle_setup(O, sc, thres_factor=2.0, sh_size=0.05, perturb=True, seed=3):
    """Returns (grid, frames, arrays, voxel_sh, thres).  Runs the reference's per-level preparation on the oracle:
    clearVoxelsOutsideThinShell (intrinsic3d.cpp:298-316) and the SVSH estimate (intrinsic3d.cpp:255-264)."""
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    fr = O.Frames(sc["frames"], sc["levels"])
    thres = thres_factor * float(sc["voxel_size"])
    g.clear_outside_shell(thres)
    if perturb:      # make sdf_refined != sdf and albedo non-constant so every residual type has a non-trivial Jacobian
        a = g.export()
        rng = np.random.default_rng(seed)
        n = len(g)
        sr = a["sdf_refined"] + rng.normal(0, 0.02 * float(sc["voxel_size"]), n)
        al = 0.6 + 0.05 * np.sin(40.0 * a["keys"][:, 0] * float(sc["voxel_size"])) + rng.normal(0, 0.01, n)
        g.import_fields(sdf_refined=sr, albedo=al)
    rc, sh, idx, vsh, has, st = O.estimate_sh(g, sh_size, 10.0, thres)
    assert rc == 0
    arrays = g.export()
    return g, fr, arrays, vsh, thres


def oracle_cfg(O, thres, **kw):
    d = dict(iterations=3, lm_steps=50, lambda_g=0.2, lambda_r0=80.0, lambda_r1=10.0, lambda_s0=120.0, lambda_s1=10.0, lambda_a=0.1,
             fix_poses=0, fix_intrinsics=0, fix_distortion=0, occlusion_distance=0.02, num_observations=5, thres_shell=thres,
             grid_level=0, rgbd_level=0, cg_fixed_iterations=-1, verbose=0, fix_sdf=0, carry_trust_radius=0)
    d.update(kw)
    return O.OptConfig(**d)


def gpu_cfg(ocfg):
    from intrinsic3d_amd import binding
    return binding.default_config(
        iterations=ocfg.iterations, lm_steps=ocfg.lm_steps, lambda_g=ocfg.lambda_g, lambda_r0=ocfg.lambda_r0, lambda_r1=ocfg.lambda_r1,
        lambda_s0=ocfg.lambda_s0, lambda_s1=ocfg.lambda_s1, lambda_a=ocfg.lambda_a, fix_poses=ocfg.fix_poses,
        fix_intrinsics=ocfg.fix_intrinsics, fix_distortion=ocfg.fix_distortion, occlusion_distance=ocfg.occlusion_distance,
        num_observations=ocfg.num_observations, thres_shell=ocfg.thres_shell, grid_level=ocfg.grid_level, rgbd_level=ocfg.rgbd_level,
        pcg_fixed_iterations=ocfg.cg_fixed_iterations, verbose=ocfg.verbose, fix_sdf=ocfg.fix_sdf, carry_trust_radius=ocfg.carry_trust_radius)


def gpu_context(sc, arrays, vsh):
    from intrinsic3d_amd import binding
    ctx = binding.Context(0)
    ctx.set_grid(sc["voxel_size"], arrays["keys"], arrays["sdf"], arrays["sdf_refined"], arrays["albedo"], arrays["weight"], arrays["color"])
    ctx.set_frames(sc["frames"], sc["levels"])
    ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
    ctx.set_voxel_sh(vsh)
    return ctx


def align_by_key(out, ref, max_frac=2e-4, ordered=True):
    """The level schedule re-sparsifies the grid from OPTIMISED values (|sdf_refined| > thres_shell and the sign tests of clearVoxelsOutsideThinShell,
    algorithms.cpp:376-440): a voxel within round-off of such a threshold may be kept on one side and dropped on the other.  The visit ORDER of what both
    keep must still be the reference's (erase keeps relative order; the children of an upsampling follow their parents), and only a handful of voxels may
    differ at all.  Returns (out, ref) restricted to the common keys (unchanged when the key arrays are equal)."""
    import numpy as np
    if np.array_equal(out["keys"], ref["keys"]):
        return out, ref
    def as_set(k): return set(map(tuple, k.tolist()))
    so, sr = as_set(out["keys"]), as_set(ref["keys"])
    assert len(so ^ sr) <= max(16, int(max_frac * len(sr))), (len(so - sr), len(sr - so), len(sr))
    common = so & sr
    mo = np.fromiter((tuple(k) in common for k in out["keys"].tolist()), bool, len(out["keys"])); mr = np.fromiter((tuple(k) in common for k in ref["keys"].tolist()), bool, len(ref["keys"]))
    out = {k: (v[mo] if getattr(v, "shape", ())[:1] == mo.shape else v) for k, v in out.items()}
    ref = {k: (v[mr] if getattr(v, "shape", ())[:1] == mr.shape else v) for k, v in ref.items()}
    if not ordered and not np.array_equal(out["keys"], ref["keys"]):
        # more than a handful of one-sided voxels: the iteration order of the reference's unordered_map depends on EVERY insertion (bucket counts, rehash
        # points), so the common voxels need not keep their relative order — compare by key
        def by_key(d):
            o = np.lexsort((d["keys"][:, 2], d["keys"][:, 1], d["keys"][:, 0])); n = len(o)
            return {k: (v[o] if getattr(v, "shape", ())[:1] == (n,) else v) for k, v in d.items()}
        out, ref = by_key(out), by_key(ref)
    assert np.array_equal(out["keys"], ref["keys"])                       # same relative visit order (ordered) / same voxels (by key)
    return out, ref


def axis_camera_dataset(root, seed=12, radius_vox=10, width=96, height=72):
    """A dataset folder in the reference's layout (rgbd/frame-XXXXXX.{color.png,depth.png,pose.txt} + the two intrinsics files) of the synthetic sphere seen by
    six cameras ON THE COORDINATE AXES: their rotations are signed permutations, so a 4x4 pose inverse is exact whatever its operation order (Eigen's
    is unpinned) and fusion results can be compared bit for bit between implementations.  Returns (folder, voxel_size, number of frames)."""
    import os
    import numpy as np
    from PIL import Image
    sc = synthetic.make_scene(radius_vox=radius_vox, K=1, width=width, height=height, levels=1, seed=seed)
    scene, center, intr = sc["scene"], sc["center"], sc["intr"]
    folder = os.path.join(str(root), "rgbd"); os.makedirs(folder); os.makedirs(os.path.join(str(root), "fusion"), exist_ok=True)
    K4 = np.eye(4); K4[0, 0], K4[1, 1], K4[0, 2], K4[1, 2] = intr
    np.savetxt(os.path.join(folder, "colorIntrinsics.txt"), K4); np.savetxt(os.path.join(folder, "depthIntrinsics.txt"), K4)
    rng = np.random.default_rng(seed); dist = 4.0 * radius_vox * float(sc["voxel_size"]); n = 0
    for axis in range(3):
        for sign in (1.0, -1.0):
            d = np.zeros(3); d[axis] = sign; eye = center + dist * d
            Rwc = np.round(synthetic.aa_to_rotmat(synthetic.look_at_pose(eye, center)[:3]))                       # signed permutation
            lum, depth, bgr = synthetic.render_frame(scene, np.concatenate([synthetic.rotmat_to_aa(Rwc), -Rwc @ eye]), intr, width, height, 0.0, rng)
            T = np.eye(4); T[:3, :3] = Rwc.T; T[:3, 3] = eye
            Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(os.path.join(folder, f"frame-{n:06d}.color.png"))
            Image.fromarray(np.round(depth * 1000.0).astype(np.uint16)).save(os.path.join(folder, f"frame-{n:06d}.depth.png"))
            np.savetxt(os.path.join(folder, f"frame-{n:06d}.pose.txt"), T); n += 1
    return folder, float(sc["voxel_size"]), n
