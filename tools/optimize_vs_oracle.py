"""GPU box: ONE outer iteration of Optimizer::optimize (every parameter group free, Ceres' own PCG stopping rule) on the device against the oracle over a sweep of
scenes — camera distance, focal length, lens distortion, image noise, surface roughness, coarse pyramid level, few / many keyframes: LM attempts, accept / reject
sequence, PCG iteration counts, cost, fields.  One iteration on purpose: it isolates the arithmetic of a solve from the amplification of round-off by later iterations."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, helpers
from oracle import oracle_py as O
from intrinsic3d_amd import binding
O.build(); O.lib()
DIST = np.array([0.08, -0.03, 0.002, 0.003, -0.002])
SCENES = [("default", dict(), None, {}), ("close-up", dict(seed=7, fx=300.0, cam_dist=0.17), None, {}), ("zoomed", dict(seed=7, fx=300.0, cam_dist=0.45), None, {}),
          ("far", dict(seed=9, cam_dist=0.8), None, {}), ("distorted", dict(seed=10), DIST, {}), ("strongly distorted close-up", dict(seed=11, fx=260.0, cam_dist=0.2), 3.0 * DIST, {}),
          ("noisy images", dict(seed=12, lum_noise=0.02), None, {}), ("rough surface", dict(seed=13, bump_amp_vox=1.5, bump_freq=80.0), None, {}),
          ("smooth surface", dict(seed=14, bump_amp_vox=0.05), None, {}), ("coarse pyramid level", dict(seed=15, levels=2), None, dict(rgbd_level=1)),
          ("two keyframes", dict(seed=16, K=2), None, {}), ("forty keyframes, 8 observations", dict(seed=17, K=40, width=80, height=60), None, dict(num_observations=8)),
          ("no occlusion test", dict(seed=18), None, dict(occlusion_distance=0.0)), ("camera fixed", dict(seed=19), None, dict(fix_poses=1, fix_intrinsics=1, fix_distortion=1))]
worst = 0.0
for name, kw, dist, ckw in SCENES:
    sc = helpers.small_scene(**kw)
    if dist is not None: sc["dist"] = np.asarray(dist, np.float64)
    thres = 2.0 * float(sc["voxel_size"])
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); fr = O.Frames(sc["frames"], sc["levels"])
    g.clear_outside_shell(thres)
    rc, _, _, vsh, _, _ = O.estimate_sh(g, 0.05, 10.0, thres)
    a0 = g.export()
    ocfg = helpers.oracle_cfg(O, thres, iterations=1, **ckw)
    rc, ointr, odist, oposes, ostats = O.optimize(g, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh)
    ref = g.export()
    with binding.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        ctx.set_frames(sc["frames"], sc["levels"]); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        ctx.clear_outside_thin_shell(thres); ctx.estimate_sh(0.05, 10.0, thres)
        gst = ctx.optimize(helpers.gpu_cfg(ocfg))
        out = ctx.export_grid(); gi, gd, gp = ctx.get_camera()
    so, sg = ostats[0], gst[0]
    same = list(so.rows) == list(sg.rows) and list(so.accepted[:so.n_attempts]) == list(sg.step_accepted[:sg.num_attempts])
    pcg_o = list(so.cg_iters[:so.n_attempts]); pcg_g = list(sg.pcg_iterations[:sg.num_attempts])
    smax = np.abs(ref["sdf_refined"]).max()
    es = np.abs(out["sdf_refined"] - ref["sdf_refined"]).max() / smax; ea = np.abs(out["albedo"] - ref["albedo"]).max() / np.abs(ref["albedo"]).max()
    ep = np.abs(np.asarray(gp).ravel() - np.asarray(oposes).ravel()).max(); ei = np.abs(np.asarray(gi) / np.asarray(ointr) - 1).max()
    worst = max(worst, es, ea)
    print("%-32s voxels %6d rows %6d  attempts %d/%d same sequence %s  pcg %s/%s  cost rel %.1e  sdf %.1e (update %.1e) albedo %.1e poses %.1e intrinsics %.1e" % (
        name, len(ref["keys"]), so.rows[0], so.n_attempts, sg.num_attempts, same, pcg_o, pcg_g, abs(so.cost_final - sg.cost_final) / so.cost_final, es,
        np.abs(ref["sdf_refined"] - a0["sdf_refined"]).max() / smax, ea, ep, ei))
    g.free(); fr.free()
print("worst field error over the sweep (relative to the field's largest value; bar 1e-4): %.2e" % worst)
