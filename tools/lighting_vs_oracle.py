"""GPU box: LightingSVSH::estimate + per-voxel interpolation on the device (fp64 MFMA Gram blocks, LM on the host) against the oracle over a sweep of scenes, subvolume
sizes and regulariser weights: subvolume set, row counts, LM iterations, SH coefficients per subvolume and per voxel."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, helpers
from oracle import oracle_py as O
from intrinsic3d_amd import binding
O.build(); O.lib()
worst = 0.0
for name, kw in (("default", dict()), ("rough surface", dict(seed=13, bump_amp_vox=1.5, bump_freq=80.0)), ("large object", dict(seed=21, radius_vox=30)), ("tiny object", dict(seed=31, radius_vox=3, band_vox=1.6, K=4, width=64, height=48)),
                 ("untinted", dict(seed=22, tint=False))):
    sc = helpers.small_scene(**kw)
    thres = 2.0 * float(sc["voxel_size"])
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    g.clear_outside_shell(thres)
    with binding.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        ctx.set_frames(sc["frames"], sc["levels"]); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        ctx.clear_outside_thin_shell(thres)
        for size in (0.012, 0.03, 0.08, 10.0):
            for lam in (0.0, 10.0, 1e3):
                rc, sh, idx, vsh, has, st = O.estimate_sh(g, size, lam, thres)
                gsh, gidx, gst = ctx.estimate_sh(size, lam, thres)
                counts = (gst.subvolumes, gst.data_rows, gst.reg_rows, gst.lm_iterations) == (st.subvolumes, st.data_rows, st.reg_rows, st.lm_iterations)
                order = {tuple(k): i for i, k in enumerate(gidx.tolist())}
                ok_set = len(order) == len(idx) and all(tuple(k) in order for k in idx.tolist())
                e_sub = e_vox = float("nan")
                if ok_set:
                    perm = np.array([order[tuple(k)] for k in idx.tolist()], int)
                    scale = np.abs(sh).max() + 1e-30
                    e_sub = np.abs(gsh[perm] - sh).max() / scale
                    gv = ctx.get_voxel_sh(); m = has.astype(bool)
                    e_vox = np.abs(gv[m] - vsh[m]).max() / (np.abs(vsh[m]).max() + 1e-30)
                    worst = max(worst, e_sub, e_vox)
                print("%-14s size %6.3f lambda %6.0f  subvolumes %4d rows %6d + %5d  LM iterations %d/%d  counts equal %s  same subvolume set %s  SH err (of the largest coefficient): per subvolume %.1e per voxel %.1e" % (
                    name, size, lam, st.subvolumes, st.data_rows, st.reg_rows, st.lm_iterations, gst.lm_iterations, counts, ok_set, e_sub, e_vox))
    g.free()
print("worst SH error over the sweep: %.2e (bar 1e-4)" % worst)
