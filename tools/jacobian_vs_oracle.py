"""GPU box: per-column error of the device Jacobian against the oracle's duals, in units of the parity tolerance (1e-4 relative + 2e-6 of the column's largest entry, as
in tests/test_gpu_parity.py), the residuals and the row sets, over a sweep of scenes: camera distance, focal length, lens distortion, image noise, bump amplitude.
The close-up scene is what exposed the fp32 cancellation in the spline derivative weights (build.hip bicubic_eval, round 4)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, helpers
from oracle import oracle_py as oracle
oracle.build(); oracle.lib()
DIST = np.array([0.08, -0.03, 0.002, 0.003, -0.002])
SCENES = [("default", dict(), None), ("close-up", dict(seed=7, fx=300.0, cam_dist=0.17), None), ("zoomed", dict(seed=7, fx=300.0, cam_dist=0.45), None),
          ("very close", dict(seed=8, radius_vox=12, fx=200.0, cam_dist=0.09), None), ("far", dict(seed=9, cam_dist=0.8), None),
          ("distorted", dict(seed=10), DIST), ("strongly distorted close-up", dict(seed=11, fx=260.0, cam_dist=0.2), 3.0 * DIST),
          ("noisy images", dict(seed=12, lum_noise=0.02), None), ("rough surface", dict(seed=13, bump_amp_vox=1.5, bump_freq=80.0), None),
          ("smooth surface", dict(seed=14, bump_amp_vox=0.05), None), ("coarse pyramid level", dict(seed=15, levels=2), None)]
worst = 0.0
for name, kw, dist in SCENES:
    sc = helpers.small_scene(**kw)
    if dist is not None: sc["dist"] = np.asarray(dist, np.float64)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    ocfg = helpers.oracle_cfg(oracle, thres, cg_fixed_iterations=5, iterations=1, rgbd_level=1 if kw.get("levels", 1) > 1 else 0)
    pv = oracle.ProblemView(g, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh, 0)
    ctx = helpers.gpu_context(sc, arrays, vsh)
    ctx.debug_assemble(helpers.gpu_cfg(ocfg), 0)
    v, f, w, r, J = pv.eg(with_jacobian=True)
    gfr, gw, gr, gJ = ctx.debug_eg_rows(jac=True)
    same_rows = set(zip(v.tolist(), f.tolist())) == set((int(i), int(gfr[i, k])) for i, k in zip(*np.nonzero(gfr >= 0)))
    if not same_rows or len(v) == 0:
        print("%-28s rows %6d ROW SETS DIFFER" % (name, len(v))); pv.free(); ctx.close(); continue
    slot = np.array([int(np.nonzero(gfr[vi] == fi)[0][0]) for vi, fi in zip(v, f)])
    Jg = gJ[v, slot]
    colmax = np.abs(J).max(axis=0, keepdims=True)
    tol = 1e-4 * np.abs(J) + 2e-6 * colmax
    q = (np.abs(Jg - J) / tol).max(axis=0)
    rr = np.abs(gr[v, slot] - r) / (1e-4 * np.abs(r) + 1e-9)
    cost, gvec, dg, free = pv.normal_eq(); gg, gd, gcost = ctx.debug_normal_eq()
    worst = max(worst, q.max(), rr.max())
    print("%-28s rows %6d  residual err/tol %.3f  weight rel %.1e  Jacobian err/tol by group: sdf %.2f albedo %.2f pose %.2f intr %.2f dist %.2f  cost rel %.1e  grad rel %.1e" % (
        name, len(v), rr.max(), np.abs(gw[v, slot] / w - 1).max(), q[:10].max(), q[10:14].max(), q[14:20].max(), q[20:24].max(), q[24:29].max(),
        abs(gcost - cost) / cost, np.abs(gg - gvec).max() / np.abs(gvec).max()))
    pv.free(); ctx.close()
print("worst err/tol over the sweep: %.3f" % worst)
