"""GPU box: per-column error of the device Jacobian against the oracle's duals, in units of the parity tolerance, on three scenes (default, close-up, zoomed).
The close-up scene is what exposed the fp32 cancellation in the spline derivative weights (build.hip bicubic_eval, round 4)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, helpers
from intrinsic3d_amd import synthetic
from oracle import oracle_py as oracle
oracle.build(); oracle.lib()
for name, kw in (("default", dict()), ("zoom", dict(seed=7, fx=300.0, cam_dist=0.17)), ("zoom_far", dict(seed=7, fx=300.0, cam_dist=0.45))):
    sc = helpers.small_scene(**kw)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    ocfg = helpers.oracle_cfg(oracle, thres, cg_fixed_iterations=5, iterations=1)
    pv = oracle.ProblemView(g, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh, 0)
    ctx = helpers.gpu_context(sc, arrays, vsh)
    ctx.debug_assemble(helpers.gpu_cfg(ocfg), 0)
    v, f, w, r, J = pv.eg(with_jacobian=True)
    gfr, gw, gr, gJ = ctx.debug_eg_rows(jac=True)
    slot = np.array([int(np.nonzero(gfr[vi] == fi)[0][0]) for vi, fi in zip(v, f)])
    Jg = gJ[v, slot]
    colmax = np.abs(J).max(axis=0, keepdims=True)
    tol = 1e-4 * np.abs(J) + 2e-6 * colmax
    err = np.abs(Jg - J)
    bad = err > tol
    print(name, "rows", len(v), "bad entries", int(bad.sum()), "bad rows", int(bad.any(axis=1).sum()), "max err/tol by column", np.round((err / tol).max(axis=0), 2).tolist())
    print("  colmax", np.round(colmax[0], 4).tolist())
    rows = np.argwhere(bad.any(axis=1))[:3, 0]
    for i in rows:
        print("  row", i, "v", v[i], "f", f[i], "r", r[i], gr[v[i], slot[i]], "J", np.round(J[i, 14:29], 5).tolist(), "Jg", np.round(Jg[i, 14:29], 5).tolist())
    pv.free(); ctx.close()
