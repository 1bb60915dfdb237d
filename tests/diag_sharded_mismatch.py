"""Diagnostic (not collected by pytest): the configuration of test_sharded_ranks_match_single_rank (2 iterations, 12 fixed PCG iterations, every group free) through the
oracle, the single-rank path in its modes, and W = 2 simulated ranks — cost per iteration, attempts, radius, field differences against the oracle.
    python tests/diag_sharded_mismatch.py"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from oracle import oracle_py as O
from intrinsic3d_amd import binding

O.build(); O.lib()
sc = helpers.small_scene()
g, fr, a0, vsh, thres = helpers.oracle_setup(O, sc)
ocfg = helpers.oracle_cfg(O, thres, iterations=int(os.environ.get("DIAG_ITERS", "2")), cg_fixed_iterations=int(os.environ.get("DIAG_CG", "12")))
cfg = helpers.gpu_cfg(ocfg)
g2 = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); g2.clear_outside_shell(thres)
g2.import_fields(sdf_refined=a0["sdf_refined"], albedo=a0["albedo"])
rc, intr, dist, poses, stats = O.optimize(g2, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh)
ref = g2.export()
print("oracle      ", [(f"{s.cost_initial:.9e}", f"{s.cost_final:.9e}", s.n_attempts, list(s.accepted[:s.n_attempts])) for s in stats])


def show(name, st, sdf, alb):
    print(f"{name:12s}", [(f"{s.cost_initial:.9e}", f"{s.cost_final:.9e}", s.num_attempts, list(s.step_accepted[:s.num_attempts]), list(s.pcg_iterations[:s.num_attempts]), f"{s.final_radius:.3e}") for s in st],
          f"sdf err {np.abs(sdf - ref['sdf_refined']).max() / np.abs(ref['sdf_refined']).max():.2e} alb err {np.abs(alb - ref['albedo']).max() / np.abs(ref['albedo']).max():.2e}", flush=True)


def single(name, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    c = helpers.gpu_context(sc, a0, vsh); st = c.optimize(cfg); sdf, alb = c.get_grid(); c.close()
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    show(name, st, sdf, alb)


single("default")
single("default again")
single("atomics", I3D_DETERMINISTIC="0")
single("serial det", I3D_LADDER="1")
single("serial atom", I3D_LADDER="1", I3D_DETERMINISTIC="0")
single("two-pass", I3D_GRADCOL="0", I3D_COST0="0")
single("two-pass atom", I3D_GRADCOL="0", I3D_COST0="0", I3D_DETERMINISTIC="0")
single("gradcol atom", I3D_DETERMINISTIC="0", I3D_COST0="0")
L = binding.load()
for W in (2,):
    for rep in range(2):
        shared = L.i3d_comm_sim_create(W)
        ctxs = [helpers.gpu_context(sc, a0, vsh) for _ in range(W)]
        for r, c in enumerate(ctxs): c.comm_init_sim(shared, r)
        out = [None] * W
        def run(r): out[r] = ctxs[r].optimize(cfg)
        th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
        [t.start() for t in th]; [t.join(timeout=120) for t in th]
        sdf, alb = ctxs[0].get_grid()
        show(f"sharded W={W}", out[0], sdf, alb)
        for c in ctxs: c.close()
        L.i3d_comm_sim_destroy(shared)
