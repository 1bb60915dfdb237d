#!/usr/bin/env python3
"""Dev tool (GPU box): prints the oracle-vs-device numbers behind tests/test_gpu_bench_parity.py without asserting."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_bench_parity as T
from oracle import oracle_py as O

O.build(); t0 = time.time()
S = T.build_slice(O)
print(f"setup {time.time() - t0:.1f}s, voxels {len(S['arrays']['sdf'])}", flush=True)
for cg in [int(a) for a in sys.argv[1:]] or [30, -1]:
    t0 = time.time()
    runs = T._run_both(S, cg)
    print(f"--- cg_fixed {cg}  ({time.time() - t0:.1f}s)")
    for k, (ref, ocam, so, dev, dcam, sg, start) in enumerate(runs):
        print(f" it{k} rows {list(so.rows)} / {list(sg.rows)}  accepted {list(so.accepted[:so.n_attempts])} / {list(sg.step_accepted[:sg.num_attempts])}  cg {list(so.cg_iters[:so.n_attempts])} / {list(sg.pcg_iterations[:sg.num_attempts])}")
        print(f"     cost {so.cost_initial:.9e}->{so.cost_final:.9e} / {sg.cost_initial:.9e}->{sg.cost_final:.9e}  rel {abs(so.cost_final - sg.cost_final) / so.cost_final:.2e}")
        sdf, alb = dev; intr, dist, poses = ocam; gi, gd, gp = dcam
        d = np.abs(sdf - ref["sdf_refined"]); a = np.abs(alb - ref["albedo"])
        us = np.abs(ref["sdf_refined"] - start["sdf_refined"]).max(); ua = np.abs(ref["albedo"] - start["albedo"]).max()
        print(f"     sdf max rel {d.max() / np.abs(ref['sdf_refined']).max():.3e}  albedo max rel {a.max() / np.abs(ref['albedo']).max():.3e}  error / update: sdf {d.max() / us:.3e} albedo {a.max() / ua:.3e}  "
              f"intr rel {np.abs(gi - intr).max() / np.abs(intr).max():.3e}  poses abs {np.abs(gp - poses).max():.3e}")
