#!/usr/bin/env python3
"""Beyond the committed seeds: the oracle against the reference's compiled classes (oracle/_ref, needs /root/reference) on many more random scenes —
row assembly + level transitions (tests/test_ref_pipeline.py's own checks with other seeds), whole optimisations with varied sizes / parameter groups / observation
counts / weights, and fusion with varied depth ranges, clip boxes, erosion windows and correction sweeps.  Prints what differs (nothing, when last run: round 3)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from oracle import oracle_py as O, ref_py  # noqa: E402
import helpers  # noqa: E402
import test_ref_pipeline as T  # noqa: E402


def main(n=24):
    ref_py.build(); R = ref_py.pipeline(); fails = []; t0 = time.time()
    for seed in range(10, 10 + n):
        for name, fn in (("rows", T.test_row_assembly_equals_the_reference_code), ("levels", T.test_level_transitions_lighting_and_recolouring_equal_the_reference_code)):
            try:
                fn(O, R, seed)
            except AssertionError as e:
                fails.append((name, seed, str(e)[:200]))
    for seed in range(20, 20 + 2 * n):
        sc = T._mangled_scene(seed, radius_vox=int(7 + seed % 5), K=int(3 + seed % 3))
        (go, fo, ao, vsh, thres), (gr, fr, _, _, _) = T._both(O, R, sc)
        kw = [dict(), dict(fix_intrinsics=1), dict(num_observations=2), dict(lambda_g=1.0, lambda_r0=5.0)][seed % 4]
        _, io, do, po, so = O.optimize(go, fo, helpers.oracle_cfg(O, thres, iterations=2, **kw), sc["intr"], sc["dist"], sc["poses"], vsh)
        _, ir, dr, pr, sr = R.optimize(gr, fr, helpers.oracle_cfg(R, thres, iterations=2, **kw), sc["intr"], sc["dist"], sc["poses"], vsh)
        for a, b in zip(so, sr):
            same = (list(a.rows) == list(b.rows) and a.n_attempts == b.n_attempts and list(a.cg_iters[:a.n_attempts]) == list(b.cg_iters[:b.n_attempts])
                    and list(a.accepted[:a.n_attempts]) == list(b.accepted[:b.n_attempts]) and abs(a.cost_final - b.cost_final) <= 1e-11 * a.cost_final)
            if not same:
                fails.append(("optimize", seed, "statistics"))
        eo = go.export(); er = gr.export(); moved = np.abs(eo["sdf_refined"] - ao["sdf_refined"]).max()
        if np.abs(eo["sdf_refined"] - er["sdf_refined"]).max() > 1e-8 * moved or np.abs(po - pr).max() > 1e-9:
            fails.append(("optimize", seed, "fields"))
        for h in (go, gr, fo, fr):
            h.free()
    for seed in range(20, 20 + n):
        sc, frames = T._fusion_frames(seed, True)
        intr = sc["intr"].astype(np.float32); cintr = intr * np.float32(0.5)
        c = np.asarray(sc["keys"], np.float64).mean(0) * float(sc["voxel_size"])
        clip = None if seed % 2 else (np.array([-0.02, 0.03, -1, 1, -1, 1]) + c[[0, 0, 1, 1, 2, 2]]).astype(np.float32)
        out = []
        for M in (O, R):
            f = M.Fusion(sc["voxel_size"], 0.1 + 0.05 * (seed % 3), 10.0 if seed % 3 else 0.9, clip)
            for d, bgr, Tm in frames:
                f.integrate(d, intr, bgr[::2, ::2].copy(), cintr, Tm, seed % 4)
            raw = f.export(); f.finish(2 + seed % 9); out.append((raw, f.export()))
        for x, y in zip(out[0], out[1]):
            for k in ("keys", "sdf", "weight", "color"):
                if not np.array_equal(x[k], y[k]):
                    fails.append(("fusion", seed, k))
    print(f"{4 * n} scenes + {n} fused volumes in {time.time() - t0:.0f} s; differences: {fails if fails else 'none'}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 24))
