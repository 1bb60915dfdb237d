"""Generates tests/golden/optimize_small.json from the CPU oracle (run: python tests/golden/make_golden.py).

The reference has no fixtures and cannot be imported or built here, so these goldens pin the ORACLE'S OWN outputs
(parity unpinned): a change in the restatement, in libstdc++'s unordered_map iteration order or in the
synthetic generator shows up as a diff.  The -m gpu tests compare the HIP path against the same numbers."""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def compute(O):
    import helpers
    sc = helpers.small_scene(seed=11, radius_vox=12, K=4, width=128, height=96)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(O, sc, sh_size=0.04)
    rc, sh, idx, _, _, _ = O.estimate_sh(g, 0.04, 10.0, thres)
    cfg = helpers.oracle_cfg(O, thres, iterations=2, cg_fixed_iterations=4)
    rc, intr, dist, poses, stats = O.optimize(g, fr, cfg, sc["intr"], sc["dist"], sc["poses"], vsh)
    out = g.export()
    res = {
        "num_voxels": int(len(g)),
        "visit_order_crc": int(zlib.crc32(arrays["keys"].tobytes())),
        "rows": [list(map(int, s.rows)) for s in stats],
        "cost": [[s.cost_initial, s.cost_final] for s in stats],
        "sdf_sum": float(np.sum(out["sdf_refined"])), "albedo_sum": float(np.sum(out["albedo"])),
        "sdf_refined_head": out["sdf_refined"][:16].tolist(), "albedo_head": out["albedo"][:16].tolist(),
        "intr": intr.tolist(), "dist": dist.tolist(), "poses": poses.tolist(),
        "sh0": sh[0].tolist(),
    }
    g.free(); fr.free()
    return res


if __name__ == "__main__":
    from oracle import oracle_py as O
    O.build()
    r = compute(O)
    with open(os.path.join(HERE, "optimize_small.json"), "w") as f:
        json.dump(r, f, indent=1)
    print("written", r["num_voxels"], r["rows"])
