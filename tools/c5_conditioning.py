#!/usr/bin/env python3
"""Dev tool (CPU only, oracle only): how far does the ORACLE's own result of the free-camera 3-level schedule (BASELINE configs[4] at test size) move when its
input poses are perturbed by 1e-7?  Used in round 6 to pick a scene for tests/test_gpu_configs.py::test_config_c5 on which a 1e-4 statement is meaningful.
usage: python tools/c5_conditioning.py name=value ...   (make_scene keywords; fixi=1 / fixd=1 fix intrinsics / distortion; it=iterations)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from intrinsic3d_amd import synthetic
from oracle import oracle_py as O


def run(sc, ocfg, poses, gl, pl, sub):
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); fr = O.Frames(sc["frames"], pl)
    rc, intr, dist, p, done = O.refine(g, fr, ocfg, gl, pl, 2.0, 1.0, 1, sub, 10.0, sc["intr"], sc["dist"], poses)
    out = g.export(); g.free(); fr.free()
    return out, intr, dist, p


def main():
    kw = dict(radius_vox=10, K=6, width=192, height=144, levels=3, seed=35, pose_noise=(0.002, 0.0035), lum_noise=0.003, cam_dist=0.2)
    opt = dict(fixi=0, fixd=0, fixp=0, it=3, gl=3, sub=0.03, eps=1e-7)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        v = eval(v)
        (opt if k in opt else kw)[k] = v
    O.build()
    sc = synthetic.make_scene(**kw)
    ocfg = helpers.oracle_cfg(O, 0.0, iterations=opt["it"], lm_steps=50, fix_poses=opt["fixp"], fix_intrinsics=opt["fixi"], fix_distortion=opt["fixd"])
    t0 = time.time()
    ref, ri, rd, rp = run(sc, ocfg, sc["poses"], opt["gl"], kw["levels"], opt["sub"])
    t1 = time.time() - t0
    worst = dict(sdf=0.0, alb=0.0, pose=0.0, intr=0.0, keys=0.0, sdf_q999=0.0, alb_q999=0.0, sdf_n_over_1e5=0)
    for eps in (opt["eps"], -opt["eps"]):
        per, pi, pd, pp = run(sc, ocfg, sc["poses"] * (1.0 + eps), opt["gl"], kw["levels"], opt["sub"])
        if per["keys"].shape != ref["keys"].shape or not np.array_equal(per["keys"], ref["keys"]):
            so, sr = set(map(tuple, per["keys"].tolist())), set(map(tuple, ref["keys"].tolist()))
            worst["keys"] = max(worst["keys"], len(so ^ sr) / len(sr))
            a, b = helpers.align_by_key(dict(per), dict(ref), max_frac=1.0, ordered=False)
        else:
            a, b = per, ref
        worst["sdf"] = max(worst["sdf"], float(np.abs(a["sdf_refined"] - b["sdf_refined"]).max() / np.abs(b["sdf_refined"]).max()))
        e = np.abs(a["sdf_refined"] - b["sdf_refined"]) / np.abs(b["sdf_refined"]).max(); worst["sdf_q999"] = max(worst["sdf_q999"], float(np.quantile(e, 0.999))); worst["sdf_n_over_1e5"] = max(worst["sdf_n_over_1e5"], int((e > 1e-5).sum()))
        worst["alb_q999"] = max(worst["alb_q999"], float(np.quantile(np.abs(a["albedo"] - b["albedo"]), 0.999) / np.abs(b["albedo"]).max()))
        worst["alb"] = max(worst["alb"], float(np.abs(a["albedo"] - b["albedo"]).max() / np.abs(b["albedo"]).max()))
        worst["pose"] = max(worst["pose"], float(np.abs(pp - rp).max())); worst["intr"] = max(worst["intr"], float(np.abs(pi - ri).max() / np.abs(ri).max()))
    print(" ".join(sys.argv[1:]) or "(default)", f"| voxels {len(ref['keys'])} | {t1:.0f}s per refine | spread under {opt['eps']:g}: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()),
          f"| pose update {np.abs(rp - sc['poses']).max():.2e}", flush=True)


if __name__ == "__main__":
    main()
