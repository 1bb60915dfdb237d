#!/usr/bin/env python3
"""Full-size parity artefact for the headline workload ("missing" 3): BASELINE.json configs[3] on one GPU — all 8.02 M stored voxels, 200 keyframes,
every parameter group free — two Gauss-Newton iterations on the device AND on the CPU oracle (the restated reference, fp64), from identical inputs:

    rows of every type, LM attempts, PCG iteration counts of every attempt, cost before / after, fields (max-norm relative and relative L2 of the update), camera.

The CPU leg is the whole workload, so its time is the UNEXTRAPOLATED `cpu_baseline` of the bench (bench.py times a 1 M-voxel cap and scales it).  Takes ~5-8 minutes of
host time on the GPU box (residual collection on one thread as in the reference, solve on 8 threads).  Writes one JSON object (stdout, or --out).

    python tools/c4_full_parity.py --out gpurun_out/r05_c4_full_parity.json        # then copy into profiles/
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--voxels", type=float, default=8.0e6)
    ap.add_argument("--band", type=float, default=3.5)
    ap.add_argument("--threaded-leg", action="store_true", help="also time one iteration with the residual collection threaded")
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse_args()
    args.voxels = a.voxels; args.band = a.band
    args.cpu_sample = 4.0 * a.voxels          # >= every stored voxel: the sample IS the workload
    args.cpu_ref_sample = 0

    def log(m):
        print(f"[c4_full_parity] {m}", file=sys.stderr, flush=True)
    t0 = time.time()
    sc = bench.build_workload(args, log)
    thres = args.shell * float(sc["voxel_size"])
    cpu = bench.cpu_baseline(args, sc, thres, log, device=0, threaded_leg=a.threaded_leg)
    if cpu is None:
        raise SystemExit("the CPU leg failed")
    out = {"what": "two Gauss-Newton iterations of the headline workload at FULL size, device vs CPU oracle, from identical inputs",
           "stored_voxels": int(sc["keys"].shape[0]), "keyframes": int(sc["K"]), "band": a.band, "wall_s": time.time() - t0, "cpu_baseline_full_size": cpu}
    txt = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(txt)
    p = cpu.get("parity_on_sample") or {}
    ok = (p.get("rows_equal") and p.get("lm_attempts", {}).get("oracle") == p.get("lm_attempts", {}).get("device") and p.get("sdf_max_rel_err", 1.0) <= 1e-4 and p.get("albedo_max_rel_err", 1.0) <= 1e-4)
    log(f"full-size parity {'OK' if ok else 'NOT within the bar'}: {json.dumps({k: p.get(k) for k in ('rows_equal', 'lm_attempts', 'pcg_iterations', 'sdf_max_rel_err', 'albedo_max_rel_err', 'poses_max_abs_err', 'intrinsics_max_rel_err')})}")
    return 0 if ok else 2


if __name__ == "__main__":
    sys.exit(main())
