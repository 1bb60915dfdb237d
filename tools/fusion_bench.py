#!/usr/bin/env python3
"""Throughput of the device TSDF fusion on the bench scene's frames (640x480, voxel 4 mm), with the CPU restatement timed on a few frames.

    python tools/fusion_bench.py --frames 40 --radius 302 [--cpu-frames 2]
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from intrinsic3d_amd import binding, synthetic
from make_dataset import pose_vec_to_cam_to_world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40); ap.add_argument("--radius", type=int, default=302)
    ap.add_argument("--width", type=int, default=640); ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cpu-frames", type=int, default=0); ap.add_argument("--correct", type=int, default=10)
    a = ap.parse_args()
    t0 = time.time()
    sc = synthetic.make_scene(radius_vox=a.radius, K=a.frames, width=a.width, height=a.height, levels=1, seed=1)
    intr = sc["intr"].astype(np.float32)
    frames = [(fr["depth"][0], fr["bgr"][0], pose_vec_to_cam_to_world(np.asarray(p, np.float64)).astype(np.float32)) for fr, p in zip(sc["frames"], sc["poses"])]
    print(f"[fusion_bench] {a.frames} frames {a.width}x{a.height} rendered in {time.time() - t0:.1f}s", file=sys.stderr)
    dmin, dmax = 0.1, 10.0
    with binding.Fusion(sc["voxel_size"], dmin, dmax, initial_capacity=1 << 25) as f:
        d, b, T = frames[0]; f.integrate(d, intr, b, intr, T, 2)                    # warm-up frame (module load, first allocations)
        t1 = time.time()
        for d, b, T in frames[1:]:
            f.integrate(d, intr, b, intr, T, 2)
        t2 = time.time()
        n = f.finish(a.correct)
        t3 = time.time()
        info = f.info()
    out = {"frames": a.frames, "image": [a.width, a.height], "voxel_size": float(sc["voxel_size"]), "ms_per_frame": 1e3 * (t2 - t1) / max(1, a.frames - 1),
           "finish_s": t3 - t2, "allocated": info["allocated"], "saved": n, "table_slots": info["capacity"], "correct_launches": info["correct_launches"]}
    if a.cpu_frames > 0:
        from oracle import oracle_py as O
        O.build()
        o = O.Fusion(sc["voxel_size"], dmin, dmax)
        t4 = time.time()
        for d, b, T in frames[:a.cpu_frames]:
            o.integrate(d, intr, b, intr, T, 2)
        out["cpu_ms_per_frame"] = 1e3 * (time.time() - t4) / a.cpu_frames; out["cpu_frames"] = a.cpu_frames
    print(json.dumps(out))


if __name__ == "__main__":
    main()
