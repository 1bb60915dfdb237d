#!/usr/bin/env python3
"""Profiling driver: builds the bench workload (optionally smaller) and runs a few outer iterations so that rocprofv3
(--kernel-trace or --pmc passes) sees every kernel of the path.  Usage: python tools/run_phase.py [--voxels N] [--frames K] [--steps S]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from intrinsic3d_amd import binding

ap = argparse.ArgumentParser()
ap.add_argument("--voxels", type=float, default=8e6); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--steps", type=int, default=2); ap.add_argument("--pcg", type=int, default=-1)
a = ap.parse_args()
args = argparse.Namespace(voxels=a.voxels, frames=a.frames, width=640, height=480, voxel_size=0.001, band=3.5, shell=1.0, subvolume=0.2, seed=1234)
sc = bench.build_workload(args, lambda m: print(m, file=sys.stderr))
arr = bench.grid_arrays(sc)
ctx = binding.Context(0)
ctx.set_grid(sc["voxel_size"], arr["keys"], arr["sdf"], arr["sdf_refined"], arr["albedo"], arr["weight"], arr["color"])
ctx.set_frames(sc["frames"], 1); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
ctx.set_voxel_sh(np.tile(np.asarray(sc["scene"].sh), (arr["keys"].shape[0], 1)))
cfg = bench.make_cfg(binding, args, a.steps, args.shell * float(sc["voxel_size"]))
cfg.pcg_fixed_iterations = a.pcg
st = ctx.optimize(cfg)
print("sizes", ctx.problem_sizes(), "pcg", [int(s.pcg_iterations[i]) for s in st for i in range(s.num_attempts)])
ctx.close()
