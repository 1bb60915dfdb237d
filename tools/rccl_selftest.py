#!/usr/bin/env python3
"""Sanity check of the RCCL bootstrap on whatever GPUs are visible: creates a 1-rank communicator through the C ABI (the same calls
bench.py makes per rank) and runs a short optimize with it attached."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from intrinsic3d_amd import binding, synthetic
sc = synthetic.make_scene(radius_vox=14, voxel_size=0.004, K=4, width=160, height=120, seed=3)
n = sc["keys"].shape[0]
ctx = binding.Context(0)
uid = binding.Context.comm_unique_id(); print("unique id bytes", len(uid))
ctx.comm_init(0, 1, uid); print("ncclCommInitRank ok (world 1)")
sdf = sc["sdf"].astype(np.float64)
ctx.set_grid(sc["voxel_size"], sc["keys"], sdf, sdf, np.full(n, 0.6), sc["weight"], sc["color"])
ctx.set_frames(sc["frames"], 1); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
ctx.set_voxel_sh(np.tile(np.asarray(sc["scene"].sh), (n, 1)))
st = ctx.optimize(binding.default_config(iterations=1, thres_shell=2 * 0.004))
print("optimize ok", list(st[0].rows), st[0].cost_initial, st[0].cost_final)
ctx.close()
