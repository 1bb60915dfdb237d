"""Development aid: distribution of product-vs-oracle differences after a two-level refine (needs a GPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from oracle import oracle_py as O
from intrinsic3d_amd import binding
import test_gpu_levels as T

O.build()
big = len(sys.argv) > 4 and sys.argv[4] == "big"
sc = helpers.small_scene(seed=5, levels=2) if big else helpers.small_scene(seed=5, radius_vox=12, K=7, width=128, height=96, levels=2)
sc = dict(sc); sc["frames"] = T._color_frames(sc)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ocfg = helpers.oracle_cfg(O, 0.0, iterations=iters, lm_steps=20, fix_distortion=1, fix_intrinsics=fi, cg_fixed_iterations=int(sys.argv[3]) if len(sys.argv) > 3 else -1)
g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); fr = O.Frames(sc["frames"], sc["levels"])
stages = []
with binding.Context(0) as ctx:
    ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    ctx.set_frames(sc["frames"], sc["levels"]); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
    rc = binding.RefineConfig(num_grid_levels=2, num_rgbd_levels=2, thin_shell_factor=2.0, thin_shell_factor_final=1.0, clear_distant_voxels=1,
                              occlusion_distance=0.02, num_observations=5, subvolume_size_sh=0.05, sh_lambda_reg=10.0)
    ctx.refine(rc, helpers.gpu_cfg(ocfg), callback=lambda gl, ng, pl, npl: stages.append(ctx.export_grid()))
    out = ctx.export_grid(); cam = ctx.get_camera()
rcode, ointr, odist, oposes, done = O.refine(g, fr, ocfg, 2, 2, 2.0, 1.0, 1, 0.05, 10.0, sc["intr"], sc["dist"], sc["poses"])
ref = g.export()
d = np.abs(out["sdf_refined"] - ref["sdf_refined"]); a = np.abs(out["albedo"] - ref["albedo"])
print("N", len(d), "sdf diff quantiles 50/99/99.9/max", np.quantile(d, [0.5, 0.99, 0.999]), d.max(), "n>2e-7:", (d > 2e-7).sum())
print("albedo diff quantiles", np.quantile(a, [0.5, 0.99, 0.999]), a.max())
bad = np.argsort(d)[-10:]
print("worst keys", ref["keys"][bad].tolist(), d[bad])
print("intr", cam[0], ointr, "pose diff", np.abs(cam[2] - oposes).max())
