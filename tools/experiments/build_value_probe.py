"""GPU box: k_build<true> alone on the bench workload — the assembly (observation pass + build) repeated on one unchanged state through the debug entry point, HIP events
around the build launch.  Run once with the tree's library and once per probe library (I3D_LIB=gpurun_ab/lib_vprobe<W>.so, built with -DI3D_BUILD_VALUE_PROBE=<W>: the value half
of a two-kernel split, storing the 112 B per row a derivative kernel would need instead of assembling the partials).  Also times a streaming copy kernel of the bytes
the derivative half would have to move at least.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
import bench
import torch

def main():
    args = bench.parse_args()
    from intrinsic3d_amd import binding
    log = lambda m: None
    sc = bench.build_workload(args, log)
    thres = args.shell * float(sc["voxel_size"])
    arrays = bench.grid_arrays(sc)
    ctx = binding.Context(0)
    ctx.set_grid(sc["voxel_size"], arrays["keys"], arrays["sdf"], arrays["sdf_refined"], arrays["albedo"], arrays["weight"], arrays["color"])
    ctx.set_frames(sc["frames"], 1); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
    ctx.estimate_sh(args.subvolume, 10.0, thres)
    cfg = bench.make_cfg(binding, args, 1, thres)
    ctx.debug_assemble(cfg, 0); ctx.debug_assemble(cfg, 0)
    ctx.timing_enable(True); ctx.timing_select(["build", "observe"]); ctx.timing_get(reset=True)
    for _ in range(10):
        ctx.debug_assemble(cfg, 0)
    torch.cuda.synchronize()
    t = ctx.timing_get(reset=True)
    sizes = ctx.problem_sizes()
    out = {"lib": os.environ.get("I3D_LIB", "tree"), "build_ms": t["build"][0] / max(1, t["build"][1]), "build_launches": t["build"][1], "eg_rows": sizes["eg"], "active": sizes["active"]}
    # the least the derivative half would move: read 112 B per row + 208 B per voxel (the fp32 point records), write the 136 B of a stored row
    rd = 112 * sizes["eg"] + 208 * sizes["active"]; wr = 136 * sizes["eg"]
    n = (rd + wr) // 2 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_(); b = torch.empty_like(a)
    for _ in range(3): torch.mul(a, 1.0, out=b)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): torch.mul(a, 1.0, out=b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.update({"derivative_half_min_bytes": rd + wr, "streaming_copy_ms_for_those_bytes": ms, "streaming_copy_TBs": (rd + wr) / ms / 1e9})
    print(json.dumps(out))

main()
