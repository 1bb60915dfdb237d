"""One rank of a real multi-process sharded run (one process per GPU, RCCL communicator of the library, mailbox transport when its start-up test passes).
Launched by tests/test_gpu_multi_device.py through torch.distributed.run; only the launcher-side gloo group is used here (to hand out the unique id).
   argv: <out.npz> <iterations> <cg_fixed_iterations>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    import helpers
    from oracle import oracle_py as O
    from intrinsic3d_amd import binding
    out_path, iterations, cg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    O.build()
    sc = helpers.small_scene(seed=21, radius_vox=22, K=5, width=128, height=96)      # the work list spans a few dozen ownership tiles
    g, fr, arrays, vsh, thres = helpers.oracle_setup(O, sc)
    cfg = helpers.gpu_cfg(helpers.oracle_cfg(O, thres, iterations=iterations, cg_fixed_iterations=cg))
    ctx = binding.Context(local)
    ctx.set_grid(sc["voxel_size"], arrays["keys"], arrays["sdf"], arrays["sdf_refined"], arrays["albedo"], arrays["weight"], arrays["color"])
    ctx.set_frames(sc["frames"], sc["levels"]); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"]); ctx.set_voxel_sh(vsh)
    if world > 1:
        box = [binding.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(rank, world, box[0])
    st = ctx.optimize(cfg)
    sdf, alb = ctx.get_grid(); intr, dist5, poses = ctx.get_camera()
    stats = ctx.comm_stats() if world > 1 else {}
    np.savez(out_path + f".rank{rank}.npz", sdf=sdf, alb=alb, intr=intr, dist=dist5, poses=poses, rows=np.array([list(s.rows) for s in st]),
             cost=np.array([[s.cost_initial, s.cost_final] for s in st]), accepted=np.array([list(s.step_accepted[:s.num_attempts]) + [-1] * (50 - s.num_attempts) for s in st]),
             transport=np.array(ctx.comm_transport() if world > 1 else ""), halo_calls=np.array(stats.get("halo_calls", 0)), reduce_calls=np.array(stats.get("reduce_calls", 0)))
    ctx.close(); g.free(); fr.free()
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
