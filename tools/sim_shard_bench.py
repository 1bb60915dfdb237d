#!/usr/bin/env python3
"""GPU box: the bench workload through the SPMD path with W ranks SIMULATED on one GPU (i3d_comm_init_sim: W host threads, W contexts,
host-mediated exchanges) — a capacity / correctness check of the sharding plan at full size (rim lists, ghost tiles, tile halos) and the
per-rank traffic log of a PCG pass; NOT a timing (the ranks share one device).

    python tools/sim_shard_bench.py --world 8 [--voxels 8e6] [--iterations 1]
"""
import argparse, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from intrinsic3d_amd import binding


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8); ap.add_argument("--iterations", type=int, default=1)
    ap.add_argument("--voxels", type=float, default=8.0e6); ap.add_argument("--frames", type=int, default=200)
    a = ap.parse_args()
    args = argparse.Namespace(voxels=a.voxels, frames=a.frames, width=640, height=480, voxel_size=0.001, band=3.5, shell=1.0, subvolume=0.06, seed=1234,
                              pcg_fixed=-1, carry_radius=False)
    log = lambda m: print(f"[sim] {m}", file=sys.stderr, flush=True)
    sc = bench.build_workload(args, log); thres = float(sc["voxel_size"]); arrays = bench.grid_arrays(sc)
    cfg = bench.make_cfg(binding, args, a.iterations, thres)

    def make():
        c = binding.Context(0)
        c.set_grid(sc["voxel_size"], arrays["keys"], arrays["sdf"], arrays["sdf_refined"], arrays["albedo"], arrays["weight"], arrays["color"])
        c.set_frames(sc["frames"], 1); c.set_camera(sc["intr"], sc["dist"], sc["poses"])
        return c
    ref = make(); sh, _, _ = ref.estimate_sh(args.subvolume, 10.0, thres); vsh = ref.get_voxel_sh()
    t0 = time.time(); rst = ref.optimize(cfg); log(f"single rank: {time.time() - t0:.2f}s, attempts {[s.num_attempts for s in rst]}")
    rsdf, ralb = ref.get_grid(); ref.close()
    W = a.world; L = binding.load()
    shared = L.i3d_comm_sim_create(W)
    ctxs = []
    for r in range(W):
        c = make(); c.set_voxel_sh(vsh); c.comm_init_sim(shared, r); ctxs.append(c)
    out = [None] * W; err = [None] * W

    def run(r):
        try:
            out[r] = ctxs[r].optimize(cfg)
        except Exception as e:
            err[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    t0 = time.time(); [t.start() for t in th]; [t.join(timeout=900) for t in th]
    assert not any(t.is_alive() for t in th), "sharded run hung"
    assert all(e is None for e in err), err
    log(f"{W} simulated ranks: {time.time() - t0:.1f}s")
    res = {"world": W, "active_voxels": int(ref_sizes(rst)), "ranks": []}
    for r, c in enumerate(ctxs):
        sdf, alb = c.get_grid(); st = c.comm_stats()
        e_sdf = float(np.abs(sdf - rsdf).max() / np.abs(rsdf).max()); e_alb = float(np.abs(alb - ralb).max() / np.abs(ralb).max())
        passes = max(1, st["halo_calls"])
        res["ranks"].append({"rank": r, "compute_list": st["compute_list"], "rim_entries_sent_per_pass": st["halo_send"], "rim_entries_received_per_pass": st["halo_recv"],
                             "rim_bytes_sent_per_pass": 8 * st["halo_send"], "ghost_tiles": st["ghost_tiles"], "operator_passes": passes,
                             "reduce_bytes_per_pass": st["reduce_bytes"] / passes, "max_rel_err_vs_single_rank": {"sdf": e_sdf, "albedo": e_alb},
                             "attempts": [int(s.num_attempts) for s in out[r]], "pcg": [[int(x) for x in s.pcg_iterations[:s.num_attempts]] for s in out[r]]})
        assert e_sdf <= 1e-4 and e_alb <= 1e-4, (r, e_sdf, e_alb)
        c.close()
    L.i3d_comm_sim_destroy(shared)
    print(json.dumps(res))


def ref_sizes(rst):
    return rst[0].valid_voxels


if __name__ == "__main__":
    main()
