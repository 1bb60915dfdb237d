"""-m "not gpu": the N>1 protocol of the sharded PCG operator with real collectives (gloo, world_size 2 and 3, CPU).

Every rank holds the replicated problem structure, asks the library's host-side planner (i3d_shard_plan, the same inline
functions the device kernels use) for its owned work-list range, rank-major vector layout and compute list, evaluates
y = J^T W J x ONLY from the rows of its compute list (rows come from the CPU oracle), keeps ONLY the outputs of the unknowns it
owns, counts the camera columns ONLY on a row's owner, all-reduces the camera block and all-gathers the slices — exactly what
host/solver.cpp does around k_eg_pass / k_gather with RCCL.  The assembled result must equal the oracle's global product."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

OFFS = [(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1),(2,0,0),(0,2,0),(0,0,2),(1,1,0),(1,0,1),(0,1,1),
        (-2,0,0),(0,-2,0),(0,0,-2),(-1,-1,0),(-1,0,-1),(0,-1,-1)]
SDF_OFF = [(0,0,0),(0,1,0),(0,2,0),(0,1,1),(0,0,1),(0,0,2),(1,0,0),(1,1,0),(1,0,1),(2,0,0)]
ALB_OFF = [(0,0,0),(1,0,0),(0,1,0),(0,0,1)]
RING = [(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import helpers
    from oracle import oracle_py as O
    from intrinsic3d_amd import binding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.build()
        sc = helpers.small_scene(seed=4, radius_vox=9, K=3, width=96, height=72)
        g, fr, arrays, vsh, thres = helpers.oracle_setup(O, sc)
        cfg = helpers.oracle_cfg(O, thres)
        pv = O.ProblemView(g, fr, cfg, sc["intr"], sc["dist"], sc["poses"], vsh, 0)
        N = len(g); K = sc["K"]; NS = 6 * K + 9
        fl = pv.flags()
        active = fl["active"].astype(bool); free_s = ~fl["fix_sdf"].astype(bool); free_a = ~fl["fix_alb"].astype(bool)
        # replicated work list (any deterministic order) and its neighbour table in list space
        in_list = active | free_s | free_a
        wl = np.nonzero(in_list)[0]; A = wl.size
        lidx = -np.ones(N, np.int64); lidx[wl] = np.arange(A)
        keys = arrays["keys"]; index = {tuple(k): i for i, k in enumerate(keys.tolist())}
        anbr = -np.ones((18, A), np.int32)
        for j, o in enumerate(OFFS):
            nb = np.array([index.get((k[0] + o[0], k[1] + o[1], k[2] + o[2]), -1) for k in keys[wl].tolist()])
            anbr[j] = np.where(nb >= 0, lidx[np.maximum(nb, 0)], -1)
        chunk, own0, own1, comp = binding.shard_plan(A, world, rank, anbr, active[wl])
        L = binding.load()
        vs = lambda a: L.i3d_shard_vec_index(int(a), chunk, 0)
        va = lambda a: L.i3d_shard_vec_index(int(a), chunk, 1)
        tail = world * 2 * chunk
        # a random vector on the free unknowns, in the rank-major layout (replicated input, like the all-gathered u)
        rng = np.random.default_rng(7)
        xg = rng.normal(0, 1, 2 * N + NS)
        is_free = np.concatenate([free_s, free_a, np.full(6 * K, cfg.fix_poses == 0), np.full(4, cfg.fix_intrinsics == 0), np.full(5, cfg.fix_distortion == 0)])
        cost, grad, diag, touched = pv.normal_eq()
        xg = xg * is_free
        u = np.zeros(tail + NS)
        for a in range(A):
            u[vs(a)] = xg[wl[a]]; u[va(a)] = xg[N + wl[a]]
        u[tail:] = xg[2 * N:]

        def col_vec(v_idx):        # list-space vector position of voxel v_idx's sdf unknown (or None when outside the list = fixed)
            a = lidx[v_idx]
            return None if a < 0 else int(a)

        y_slice = np.zeros(tail + NS); cam = np.zeros(NS)
        owned = lambda a: own0 <= a < own1

        def add_row(centre_a, cols, coefs, w, cam_cols=None, cam_coefs=None):
            # cols: vector positions (or None); t = w * (J . u); outputs only on owned unknowns; camera only on the owner of the row
            d = sum(c * u[p] for p, c in zip(cols, coefs) if p is not None)
            if cam_cols is not None:
                d += sum(c * u[tail + p] for p, c in zip(cam_cols, cam_coefs))
            t = w * d
            for p, c, a_of in zip(cols, coefs, row_entries):
                if p is not None and owned(a_of):
                    y_slice[p] += c * t
            if cam_cols is not None and owned(centre_a):
                for p, c in zip(cam_cols, cam_coefs):
                    cam[p] += c * t

        v, f, w, r, J = pv.eg(True)
        for i in range(len(v)):
            ca = lidx[v[i]]
            if ca < 0 or not comp[ca]:
                continue
            k = keys[v[i]]
            ent = [lidx[index[(k[0] + o[0], k[1] + o[1], k[2] + o[2])]] for o in SDF_OFF] + [lidx[index[(k[0] + o[0], k[1] + o[1], k[2] + o[2])]] for o in ALB_OFF]
            cols = [None if e < 0 else vs(e) for e in ent[:10]] + [None if e < 0 else va(e) for e in ent[10:]]
            row_entries = ent
            cam_cols = list(range(6 * f[i], 6 * f[i] + 6)) + list(range(6 * K, 6 * K + 9))
            add_row(ca, cols, J[i, :14], w[i], cam_cols, J[i, 14:])
        for t_id, coefs_fn in ((1, None), (2, None), (3, None)):
            vv, dd, ww, rr = pv.reg(t_id)
            for i in range(len(vv)):
                ca = lidx[vv[i]]
                if ca < 0 or not comp[ca]:
                    continue
                k = keys[vv[i]]
                if t_id == 1:
                    ent = [ca] + [lidx[index[(k[0] + o[0], k[1] + o[1], k[2] + o[2])]] for o in RING]
                    cols = [None if e < 0 else vs(e) for e in ent]; coefs = [-6.0] + [1.0] * 6
                elif t_id == 2:
                    if rr[i] == 1e-7:      # Es row whose residual is exactly 0 has a zero Jacobian (surface_stab_regularizer.h:62-64)
                        continue
                    ent = [ca]; cols = [vs(ca)]; coefs = [1.0]
                else:
                    o = RING[dd[i]]; nb = lidx[index[(k[0] + o[0], k[1] + o[1], k[2] + o[2])]]
                    ent = [ca, nb]; cols = [va(ca), None if nb < 0 else va(nb)]; coefs = [1.0, -1.0]
                row_entries = ent
                add_row(ca, cols, coefs, ww[i])
        # fixed unknowns have no columns / outputs
        maskv = np.zeros(tail + NS)
        for a in range(A):
            maskv[vs(a)] = free_s[wl[a]]; maskv[va(a)] = free_a[wl[a]]
        maskv[tail:] = is_free[2 * N:]
        # the exchange: camera block all-reduced, slices all-gathered
        cam_t = torch.from_numpy(cam); dist.all_reduce(cam_t)
        sl = torch.from_numpy(y_slice[rank * 2 * chunk:(rank + 1) * 2 * chunk].copy())
        parts = [torch.zeros_like(sl) for _ in range(world)]
        dist.all_gather(parts, sl)
        y = np.concatenate([p.numpy() for p in parts] + [cam_t.numpy()]) * maskv
        # reference: the oracle's global product (u already carries zeros on fixed unknowns because x was masked)
        yref_g = pv.jtj_apply(xg)
        yref = np.zeros(tail + NS)
        for a in range(A):
            yref[vs(a)] = yref_g[wl[a]]; yref[va(a)] = yref_g[N + wl[a]]
        yref[tail:] = yref_g[2 * N:]
        err = np.abs(y - yref).max() / (np.abs(yref).max() + 1e-30)
        q.put((rank, float(err), int(comp.sum()), int(own1 - own0), A))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_operator_protocol_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(world))
    A = res[0][4]
    assert sum(r[3] for r in res) == A                       # owned ranges partition the work list
    for rank, err, ncomp, nown, _ in res:
        assert err < 1e-12, (rank, err)
        assert nown <= ncomp < A                              # compute list = owned + a halo, not everything
