"""-m "not gpu": the N>1 protocol of the sharded PCG operator with real collectives (gloo, world_size 2 and 3, CPU).

Every rank holds the replicated problem structure and asks the library's host-side planner (i3d_shard_plan / i3d_shard_need, the same
inline functions the device plan uses) for its owned tile-aligned range, its compute list (owned + ghost entries) and the NEED sets.
Then one operator application exactly as host/solver.cpp + Comm run it:
  1. every rank holds the operator input u ONLY on the unknowns it owns — everything else is poisoned with NaN;
  2. the owners push the rim values their neighbours need (point-to-point: here all_to_all of (index, value) lists) — if the need
     sets were too small, a NaN would reach a row;
  3. y = J^T W J u from the rows of the compute list (rows come from the CPU oracle), kept ONLY on owned unknowns; the camera columns and
     p.q counted ONLY on a row's owner;
  4. ONE all-reduce of [camera block | p.q]; nothing else is exchanged — no vector is gathered (the check below assembles the owned
     segments only to compare with the oracle's global product).
Round 6, the ladder batch (solver.cpp pcg_solve_ladder, sharded): B = 3 systems iterate in lock step and share the exchanges of a pass — the rim message carries B values
per entry (Comm::push_halo_multi), the all-reduce carries [LADDER_MAX = 6][6K + 10] doubles whatever the number of live systems (the slots of stopped systems ride
along unused, so the ranks' collectives always match): the same checks per system, with the same two collectives per pass."""
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


def _worker(rank, world, port, q, B=1):
    LADDER_MAX = 6
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
        need = binding.shard_need(A, world, anbr, active[wl])
        slice_ = chunk // world
        assert own0 == min(rank * slice_, A) and slice_ % 1024 == 0 and chunk >= A
        tail = 2 * chunk
        vs = lambda a: int(a)
        va = lambda a: chunk + int(a)
        rng = np.random.default_rng(7)
        xg = rng.normal(0, 1, (B, 2 * N + NS))
        is_free = np.concatenate([free_s, free_a, np.full(6 * K, cfg.fix_poses == 0), np.full(4, cfg.fix_intrinsics == 0), np.full(5, cfg.fix_distortion == 0)])
        cost, grad, diag, touched = pv.normal_eq()
        xg = xg * is_free
        # 1. the operator input: owned segments + replicated camera tail; everything else is NaN
        u = np.full((B, tail + NS), np.nan)
        for a in range(own0, own1):
            u[:, vs(a)] = xg[:, wl[a]]; u[:, va(a)] = xg[:, N + wl[a]]
        u[:, tail:] = xg[:, 2 * N:]
        # 2. the rim exchange: owner -> every rank whose rows read the entry
        send = [[] for _ in range(world)]
        for a in range(own0, own1):
            m = int(need[a])
            for k in range(world):
                if k != rank and (m >> k) & 1:
                    send[k].append((a, u[:, vs(a)].copy(), u[:, va(a)].copy()))       # ONE item per rim entry: the values of all B systems
        recv = [None] * world
        dist.all_to_all_single  # (gloo has no variable all_to_all for objects: use all_gather_object of the per-destination lists)
        allsend = [None] * world
        dist.all_gather_object(allsend, send)
        n_recv = 0
        for k in range(world):
            if k == rank: continue
            for a, us_, ua_ in allsend[k][rank]:
                assert (int(need[a]) >> rank) & 1 and not (own0 <= a < own1)
                u[:, vs(a)] = us_; u[:, va(a)] = ua_; n_recv += 1
        n_send = sum(len(x) for x in send)

        y_slice = np.zeros((B, tail + NS)); cam = np.zeros((LADDER_MAX, NS + 1))       # the message has LADDER_MAX slots; B of them are live
        owned = lambda a: own0 <= a < own1

        def add_row(centre_a, cols, coefs, w, cam_cols=None, cam_coefs=None):
            # cols: vector positions (or None); t = w * (J . u); outputs only on owned unknowns; camera + p.q only on the owner of the row
            d = sum(c * u[:, p] for p, c in zip(cols, coefs) if p is not None)
            if cam_cols is not None:
                d = d + sum(c * u[:, tail + p] for p, c in zip(cam_cols, cam_coefs))
            d = d + np.zeros(B)
            assert np.all(d == d), "a row read an operator-input value that was neither owned nor pushed (need set too small)"
            t = w * d
            for p, c, a_of in zip(cols, coefs, row_entries):
                if p is not None and owned(a_of):
                    y_slice[:, p] += c * t
            if owned(centre_a):
                cam[:B, NS] += t * d                             # p.q row by row (tile_pass.hip)
                if cam_cols is not None:
                    for p, c in zip(cam_cols, cam_coefs):
                        cam[:B, p] += c * t

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
        for t_id in (1, 2, 3):
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
        # 4. the ONE collective of the pass: [camera block | p.q]
        cam_t = torch.from_numpy(cam); dist.all_reduce(cam_t)      # [LADDER_MAX][NS + 1] in ONE message
        assert np.all(cam_t.numpy()[B:] == 0.0)
        # (test only) assemble the owned segments to compare with the oracle's global product
        err = 0.0; err_pq = 0.0
        for b in range(B):
            seg = np.zeros(2 * slice_)
            seg[:slice_] = y_slice[b, rank * slice_:(rank + 1) * slice_]; seg[slice_:] = y_slice[b, chunk + rank * slice_:chunk + (rank + 1) * slice_]
            parts = [torch.zeros(2 * slice_, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(seg))
            y = np.zeros(tail + NS)
            for k2 in range(world):
                pk = parts[k2].numpy(); y[k2 * slice_:(k2 + 1) * slice_] = pk[:slice_]; y[chunk + k2 * slice_:chunk + (k2 + 1) * slice_] = pk[slice_:]
            y[tail:] = cam_t.numpy()[b, :NS]
            y *= maskv
            yref_g = pv.jtj_apply(xg[b])
            yref = np.zeros(tail + NS)
            for a in range(A):
                yref[vs(a)] = yref_g[wl[a]]; yref[va(a)] = yref_g[N + wl[a]]
            yref[tail:] = yref_g[2 * N:]
            err = max(err, np.abs(y - yref).max() / (np.abs(yref).max() + 1e-30))
            pq_ref = float(xg[b] @ yref_g)
            err_pq = max(err_pq, abs(float(cam_t.numpy()[b, NS]) - pq_ref) / abs(pq_ref))
        q.put((rank, float(err), int(comp.sum()), int(own1 - own0), A, n_send, n_recv, float(err_pq)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 1), (3, 1), (2, 3)])
def test_sharded_operator_protocol_gloo(world, batch):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + world + 10 * batch
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, batch)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(world))
    A = res[0][4]
    assert sum(r[3] for r in res) == A                       # owned ranges partition the work list
    assert sum(r[5] for r in res) == sum(r[6] for r in res)  # everything pushed is received
    for rank, err, ncomp, nown, _, n_send, n_recv, err_pq in res:
        assert err < 1e-12 and err_pq < 1e-12, (rank, err, err_pq)
        assert nown <= ncomp <= A
    assert all(r[3] > 0 for r in res) and sum(r[5] for r in res) > 0, res      # every rank owns a range and rim values do travel
