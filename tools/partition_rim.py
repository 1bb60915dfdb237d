#!/usr/bin/env python3
"""Rim / ghost sizes of the two ownership schemes for one-process-per-GPU runs, measured on the bench workload (CPU only, numpy).

SURVEY.md section 8(e) and BASELINE.json's north star shard the work by SH SUBVOLUME (lighting/subvolumes.cpp:281-295: a voxel belongs to floor(world / subvolume_size));
the library shards by contiguous RANGES of the brick-Morton ordered work list (whole 1024-entry tiles of the operator pass).  VERDICT r4 item 7:
"align ownership with SH subvolumes or prove it does not matter: commit the measured rim size / ghost fraction for both partitions at 8 ranks".

For W ranks this script builds the work list of the bench scene the way the device does (thin-shell voxels + the stored voxels their rows read; order: Morton code of the
8^3 brick, then position in the brick), assigns every entry an owner under both schemes

    range     : rank k owns entries [k * slice, (k + 1) * slice), slice = ceil(tiles / W) * 1024                              (solver.cpp shard_range)
    subvolume : subvolumes sorted by the Morton code of their index, cut into W runs of ~equal entry count; an entry follows its subvolume

and counts, per rank: owned entries, GHOST entries (foreign active entries whose rows touch an owned unknown: their rows are rebuilt locally, common.hpp
shard_needs_entry), the RIM a rank must receive (foreign entries its rows read), and how many contiguous list ranges a rank's entries form (the tiled operator pass
wants few: a range boundary is a tile boundary).

    python tools/partition_rim.py [--voxels 8e6] [--ranks 8] [--subvolume 0.06] > profiles/r05_partition_rim.json
"""
import argparse
import json
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# ring (+x,-x,+y,-y,+z,-z) then the rest of the forward stencil of an Eg row (common.hpp nbr_offset 0..11): what an entry's rows READ
READ = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [2, 0, 0], [0, 2, 0], [0, 0, 2], [1, 1, 0], [1, 0, 1], [0, 1, 1]], np.int64)
FWD = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2], [1, 1, 0], [1, 0, 1], [0, 1, 1]], np.int64)


def pack(k):
    B = 1 << 20
    return ((k[:, 0] + B) << 42) | ((k[:, 1] + B) << 21) | (k[:, 2] + B)


def morton3(x, y, z):
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    return spread(x) | (spread(y) << np.uint64(1)) | (spread(z) << np.uint64(2))


def lookup(sorted_keys, order, q):
    """index into the original arrays of packed key q, -1 if absent"""
    pos = np.searchsorted(sorted_keys, q)
    pos = np.minimum(pos, len(sorted_keys) - 1)
    hit = sorted_keys[pos] == q
    return np.where(hit, order[pos], -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=float, default=8.0e6)
    ap.add_argument("--band", type=float, default=3.5)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--subvolume", type=float, default=0.06)
    ap.add_argument("--voxel-size", type=float, default=0.001)
    a = ap.parse_args()
    from intrinsic3d_amd import synthetic
    t0 = time.time()
    radius_vox = int(round(np.sqrt(a.voxels / (4.0 * np.pi * 2.0 * a.band))))
    sc = synthetic.make_scene(radius_vox=radius_vox, voxel_size=a.voxel_size, K=1, width=64, height=48, levels=1, band_vox=a.band, seed=1234,
                              cam_dist=2.6 * radius_vox * a.voxel_size, bump_amp_vox=0.5, bump_freq=40.0)
    keys = sc["keys"].astype(np.int64); sdf = sc["sdf"].astype(np.float64); N = len(keys)
    pk = pack(keys); order = np.argsort(pk, kind="stable"); spk = pk[order]
    thres = 1.0 * a.voxel_size
    shell = np.abs(sdf) <= thres
    # active = in the shell with the whole forward stencil stored (owns Eg rows); entries = active voxels + everything their rows read
    fwd_ok = np.ones(N, bool)
    for o in FWD:
        fwd_ok &= lookup(spk, order, pack(keys + o)) >= 0
    active = shell & fwd_ok
    entry = active.copy()
    ai = np.nonzero(active)[0]
    for o in READ:
        j = lookup(spk, order, pack(keys[ai] + o)); entry[j[j >= 0]] = True
    ei = np.nonzero(entry)[0]
    # list order: Morton code of the brick (key >> 3), then position in the brick
    k = keys[ei]; kmin = k.min(0)
    b = (k - kmin) >> 3; inb = (k - kmin) & 7
    code = (morton3(b[:, 0], b[:, 1], b[:, 2]) << np.uint64(9)) | ((inb[:, 2] << 6) | (inb[:, 1] << 3) | inb[:, 0]).astype(np.uint64)
    lo = np.argsort(code, kind="stable"); ei = ei[lo]; k = k[lo]
    A = len(ei); list_pos = np.full(N, -1, np.int64); list_pos[ei] = np.arange(A)
    act = active[ei]
    # neighbours in list space
    cols = []
    for o in READ:
        j = lookup(spk, order, pack(keys[ei] + o))
        cols.append(np.where(j >= 0, list_pos[np.maximum(j, 0)], -1))
    nb = np.stack(cols, 1)
    W = a.ranks
    tiles = (A + 1023) // 1024; slice_ = ((tiles + W - 1) // W) * 1024
    own_range = np.minimum(np.arange(A) // slice_, W - 1)
    sv = np.floor(keys[ei].astype(np.float64) * a.voxel_size / a.subvolume).astype(np.int64)
    svk = pack(sv); usv, inv, cnt = np.unique(svk, return_inverse=True, return_counts=True)
    usv_idx = np.stack([((usv >> 42) & 0x1FFFFF) - (1 << 20), ((usv >> 21) & 0x1FFFFF) - (1 << 20), (usv & 0x1FFFFF) - (1 << 20)], 1)
    m = morton3(usv_idx[:, 0] - usv_idx[:, 0].min(), usv_idx[:, 1] - usv_idx[:, 1].min(), usv_idx[:, 2] - usv_idx[:, 2].min())
    so = np.argsort(m); cum = np.cumsum(cnt[so]); sv_rank = np.empty(len(usv), np.int64); sv_rank[so] = np.minimum((cum - 1) * W // A, W - 1)
    own_sv = sv_rank[inv]

    def measure(own):
        out = []
        nb_own = np.where(nb >= 0, own[np.maximum(nb, 0)], -1)
        for r in range(W):
            mine = own == r
            touches = (nb_own == r).any(1)
            ghost = act & ~mine & touches                       # foreign entries whose rows land on an owned unknown
            compute = mine | ghost
            reads = nb[compute & act]; reads = reads[reads >= 0]
            rim_in = np.unique(reads[own[reads] != r])          # foreign entries this rank's rows read (values pushed to it once per pass)
            runs = int(np.count_nonzero(np.diff(np.nonzero(mine)[0]) > 1) + 1) if mine.any() else 0
            out.append({"owned": int(mine.sum()), "owned_active": int((mine & act).sum()), "ghost": int(ghost.sum()), "rim_in": int(len(rim_in)), "list_ranges": runs,
                        "tiles_touched_1024": int(len(np.unique(np.nonzero(compute)[0] // 1024)))})
        tot = {k2: int(sum(o[k2] for o in out)) for k2 in ("owned", "owned_active", "ghost", "rim_in")}
        return {"per_rank": out, "ghost_fraction": tot["ghost"] / max(1, tot["owned_active"]), "rim_fraction": tot["rim_in"] / max(1, tot["owned"]),
                "max_owned_over_mean": max(o["owned"] for o in out) / (tot["owned"] / float(W)), "max_list_ranges": max(o["list_ranges"] for o in out),
                "max_work_over_mean": max(o["owned_active"] + o["ghost"] for o in out) / ((tot["owned_active"] + tot["ghost"]) / float(W))}
    res = {"what": "ownership of the bench work list at %d ranks: contiguous list ranges (the library) vs SH subvolumes (SURVEY 8(e)); ghost = foreign active entries whose rows are rebuilt "
                   "locally, rim_in = foreign entries a rank's rows read (2 floats each per PCG pass)" % W,
           "stored_voxels": int(N), "work_list_entries": int(A), "active_entries": int(act.sum()), "subvolumes": int(len(usv)), "subvolume_m": a.subvolume, "ranks": W,
           "range": measure(own_range), "subvolume": measure(own_sv), "seconds": time.time() - t0}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
