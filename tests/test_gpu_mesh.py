"""-m gpu: mesh export (marching cubes on the resident grid + PLY), SURVEY.md §8f rank 1.

Triangle-for-triangle identity: vertices, colours and faces of i3d_get_mesh equal the oracle's restatement of
MarchingCubes<VoxelSBR>::extractSurface (mesh/marching_cubes.cpp:65-343) bit for bit.  Also held: the vertex arithmetic
against a numpy restatement, crack-freeness on arbitrary sign patterns, outward orientation, and the byte layout of the PLY stream."""
import collections
import struct

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _edge_counts(faces):
    und = collections.Counter(); dirc = collections.Counter()
    for a, b, c in faces.tolist():
        for u, v in ((a, b), (b, c), (c, a)):
            und[(min(u, v), max(u, v))] += 1; dirc[(u, v)] += 1
    return und, dirc


def _np_lerp(t0, t1, v0, v1):
    t0 = np.float32(t0); t1 = np.float32(t1); v0 = np.float32(v0); v1 = np.float32(v1)
    mu = np.clip((np.float32(0) - t0) / (t1 - t0), np.float32(0), np.float32(1)).astype(np.float32)
    out = (v0 + mu * (v1 - v0)).astype(np.float32)
    out = np.where(np.abs(t0 - t1) < np.float32(1e-5), v0, out)
    out = np.where(np.abs(np.float32(0) - t1) < np.float32(1e-5), v1, out)
    out = np.where(np.abs(np.float32(0) - t0) < np.float32(1e-5), v0, out)
    return out.astype(np.float32)


def test_sphere_mesh_is_closed_oriented_and_on_the_surface(oracle, tmp_path):
    from intrinsic3d_amd import binding
    sc = helpers.small_scene(seed=2, radius_vox=10, K=1, width=64, height=48)
    with binding.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        v, c, f = ctx.extract_mesh(use_refined_sdf=True, color_mode=0, largest_component_only=True)
        ctx.export_mesh_ply(tmp_path / "mesh.ply", True, 0, True)
        va, ca, fa = ctx.extract_mesh(use_refined_sdf=True, color_mode=1)
    assert len(f) > 1000 and f.max() < len(v)
    # The same lattice edge is interpolated in BOTH directions by neighbouring cells (edge 2 of one cell is edge 0 of the next, with the
    # endpoints swapped: marching_cubes.cpp:183-247), which can differ in the last float bit; merge() only unifies bit-identical
    # positions, so the reference mesh carries such duplicate vertices too.  Topology is therefore checked on a welded copy.
    _, weld = np.unique(np.round(v.astype(np.float64) / float(sc["voxel_size"]) * 1e4).astype(np.int64), axis=0, return_inverse=True)
    weld = weld.ravel()
    assert len(v) - (weld.max() + 1) < 0.1 * len(v)
    fw = weld[f]
    und, dirc = _edge_counts(fw)
    assert set(und.values()) == {2}                                   # watertight 2-manifold
    assert set(dirc.values()) == {1}                                  # consistently oriented
    E = len(und); used = np.unique(fw)
    assert len(used) - E + len(fw) == 2                               # one sphere
    # outward normals (from sdf < 0 to sdf > 0): the scene is a bumpy sphere around sc["center"]
    p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(p1 - p0, p2 - p0); cen = (p0 + p1 + p2) / 3 - sc["center"][None, :]
    assert ((n * cen).sum(axis=1) > 0).mean() > 0.999
    # every vertex sits on a lattice edge (two coordinates are exact voxel multiples) and on the iso-surface of the scene's sdf
    vs = np.float32(sc["voxel_size"])
    onlat = np.isclose(v / vs, np.round(v / vs), atol=1e-4).sum(axis=1)
    assert (onlat >= 2).all()
    assert np.abs(sc["scene"].sdf(v.astype(np.float64))).max() < 0.2 * float(vs)
    # albedo mode: grey = clamp(albedo * 255) = 153 for the constant initial albedo 0.6
    assert len(va) == len(v) or len(va) > 0
    assert (ca == np.uint8(0.6 * 255.0)).all()
    # the PLY stream (mesh.cpp:41-100)
    raw = open(tmp_path / "mesh.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().strip().split("\n")
    assert lines[:6] == ["ply", "format binary_little_endian 1.0", f"element vertex {len(v)}", "property float x", "property float y", "property float z"]
    assert lines[6:9] == ["property uchar red", "property uchar green", "property uchar blue"] and lines[9] == f"element face {len(f)}" and lines[10] == "property list uchar int vertex_indices"
    assert len(body) == len(v) * 15 + len(f) * 13
    vx = np.frombuffer(body[:15 * len(v)], dtype=np.dtype([("p", "<f4", 3), ("c", "u1", 3)]))
    assert np.array_equal(vx["p"], v) and np.array_equal(vx["c"], c)
    fc = np.frombuffer(body[15 * len(v):], dtype=np.dtype([("n", "u1"), ("i", "<i4", 3)]))
    assert (fc["n"] == 3).all() and np.array_equal(fc["i"], f)


def test_random_signs_vertex_set_and_manifoldness(oracle):
    """dense 14^3 block with random sdf values (every ambiguous configuration occurs), a few missing / invalid voxels:
    vertex set == numpy restatement of the reference's rules; the surface is an oriented manifold (possibly with boundary)."""
    from intrinsic3d_amd import binding
    rng = np.random.default_rng(5)
    R = 14; vs = np.float32(0.004)
    g = np.stack(np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.int32)
    sdf = rng.normal(0, 0.004, len(g)).astype(np.float32)
    sdf[rng.integers(0, len(g), 20)] = 0.0                              # exact zeros: the |iso - sdf| < 1e-5 branches
    w = np.ones(len(g), np.float32)
    col = rng.integers(0, 256, (len(g), 3)).astype(np.uint8)
    present = np.ones(len(g), bool); present[rng.integers(0, len(g), 15)] = False      # missing voxels
    w[rng.integers(0, len(g), 15)] = 0.0                                                # invalid voxels (weight 0)
    keys, sdf_p, w_p, col_p = g[present], sdf[present], w[present], col[present]
    with binding.Context(0) as ctx:
        ctx.set_grid(vs, keys, sdf_p.astype(np.float64), sdf_p.astype(np.float64), np.full(len(keys), 0.6), w_p, col_p)
        v, c, f = ctx.extract_mesh(use_refined_sdf=False, color_mode=0, largest_component_only=False)
    # ---- numpy restatement: eligible cells, cut edges, vertex positions ----
    vol = np.full((R + 1, R + 1, R + 1), np.nan, np.float32); wv = np.zeros((R + 1, R + 1, R + 1), np.float32); ex = np.zeros((R + 1, R + 1, R + 1), bool)
    vol[keys[:, 0], keys[:, 1], keys[:, 2]] = sdf_p; wv[keys[:, 0], keys[:, 1], keys[:, 2]] = w_p; ex[keys[:, 0], keys[:, 1], keys[:, 2]] = True
    corners = [(1, 1, 0), (1, 0, 0), (0, 0, 0), (0, 1, 0), (1, 1, 1), (1, 0, 1), (0, 0, 1), (0, 1, 1)]
    EA = [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3]; EB = [1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7]
    expect = set(); skipped_cells = set()
    for x, y, z in keys.tolist():
        ok = all(ex[x + dx, y + dy, z + dz] and wv[x + dx, y + dy, z + dz] != 0.0 for dx, dy, dz in corners)
        if not ok:
            skipped_cells.add((x, y, z)); continue
        s = [vol[x + dx, y + dy, z + dz] for dx, dy, dz in corners]
        idx = sum(1 << i for i in range(8) if s[i] < 0)
        if idx in (0, 255):
            continue
        for e in range(12):
            a, b = EA[e], EB[e]
            if (s[a] < 0) != (s[b] < 0):
                pa = np.float32([x + corners[a][0], y + corners[a][1], z + corners[a][2]]) * vs
                pb = np.float32([x + corners[b][0], y + corners[b][1], z + corners[b][2]]) * vs
                p = _np_lerp(s[a], s[b], pa, pb)
                expect.add(tuple(p.tolist()))
    got = set(map(tuple, v[np.unique(f)].tolist())) if len(f) else set()
    assert got <= set(map(tuple, v.tolist()))
    assert set(map(tuple, v.tolist())) == expect, (len(v), len(expect))
    # ---- crack-freeness on a block WITHOUT degenerate values / holes: every (welded) edge is shared by exactly two triangles unless it
    #      lies on the outer faces of the block, whatever the sign pattern (all ambiguous face / cell configurations occur)
    sdf2 = rng.normal(0, 0.004, len(g)).astype(np.float32); sdf2[np.abs(sdf2) < 1e-4] = 1e-4
    with binding.Context(0) as ctx:
        ctx.set_grid(vs, g, sdf2.astype(np.float64), sdf2.astype(np.float64), np.full(len(g), 0.6), np.ones(len(g), np.float32), col)
        v, c, f = ctx.extract_mesh(use_refined_sdf=False, color_mode=0, largest_component_only=False)
    _, weld = np.unique(np.round(v.astype(np.float64) / float(vs) * 1e4).astype(np.int64), axis=0, return_inverse=True)   # see the note in the sphere test
    weld = weld.ravel(); rep = np.zeros(weld.max() + 1, np.int64); rep[weld] = np.arange(len(v))
    und, dirc = _edge_counts(weld[f])
    assert max(und.values()) == 2 and max(dirc.values()) == 1           # manifold and consistently oriented everywhere
    # open edges: the border of the block and — with Bourke's table, which the reference uses — the ambiguous faces that two neighbouring
    # cells triangulate differently (a known property of the classic 256-case table; the reference's meshes have the same holes)
    lonely = [1 for n in und.values() if n == 1]
    assert 0 < len(lonely) < 0.2 * len(und)

def test_mesh_identical_to_the_oracle_and_the_reference_code(oracle, tmp_path):
    """vertices, colours, faces and the PLY bytes == the restated (and, if present, the reference's own) marching cubes"""
    from intrinsic3d_amd import binding
    O = oracle
    rng = np.random.default_rng(9)
    R = 16; vs = np.float32(0.004)
    g = np.stack(np.meshgrid(np.arange(-R // 2, R // 2), np.arange(-R // 2, R // 2), np.arange(-R // 2, R // 2), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.int32)
    c = g.astype(np.float64) * float(vs)
    sdf = (np.linalg.norm(c + 0.0007, axis=1) - 0.0221 + 0.002 * np.sin(900 * c[:, 0]) * np.cos(700 * c[:, 1])).astype(np.float32)
    sdf[rng.integers(0, len(g), 30)] = 0.0
    keep = rng.random(len(g)) < 0.95
    g, sdf = g[keep], sdf[keep]
    w = np.where(rng.random(len(g)) < 0.03, 0.0, 1.0).astype(np.float32)
    col = rng.integers(0, 256, (len(g), 3)).astype(np.uint8)
    perm = rng.permutation(len(g)); g, sdf, w, col = g[perm], sdf[perm], w[perm], col[perm]
    og = O.Grid.from_voxels(vs, g, sdf, w, col)
    a = og.export()
    refined = a["sdf"] + rng.normal(0, 0.0004, len(a["sdf"]))
    og.import_fields(sdf_refined=refined)
    for use_refined in (False, True):
        ov, oc, of = O.marching_cubes(og, use_refined=use_refined)
        with binding.Context(0) as ctx:
            ctx.set_grid(vs, a["keys"], a["sdf"], refined, a["albedo"], a["weight"], a["color"])
            v, cc, f = ctx.extract_mesh(use_refined_sdf=use_refined, color_mode=0, largest_component_only=False)
            ctx.export_mesh_ply(tmp_path / "dev.ply", use_refined, 0, False)
        assert len(of) > 300
        assert v.tobytes() == ov.tobytes() and cc.tobytes() == oc.tobytes() and f.tobytes() == of.tobytes()
        binding.write_ply(tmp_path / "orc.ply", ov, oc, of)
        assert open(tmp_path / "dev.ply", "rb").read() == open(tmp_path / "orc.ply", "rb").read()
    og.free()


def test_export_colour_modes_device_equals_host_instantiation(tmp_path):
    """the debug colour modes of the export (k_vis_colors: the device instantiation of vis_colors.hpp) against i3d_visualization_colors (the host instantiation
    of the same header) on a grid with holes, zero-weight voxels, random colours and several lighting subvolumes: a mesh exported in mode
    X must equal, byte for byte, the plain export of the same grid repainted with the host's colours for X.  Plus the two refusals."""
    from intrinsic3d_amd import binding
    rng = np.random.default_rng(5)
    vs = 0.004; r = 11
    g = np.stack(np.meshgrid(*[np.arange(-r, r + 1)] * 3, indexing="ij"), -1).reshape(-1, 3)
    d = np.linalg.norm(g + 0.3, axis=1) - 7.3
    keep = (np.abs(d) < 2.6) & (rng.random(len(g)) > 0.03)
    keys = (g[keep] + np.array([31, -4, 12])).astype(np.int32); n = len(keys); keys = keys[rng.permutation(n)]
    sdf = (np.linalg.norm(keys - np.array([31, -4, 12]) + 0.3, axis=1) - 7.3) * vs + rng.normal(0, 0.1 * vs, n)
    w = rng.uniform(0.5, 30, n).astype(np.float32); w[rng.random(n) < 0.04] = 0.0
    alb = rng.uniform(0.05, 1.1, n); col = rng.integers(0, 256, (n, 3)).astype(np.uint8); col[rng.random(n) < 0.03] = 0
    size = 0.02
    with binding.Context(0) as ctx:
        ctx.set_grid(vs, keys, sdf, sdf, alb, w, col)
        with pytest.raises(binding.I3DError):
            ctx.export_mesh_ply(tmp_path / "x.ply", True, 6, False)                      # a shading view before any lighting estimate
        with pytest.raises(binding.I3DError):
            ctx.export_mesh_ply(tmp_path / "x.ply", True, 9, False)                      # no such mode
        sh, idx, _ = ctx.estimate_sh(size, 10.0, 2.0 * vs)
        assert len(sh) > 8
        for mode, mid in binding.COLOR_MODES.items():
            if mid == 0:
                continue
            ctx.export_mesh_ply(tmp_path / "dev.ply", True, mid, False)
            host = binding.visualization_colors(mode, vs, keys, sdf, alb, w, col, size, idx, sh, visit_rank=np.arange(n))      # the visit order is the order the grid was handed over in
            ctx.update_grid(color=host); ctx.export_mesh_ply(tmp_path / "host.ply", True, 0, False); ctx.update_grid(color=col)
            a = open(tmp_path / "dev.ply", "rb").read(); b = open(tmp_path / "host.ply", "rb").read()
            assert a == b and len(a) > 10000, (mode, len(a), len(b))
            assert len(np.unique(host)) > 3, mode
