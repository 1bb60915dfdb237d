"""CPU: the restated oracle against THE REFERENCE'S OWN hot-path code, end to end.

oracle/_ref compiles — besides the functors and primitives of tests/test_oracle_vs_ref.py — the reference's whole classes from /root/reference
(oracle/extract_ref.py, round-3 chunks): Optimizer::optimize / addVoxelResiduals / buildProblem / fixVoxelParams (optimizer.cpp:92-361), NLSSolver
(nls_solver.cpp:45-394), SDFColorization incl. collectObservations / computeObservation / add / compute (colorization.cpp:52-370), the cost-function
factories ShadingCost / VolumetricRegularizer / SurfaceStabRegularizer / AlbedoRegularizer::create, SDFOperators::computeSurfaceNormal, SDFAlgorithms
(convert, upsample, interpolate, clearVoxelsOutsideThinShell, correctSDF, clearInvalidVoxels; algorithms.cpp:47-458), math.cpp:43-163, Subvolumes
(subvolumes.cpp:43-304), LightingSVSH::estimate / computeVoxelShCoeffs (lighting_svsh.cpp:54-346), SparseVoxelGrid incl. integrate / alloc
(sparse_voxel_grid.cpp:43-467,572-602), Camera, rgbd/processing.cpp:49-301 and Intrinsic3D::refine / prepare* / finish* / recomputeColors
(intrinsic3d.cpp:206-409).  Underneath run stand-ins for Eigen / OpenCV containers and a SECOND, independently written Ceres-2.1.0 LM + CGNR
(oracle/ref_shim/mini_ceres_solver.hpp).  `ref_py.pipeline()` exposes all of it behind the same Python classes as the oracle, so every test
below runs one function on both and compares.

What this pins: every order- / structure-critical piece the round-2 review listed as "restated on both sides" — Eg / Er / Es / Ea row sets
in reference order, the voxels_added edge rule, fixed flags, type weights, residuals and all 29 partials, thin shell + upsample key order and
fields, subvolume ids + interpolated SH, recolourisation, fusion alloc / integrate / correctSDF record order, and the level schedule.
What stays unpinned: Ceres itself (two restatements agree), Eigen's 4x4 inverse and OpenCV's pyrDown / cvtColor.
"""
import os
import re
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def R():
    from oracle import ref_py
    if os.path.isdir("/root/reference"):
        ref_py.build()
    if not ref_py.available():
        pytest.skip("oracle/_ref/libref_i3d.so not built (no /root/reference on this box and no prebuilt library in the tree)")
    return ref_py.pipeline()


def _mangled_scene(seed, radius_vox=10, K=4, levels=1):
    """A scene with what the review asked the row tests to cover: holes, zero-weight voxels, black voxels (NaN chroma weight), negative
    coordinates (keys shifted by an integer vector, poses compensated) and non-zero lens distortion."""
    sc = helpers.small_scene(seed=seed, radius_vox=radius_vox, K=K, levels=levels)
    rng = np.random.default_rng(100 + seed)
    n = sc["keys"].shape[0]
    keep = rng.random(n) > 0.04                                        # holes
    for k in ("keys", "sdf", "weight", "color"):
        sc[k] = np.ascontiguousarray(sc[k][keep])
    n = sc["keys"].shape[0]
    sc["weight"][rng.random(n) < 0.03] = 0.0                           # never-observed voxels inside the shell
    sc["color"][rng.random(n) < 0.01] = 0                              # black: chroma weight is NaN -> no Ea row (albedo_regularizer.cpp:71-72)
    off = np.array([-(2 * radius_vox + 9), -7, -(radius_vox + 3)], np.int32)
    sc["keys"] = np.ascontiguousarray(sc["keys"] + off)
    from intrinsic3d_amd.synthetic import aa_to_rotmat
    d = off.astype(np.float64) * float(sc["voxel_size"])
    for f in range(sc["K"]):
        sc["poses"][f, 3:] -= aa_to_rotmat(sc["poses"][f, :3]) @ d
    sc["dist"] = np.array([0.02, -0.01, 0.003, 0.0004, -0.0003])
    return sc


def _setup_upsampled(M, sc, seed=3):
    """Like helpers.oracle_setup but one level further down the schedule: coarse grid -> thin shell -> x2 upsample -> thin shell.  The upsampled
    grid holds what a fused volume never does: zero-weight voxels INSIDE the shell (<= 4 valid corners, algorithms.cpp:163-164)."""
    g0 = M.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    fr = M.Frames(sc["frames"], sc["levels"])
    g0.clear_outside_shell(2.0 * float(sc["voxel_size"]))
    g = g0.upsample(); g0.free()
    thres = 1.5 * float(g.voxel_size)
    g.clear_outside_shell(thres)
    a = g.export(); rng = np.random.default_rng(seed); n = len(g)
    g.import_fields(sdf_refined=a["sdf_refined"] + rng.normal(0, 0.02 * float(g.voxel_size), n), albedo=0.6 + rng.normal(0, 0.02, n))
    rc, sh, idx, vsh, has, st = M.estimate_sh(g, 0.05, 10.0, thres)
    assert rc == 0
    return g, fr, g.export(), vsh, thres


def _both(O, R, sc, upsampled=False, **kw):
    if upsampled:
        return _setup_upsampled(O, sc), _setup_upsampled(R, sc)
    a = helpers.oracle_setup(O, sc, **kw); b = helpers.oracle_setup(R, sc, **kw)
    return a, b


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_row_assembly_equals_the_reference_code(oracle, R, seed):
    """optimizer.cpp:176-361 + nls_solver.cpp:172-187,228-235,379-394 executed from the reference vs the oracle: bit for bit."""
    O = oracle
    sc = _mangled_scene(seed, radius_vox=7 if seed == 5 else 10)
    if seed == 5:                                                                    # coarse level first: the rows of an UPSAMPLED grid
        sc["voxel_size"] = np.float32(2.0 * float(sc["voxel_size"]))
        sc["keys"] = np.ascontiguousarray(sc["keys"] // 2); _, first = np.unique(sc["keys"], axis=0, return_index=True); first.sort()
        for k in ("keys", "sdf", "weight", "color"):
            sc[k] = np.ascontiguousarray(sc[k][first])
    (go, fo, ao, vsh, thres), (gr, fr, ar, vshr, _) = _both(O, R, sc, upsampled=(seed == 5))
    for k in ("keys", "sdf", "sdf_refined", "albedo", "weight", "color"):          # convert + thin shell (+ upsample): visit order and fields
        assert np.array_equal(ao[k], ar[k]), k
    assert np.abs(vsh - vshr).max() < 1e-9
    assert (ao["keys"] < 0).any() and ((ao["weight"] == 0).any() or seed != 5)
    for it in (0, 2):
        po = O.ProblemView(go, fo, helpers.oracle_cfg(O, thres, iterations=3), sc["intr"], sc["dist"], sc["poses"], vsh, iteration=it)
        pr = R.ProblemView(gr, fr, helpers.oracle_cfg(R, thres, iterations=3), sc["intr"], sc["dist"], sc["poses"], vsh, iteration=it)
        assert po.rows == pr.rows and min(po.rows) > 100, (po.rows, pr.rows)
        vo, f_o, wo, ro, Jo = po.eg(); vr, f_r, wr, rr, Jr = pr.eg()
        assert np.array_equal(vo, vr) and np.array_equal(f_o, f_r)                  # Eg rows: same (voxel, keyframe) SEQUENCE
        assert np.array_equal(wo, wr), np.abs(wo - wr).max()                        # obs.weight * sdfToWeight * lambda / sum * 1000, bit-exact
        assert np.array_equal(ro, rr) and np.array_equal(Jo, Jr)                    # residual + 29 partials (DynamicAutoDiff stride-4 passes)
        for t in (1, 2, 3):
            a = po.reg(t); b = pr.reg(t)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), t
            if t == 3:
                assert np.array_equal(a[1], b[1])                                   # Ea: the visit-order-dependent edge set (voxels_added rule)
        flo = po.flags(); flr = pr.flags()                                          # fixVoxelParams; 2 = block not in the reference's problem
        for k in ("fix_sdf", "fix_alb"):
            present = flr[k] != 2
            assert np.array_equal(flo[k][present], flr[k][present]) and present.sum() > 100
            if k == "fix_sdf":       # an active voxel always owns its Es row; its albedo may be touched by no row at all (no Eg, ring invalid) — then
                assert not ((~present) & (flo[k] == 0) & (flo["active"] == 1)).any()     # Ceres never sees the block and the oracle keeps a zero column
        po.free(); pr.free()
    for h in (go, gr, fo, fr):
        h.free()


def test_optimize_equals_the_reference_code_on_a_second_ceres(oracle, R):
    """Optimizer::optimize of the reference (3 outer iterations; NLSSolver::solve -> mini-ceres) vs oracle.optimize: same accept / reject
    sequence and PCG counts, fields to round-off — with Ceres' own PCG stop and with a pinned count; fixed poses as well."""
    O = oracle
    sc = _mangled_scene(6)
    (go, fo, ao, vsh, thres), (gr, fr, _, _, _) = _both(O, R, sc)
    for kw in (dict(cg_fixed_iterations=-1), dict(cg_fixed_iterations=5, fix_poses=1, fix_distortion=1), dict(cg_fixed_iterations=-1, lambda_a=-1.0)):
        for g in (go, gr):
            g.import_fields(sdf_refined=ao["sdf_refined"], albedo=ao["albedo"], color=ao["color"])
        rc1, io, do, po, so = O.optimize(go, fo, helpers.oracle_cfg(O, thres, iterations=3, **kw), sc["intr"], sc["dist"], sc["poses"], vsh)
        rc2, ir, dr, pr, sr = R.optimize(gr, fr, helpers.oracle_cfg(R, thres, iterations=3, **kw), sc["intr"], sc["dist"], sc["poses"], vsh)
        assert rc1 == 0 and rc2 == 0
        for a, b in zip(so, sr):
            assert list(a.rows) == list(b.rows) and a.num_params == b.num_params and a.num_rows_reduced == b.num_rows_reduced
            assert a.n_attempts == b.n_attempts and list(a.cg_iters[:a.n_attempts]) == list(b.cg_iters[:b.n_attempts])
            assert list(a.accepted[:a.n_attempts]) == list(b.accepted[:b.n_attempts]) and a.termination == b.termination
            assert abs(a.cost_initial - b.cost_initial) <= 1e-12 * a.cost_initial and abs(a.cost_final - b.cost_final) <= 1e-12 * a.cost_final
            assert abs(a.final_radius - b.final_radius) <= 1e-9 * a.final_radius
        eo = go.export(); er = gr.export()
        moved = np.abs(eo["sdf_refined"] - ao["sdf_refined"]).max()
        assert moved > 1e-2 * float(sc["voxel_size"])
        assert np.abs(eo["sdf_refined"] - er["sdf_refined"]).max() <= 1e-9 * moved and np.abs(eo["albedo"] - er["albedo"]).max() <= 1e-10
        assert np.abs(io - ir).max() <= 1e-9 * np.abs(io).max() and np.abs(do - dr).max() <= 1e-10 and np.abs(po - pr).max() <= 1e-10


@pytest.mark.parametrize("seed", [2, 8])
def test_level_transitions_lighting_and_recolouring_equal_the_reference_code(oracle, R, seed):
    """recomputeColors, clearVoxelsOutsideThinShell, Subvolumes + LightingSVSH::estimate + computeVoxelShCoeffs, upsample (x2, twice):
    key ORDER, fields, subvolume ids — bit-exact; SH coefficients to 1e-12 (two LM implementations)."""
    sc = _mangled_scene(seed, radius_vox=9, levels=2)

    def run(M):
        g = M.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        fr = M.Frames(sc["frames"], sc["levels"]); out = {}
        assert M.recompute_colors(g, fr, sc["intr"], sc["dist"], sc["poses"], 0.02, 5) == 0
        out["recolour"] = g.export()
        thres = 2.0 * float(sc["voxel_size"])
        g.clear_outside_shell(thres); out["shell"] = g.export()
        rng = np.random.default_rng(seed); a = out["shell"]
        g.import_fields(albedo=0.6 + rng.normal(0, 0.02, len(g)), sdf_refined=a["sdf_refined"] + rng.normal(0, 2e-5, len(g)))
        rc, sh, idx, vsh, has, st = M.estimate_sh(g, 0.03, 10.0, thres); assert rc == 0
        out["sh"] = (sh, idx, vsh, has, st.data_rows, st.reg_rows, st.lm_iterations)
        up = g.upsample(); out["up"] = up.export()
        up.clear_outside_shell(1.5 * float(up.voxel_size)); out["up_shell"] = up.export()
        up2 = up.upsample(); out["up2"] = up2.export()
        for h in (g, up, up2, fr):
            h.free()
        return out

    a = run(oracle); b = run(R)
    for stage in ("recolour", "shell", "up", "up_shell", "up2"):
        for k in ("keys", "sdf", "sdf_refined", "albedo", "weight", "color"):
            assert np.array_equal(a[stage][k], b[stage][k]), (stage, k)
    assert len(a["up2"]["keys"]) > 8 * len(a["up_shell"]["keys"]) - 1 and (a["recolour"]["color"] != sc["color"][:1]).any()
    sa, sb = a["sh"], b["sh"]
    assert np.array_equal(sa[1], sb[1]) and len(sa[0]) > 8                              # subvolume ids in map order
    assert np.array_equal(sa[3], sb[3]) and sa[4:] == sb[4:]                            # in-shell mask; data rows, regulariser rows, LM iterations
    assert np.abs(sa[0] - sb[0]).max() <= 1e-12 and np.abs(sa[2] - sb[2]).max() <= 1e-12


def test_refine_schedule_equals_the_reference_code(oracle, R):
    """Intrinsic3D::refine (intrinsic3d.cpp:206-290: 2 grid levels x (2, 1) pyramid levels, thin shell, SVSH, optimize, recolourise, upsample)
    compiled from the reference vs oracle.refine."""
    sc = helpers.small_scene(seed=5, radius_vox=8, K=4, levels=2)

    def run(M):
        g = M.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        fr = M.Frames(sc["frames"], sc["levels"])
        rc, intr, dist, poses, done = M.refine(g, fr, helpers.oracle_cfg(M, 0.0, iterations=2), 2, 2, 2.0, 1.0, 1, 0.03, 10.0, sc["intr"], sc["dist"], sc["poses"])
        e = g.export(); g.free(); fr.free()
        o = np.lexsort(e["keys"].T[::-1])                                               # the reference hands its last grid to a callback: compare by key
        return rc, intr, dist, poses, done, {k: (v[o] if getattr(v, "shape", ())[:1] == e["keys"].shape[:1] else v) for k, v in e.items()}

    a = run(oracle); b = run(R)
    assert a[0] == 0 and b[0] == 0 and a[4] == b[4] == 3
    ea, eb = a[5], b[5]
    for k in ("keys", "sdf", "weight", "color"):
        assert np.array_equal(ea[k], eb[k]), k
    assert np.abs(ea["sdf_refined"] - eb["sdf_refined"]).max() <= 1e-10 and np.abs(ea["albedo"] - eb["albedo"]).max() <= 1e-9
    assert np.abs(a[1] - b[1]).max() <= 1e-8 and np.abs(a[2] - b[2]).max() <= 1e-8 and np.abs(a[3] - b[3]).max() <= 1e-9
    assert np.abs(ea["sdf_refined"] - ea["sdf"]).max() > 1e-3 * float(ea["voxel_size"])


def _fusion_frames(seed, exact):
    from intrinsic3d_amd import synthetic
    from make_dataset import pose_vec_to_cam_to_world
    sc = synthetic.make_scene(radius_vox=10, K=4, width=96, height=72, levels=1, seed=seed)
    rng = np.random.default_rng(seed); frames = []
    rots = [np.eye(3), np.array([[0., -1, 0], [1, 0, 0], [0, 0, 1]]), np.array([[1., 0, 0], [0, 0, -1], [0, 1, 0]]), np.array([[0., 0, 1], [0, 1, 0], [-1, 0, 0]])]
    for i, (fr, pose) in enumerate(zip(sc["frames"], sc["poses"])):
        d = fr["depth"][0].copy(); d[d > 0] += rng.normal(0, 0.0015, int((d > 0).sum())).astype(np.float32)
        bgr = fr["bgr"][0].copy(); bgr[..., 0] //= 2
        T = pose_vec_to_cam_to_world(np.asarray(pose, np.float64)).astype(np.float32)
        if exact:      # quarter-turn rotations and dyadic translations: Matrix4f::inverse() is exact whatever its operation order
            T = T.copy(); T[:3, :3] = rots[i % 4].astype(np.float32); T[:3, 3] = np.round(T[:3, 3] * 64) / 64
        frames.append((d, bgr, T))
    return sc, frames


def test_fusion_equals_the_reference_code(oracle, R):
    """SparseVoxelGrid<Voxel>::alloc / integrate, correctSDF, clearInvalidVoxels (sparse_voxel_grid.cpp:301-467, algorithms.cpp:260-366) from the
    reference vs the oracle: record ORDER, sdf, weights, colours bit-exact for poses whose inverse is exact; general poses to 1e-6 (Eigen's 4x4 inverse
    is unpinned)."""
    for exact in (True, False):
        sc, frames = _fusion_frames(9, exact)
        intr = sc["intr"].astype(np.float32); cintr = intr * np.float32(0.5)
        out = []
        for M in (oracle, R):
            f = M.Fusion(sc["voxel_size"], 0.1, 10.0)
            for d, bgr, T in frames:
                f.integrate(d, intr, bgr[::2, ::2].copy(), cintr, T, 2)             # colour camera != depth camera
            raw = f.export(); f.finish(10); out.append((raw, f.export()))
        (raw_o, fin_o), (raw_r, fin_r) = out
        assert len(raw_o["sdf"]) > 3000 and len(fin_o["sdf"]) < len(raw_o["sdf"])
        for stage, (x, y) in enumerate(((raw_o, raw_r), (fin_o, fin_r))):
            assert np.array_equal(x["keys"], y["keys"])                                 # first-insertion order -> map order -> file order
            if exact:
                for k in ("sdf", "weight", "color"):
                    assert np.array_equal(x[k], y[k]), k
            else:    # last-bit differences of the inverted pose; after correctSDF they can flip a '<' between near-equal candidates (2 % allowed)
                assert np.abs(x["weight"] - y["weight"]).max() <= 1e-4 and (np.abs(x["sdf"] - y["sdf"]) > 1e-6).mean() < (0.02 if stage else 0.002)


def test_small_helpers_equal_the_reference_code(oracle, R):
    """poseVecAAToMat (math.cpp:151-163), erodeDiscontinuities / computeNormals / resizeDepth (processing.cpp:49-232), Pyramid::downsampleDepth."""
    rng = np.random.default_rng(0)
    for _ in range(50):
        p = np.concatenate([rng.normal(0, 1.0, 3) * rng.choice([1e-9, 0.3, 2.0]), rng.normal(0, 1, 3)])
        Ro, to = oracle.pose_to_mat(p); Rr, tr = R.pose_to_mat(p)
        assert np.array_equal(Ro, Rr) and np.array_equal(to, tr)
    d = (1.0 + 0.3 * rng.random((60, 80))).astype(np.float32); d[rng.random(d.shape) < 0.1] = 0; d[20:30, 30:50] += 0.8
    assert np.array_equal(oracle.erode_discontinuities(d, 2, 0.5), R.erode_discontinuities(d, 2, 0.5))
    cam = np.array([90.0, 91.0, 39.5, 29.5], np.float32)
    assert np.array_equal(oracle.compute_normals(d, cam, 0.3), R.compute_normals(d, cam, 0.3))
    assert np.array_equal(oracle.depth_down(d), R.depth_down(d))
    out_cam = np.array([130.0, 131.0, 63.5, 47.5], np.float32)
    assert np.array_equal(oracle.resize_depth(d, cam, 128, 96, out_cam), R.resize_depth(d, cam, 128, 96, out_cam))
    same = R.resize_depth(d, cam, 80, 60, out_cam)                                       # same size: a clone, whatever the two cameras are (processing.cpp:135-139)
    assert np.array_equal(same, d) and np.array_equal(oracle.resize_depth(d, cam, 80, 60, out_cam), d)


def _by_key(d):
    o = np.lexsort((d["keys"][:, 2], d["keys"][:, 1], d["keys"][:, 0]))
    return {k: (v[o] if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == len(o) else v) for k, v in d.items()}


def test_tsdf_files_equal_the_reference_writer_and_reader(R, tmp_path):
    """SparseVoxelGrid<Voxel>::save / load (sparse_voxel_grid.cpp:484-569) of the reference on a fused grid vs the product's host/io.cpp: the product
    reads the reference's file record for record in file order; the product's file of the same records equals the reference's byte for byte outside the
    struct's pad byte (record offset 23, indeterminate in the reference); the reference's load() reads the product's file."""
    from intrinsic3d_amd import binding
    from oracle import ref_py
    sc, frames = _fusion_frames(11, True)
    intr = sc["intr"].astype(np.float32)
    f = R.Fusion(sc["voxel_size"], 0.1, 10.0)
    for d, bgr, T in frames[:2]:
        f.integrate(d, intr, bgr, intr, T, 2)
    f.finish(3); rec = f.export(); n = len(rec["sdf"]); assert n > 2000
    ref_path = tmp_path / "ref.tsdf"; ref_py.tsdf_save(f, ref_path)
    got = binding.tsdf_read(ref_path)
    for k in ("keys", "sdf", "weight", "color"):
        assert np.array_equal(got[k], rec[k]), k
    assert got["voxel_size"] == np.float32(sc["voxel_size"]) and got["truncation"] == np.float32(sc["voxel_size"]) * np.float32(5)
    our_path = tmp_path / "ours.tsdf"
    binding.tsdf_write(our_path, got["voxel_size"], got["keys"], got["sdf"], got["weight"], got["color"], truncation=got["truncation"],
                       integration_weight_sample=got["integration_weight_sample"], max_load_factor=got["max_load_factor"])
    a = np.frombuffer(open(ref_path, "rb").read(), np.uint8); b = np.frombuffer(open(our_path, "rb").read(), np.uint8)
    assert a.size == b.size == 24 + 24 * n and np.array_equal(a[:24], b[:24])
    ra = a[24:].reshape(n, 24); rb = b[24:].reshape(n, 24)
    assert np.array_equal(ra[:, :23], rb[:, :23])
    back = ref_py.tsdf_load(our_path)
    assert back["voxel_size"] == got["voxel_size"] and back["truncation"] == got["truncation"] and back["integration_weight_sample"] == got["integration_weight_sample"]
    x, y = _by_key(back), _by_key(rec)                                      # a re-filled hash map iterates in its own order
    for k in ("keys", "sdf", "weight", "color"):
        assert np.array_equal(x[k], y[k]), k
    assert ref_py.tsdf_load(tmp_path / "missing.tsdf") is None
    with pytest.raises(binding.I3DError):
        binding.tsdf_read(tmp_path / "missing.tsdf")


def test_voxel_sbr_files_equal_the_reference_writer_and_reader(R, tmp_path):
    """SparseVoxelGrid<VoxelSBR>::save / load of the reference (the per-level dumps Intrinsic3D writes) vs i3d_sbr_write / i3d_sbr_read."""
    from intrinsic3d_amd import binding
    from oracle import ref_py
    sc = _mangled_scene(6)
    g = R.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    e = g.export(); n = len(e["sdf"]); rng = np.random.default_rng(0)
    g.import_fields(e["sdf"] + rng.normal(0, 1e-4, n), rng.uniform(0.1, 0.9, n), None); e = g.export()
    ref_path = tmp_path / "ref.sbr"; ref_py.sbr_save(g, ref_path)
    got = binding.sbr_read(ref_path)
    for k in ("keys", "sdf", "sdf_refined", "albedo", "weight", "color"):
        assert np.array_equal(got[k], e[k]), k
    our_path = tmp_path / "ours.sbr"
    hdr = binding.tsdf_read.__globals__["C"]                                # header floats come back through the generic header reader
    vs, tr, iw, ml = hdr.c_float(), hdr.c_float(), hdr.c_float(), hdr.c_float(); cnt = hdr.c_uint64()
    assert binding.load().i3d_tsdf_read_header(str(ref_path).encode(), hdr.byref(vs), hdr.byref(tr), hdr.byref(iw), hdr.byref(cnt), hdr.byref(ml)) == 0
    binding.sbr_write(our_path, vs.value, got, truncation=tr.value, integration_weight_sample=iw.value, max_load_factor=ml.value)
    a = np.frombuffer(open(ref_path, "rb").read(), np.uint8); b = np.frombuffer(open(our_path, "rb").read(), np.uint8)
    assert a.size == b.size == 24 + 44 * n and np.array_equal(a[:24], b[:24])
    ra = a[24:].reshape(n, 44); rb = b[24:].reshape(n, 44)
    named = np.r_[0:12, 12:20, 20:24, 24:27, 28:36, 36:44]                  # key | sdf | weight | colour | albedo | sdf_refined; byte 27 is padding
    assert np.array_equal(ra[:, named], rb[:, named])
    back = ref_py.sbr_load(our_path); x, y = _by_key(back.export()), _by_key(e)
    for k in ("keys", "sdf", "sdf_refined", "albedo", "weight", "color"):
        assert np.array_equal(x[k], y[k]), k
    assert x["voxel_size"] == y["voxel_size"]


def test_intrinsics_files_equal_the_reference_writer_and_reader(R, tmp_path):
    """Camera::save / Camera::load (camera.cpp:202-274) of the reference vs i3d_write_intrinsics / i3d_read_intrinsics: same text, same parsed floats,
    same defaults when the file is missing."""
    from intrinsic3d_amd import binding
    from oracle import ref_py
    rng = np.random.default_rng(4)
    cases = [(640, 480, [525.0, 525.0, 319.5, 239.5], [0, 0, 0, 0, 0]),
             (1296, 968, [1170.187988, 1170.187988, 647.75, 483.75], [0.01, -0.0234567891, 0.0, 1e-5, -3.3e-7]),
             (320, 240, [1e6 + 0.5, 123456.789, 0.000123456, 1e-7], [1e10, -1e-10, 123456789.0, 0.1, 100000.0])]
    for _ in range(20):
        cases.append((int(rng.integers(1, 4000)), int(rng.integers(1, 4000)), (rng.uniform(50, 3000, 4) * rng.choice([1, 1e-3, 1e3], 4)).tolist(),
                      (rng.normal(0, 0.1, 5) * rng.choice([1, 1e-5, 1e4], 5)).tolist()))
    for i, (w, h, k, d) in enumerate(cases):
        a = tmp_path / f"ref_{i}.txt"; b = tmp_path / f"ours_{i}.txt"
        assert ref_py.camera_save(a, w, h, k, d)
        binding.write_intrinsics(b, w, h, k, d)
        assert open(a).read() == open(b).read(), (i, open(a).read(), open(b).read())
        ok_r, wr, hr, kr, dr = ref_py.camera_load(b); ok_o, wo, ho, ko, do = binding.read_intrinsics(a)
        assert ok_r and ok_o and (wr, hr) == (wo, ho) == (w, h)
        assert np.array_equal(kr, np.float32(ko)) and np.array_equal(dr, np.float32(do))
    ok_r, wr, hr, kr, dr = ref_py.camera_load(tmp_path / "nope.txt"); ok_o, wo, ho, ko, do = binding.read_intrinsics(tmp_path / "nope.txt")
    assert not ok_r and not ok_o and np.array_equal(kr, np.float32(ko)) and np.array_equal(dr, np.float32(do)) and (wr, hr) == (wo, ho) == (640, 480)
    # the one deliberate difference: a file that ends early.  The reference's stream reads are unchecked (it reports success, the last value read repeated in the missing entries);
    # the product reports I3D_ERR_IO and hands back the defaults
    open(tmp_path / "short.txt", "w").write("640 480\n525 0 319.5\n0 525")
    ok_r, _, _, kr, _ = ref_py.camera_load(tmp_path / "short.txt"); ok_o, _, _, ko, _ = binding.read_intrinsics(tmp_path / "short.txt")
    assert ok_r and kr[3] == 525.0 and not ok_o and list(ko) == [525.0, 525.0, 319.5, 239.5]


def test_keyframe_selection_equals_the_reference_class(R, tmp_path):
    """KeyframeSelection (keyframe_selection.cpp:46-126, 139-310) of the reference vs the product's host entry points: window selection incl. ties, all-zero
    and ragged last windows; keyframes.txt written byte for byte and parsed the same; the Crete blur metric on colour and grey images (its OpenCV calls are
    stand-ins on the reference side — what is pinned is the metric's own loops and normalisation)."""
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    rng = np.random.default_rng(11)
    for n, win in ((47, 10), (10, 10), (9, 10), (1, 3), (64, 7), (30, 1)):
        scores = rng.uniform(0, 1, n)
        if n > 20:
            scores[5:9] = scores[5]                      # ties inside a window: the first maximum wins
            scores[10:20] = 0.0                          # an all-zero window selects its first frame
        a = B.keyframes_select(win, scores); b = ref_py.keyframes_select(win, scores)
        assert np.array_equal(a, b), (n, win)
        pa = str(tmp_path / f"ours_{n}_{win}.txt"); pb = str(tmp_path / f"ref_{n}_{win}.txt")
        B.keyframes_save(pa, win, scores, a); assert ref_py.keyframes_save(pb, win, scores, b)
        assert open(pa).read() == open(pb).read()
        wa, sa, ka = B.keyframes_load(pb); wb, sb, kb = ref_py.keyframes_load(pa)
        assert wa == wb == win and np.array_equal(sa, sb) and np.array_equal(ka, kb) and np.array_equal(ka, a)
    assert ref_py.keyframes_load(str(tmp_path / "missing.txt")) is None
    with pytest.raises(B.I3DError):
        B.keyframes_load(str(tmp_path / "missing.txt"))
    # blur metric: smooth, sharp, noisy, tiny and single-row images; colour and grey
    yy, xx = np.mgrid[0:61, 0:83]
    imgs = [(127 + 100 * np.sin(xx / 3.0) * np.cos(yy / 5.0)).astype(np.uint8), rng.integers(0, 256, (61, 83)).astype(np.uint8),
            ((xx // 8 + yy // 8) % 2 * 255).astype(np.uint8), rng.integers(0, 256, (12, 7)).astype(np.uint8), rng.integers(0, 256, (1, 40)).astype(np.uint8)]
    imgs += [np.stack([im, np.roll(im, 3, 1), 255 - im], -1) for im in imgs[:4]]
    for im in imgs:
        sa, sb = B.blur_score(im), ref_py.blur_score(im)
        assert (np.isnan(sa) and np.isnan(sb)) or abs(sa - sb) <= 1e-12 * max(1.0, abs(sb)), (im.shape, sa, sb)


def test_loose_component_removal_equals_the_reference_code(R):
    """MeshUtil::removeLooseComponents + removeUnusedVertices (mesh/util.cpp:47-171, Boost's graph pieces as stand-ins) vs i3d_mesh_remove_loose_components —
    what `largest_component_only` applies to an extracted mesh: several components of different and of EQUAL size (the first one wins), components that
    touch in a single vertex, vertices no face uses, with and without colours."""
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    rng = np.random.default_rng(3)

    def blob(n_faces, v0, chain=True):
        """n_faces triangles, consecutive ones sharing an edge (one component), on fresh vertices starting at index v0"""
        nv = n_faces + 2; f = np.array([[v0 + i, v0 + i + 1, v0 + i + 2] for i in range(n_faces)], np.int32)
        return nv, f
    for trial, sizes in enumerate(([5, 9, 3], [4, 7, 7, 2], [6], [3, 3, 3], [1, 12, 12, 5, 12])):
        faces = []; nv = 0
        for s in sizes:
            n, f = blob(s, nv); faces.append(f); nv += n
            nv += int(rng.integers(0, 3))                                  # vertices no face uses, between the components
        faces = np.concatenate(faces)
        if trial == 1:                                                      # two components touching in ONE vertex become one
            faces[faces == faces[4][0]] = faces[0][0]
        order = rng.permutation(len(faces)) if trial in (0, 3, 4) else np.arange(len(faces))      # interleave the components' faces in the file order
        faces = faces[order]
        verts = rng.normal(0, 1, (nv, 3)).astype(np.float32); cols = rng.integers(0, 256, (nv, 3)).astype(np.uint8)
        for c in (cols, None):
            vo, co, fo = B.mesh_remove_loose_components(verts, c, faces)
            vr, cr, fr = ref_py.mesh_remove_loose_components(verts, c, faces)
            assert np.array_equal(vo, vr) and np.array_equal(fo, fr) and (c is None or np.array_equal(co, cr)), (trial, len(vo), len(vr), len(fo), len(fr))
            assert len(fo) >= max(sizes) and fo.max() == len(vo) - 1
    # an unmerged triangle soup: every face owns three vertices, neighbouring faces share POSITIONS only.  The reference connects faces through its
    # position-keyed vertex map (mesh/util.cpp:52-62), so the soup's components are those of the merged mesh
    n0, f0 = blob(9, 0); n1, f1 = blob(4, n0)
    merged = np.concatenate([f0, f1]); pos = rng.normal(0, 1, (n0 + n1, 3)).astype(np.float32)
    soup_v = pos[merged.reshape(-1)]; soup_f = np.arange(3 * len(merged), dtype=np.int32).reshape(-1, 3)
    vo, _, fo = B.mesh_remove_loose_components(soup_v, None, soup_f)
    vr, _, fr = ref_py.mesh_remove_loose_components(soup_v, None, soup_f)
    assert len(fo) == 9 and np.array_equal(vo, vr) and np.array_equal(fo, fr)


def test_dataset_folder_sensor_equals_the_reference_classes(R, tmp_path):
    """SensorI3d::init / listFiles / loadPose / loadIntrinsics + Sensor::depth (millimetre scale, min / max thresholds) / color / pose of the reference
    (PNG decoding handed to Pillow through cv::imdecode's hook) vs the product's i3d_sensor_*: the frames listed and stored, image sizes, intrinsics,
    poses, thresholded depth and colour, for a full folder, a frame limit, a gap in the numbering and a missing pose file."""
    import shutil
    from PIL import Image
    from scipy.spatial.transform import Rotation
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    rng = np.random.default_rng(8); folder = tmp_path / "rgbd"; folder.mkdir()
    Kc = np.eye(4); Kc[0, 0] = 570.3; Kc[1, 1] = 571.1; Kc[0, 2] = 31.5; Kc[1, 2] = 23.25
    Kd = np.eye(4); Kd[0, 0] = 285.7; Kd[1, 1] = 286.2; Kd[0, 2] = 15.5; Kd[1, 2] = 11.75
    np.savetxt(folder / "colorIntrinsics.txt", Kc); np.savetxt(folder / "depthIntrinsics.txt", Kd)
    for i in range(6):
        Image.fromarray(rng.integers(0, 256, (48, 64, 3), np.uint8)).save(folder / f"frame-{i:06d}.color.png")
        Image.fromarray(rng.integers(0, 4000, (24, 32)).astype(np.uint16)).save(folder / f"frame-{i:06d}.depth.png")
        T = np.eye(4); T[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * 0.4).as_matrix(); T[:3, 3] = rng.normal(size=3)
        np.savetxt(folder / f"frame-{i:06d}.pose.txt", T)

    def compare(fd, max_frames=0, dmin=0.0, dmax=0.0):
        a = B.Sensor(fd, max_frames, dmin, dmax); b = ref_py.Sensor(fd, max_frames, dmin, dmax)
        assert (a.num_frames, a.num_loaded) == (b.num_frames, b.num_stored)
        assert a.color_size == b.color_size and a.depth_size == b.depth_size
        assert np.array_equal(a.color_intrinsics, b.color_intrinsics) and np.array_equal(a.depth_intrinsics, b.depth_intrinsics)
        for i in range(b.num_stored):
            assert np.array_equal(a.pose(i), b.pose(i)) and np.array_equal(a.depth(i), b.depth(i)) and np.array_equal(a.color(i), b.color(i)), i
        assert np.array_equal(a.pose(99), b.pose(99))
        n = (a.num_frames, a.num_loaded); a.close(); b.close()
        return n
    assert compare(folder) == (6, 6)
    assert compare(folder, 0, 0.5, 2.5) == (6, 6)                       # depth thresholds
    assert compare(folder, 2) == (6, 2)                                 # num_frames_max: everything is listed, two frames are stored
    f2 = tmp_path / "gap"; shutil.copytree(folder, f2); (f2 / "frame-000004.depth.png").unlink()
    assert compare(f2) == (4, 4)                                        # listFiles stops at the first missing depth map
    f3 = tmp_path / "nopose"; shutil.copytree(folder, f3); (f3 / "frame-000003.pose.txt").unlink()
    assert compare(f3) == (6, 3)                                        # a missing pose file ends the loading loop, the listing keeps its count
    f4 = tmp_path / "nocolor"; shutil.copytree(folder, f4); (f4 / "frame-000002.color.png").unlink()
    assert compare(f4) == (6, 2)                                        # a missing colour image as well
    # Sensor::savePoses: TUM trajectory lines (timestamp, camera-to-world translation, Eigen::Quaternionf(R) as x y z w, 6 decimals) — the same text from both,
    # incl. a rotation close to 180 degrees (negative-trace branch of the conversion); and the reference's loadPoses reads the product's file back
    a = B.Sensor(folder); b = ref_py.Sensor(folder)
    T = a.pose(1).copy(); T[:3, :3] = Rotation.from_rotvec([3.1, 0.02, -0.01]).as_matrix().astype(np.float32); a.set_pose(1, T); b.set_pose(1, T)
    a.save_poses(tmp_path / "poses_ours.txt"); assert b.save_poses(tmp_path / "poses_ref.txt")
    assert open(tmp_path / "poses_ours.txt").read() == open(tmp_path / "poses_ref.txt").read()
    ts, mats = ref_py.load_poses(tmp_path / "poses_ours.txt")
    assert np.array_equal(ts, np.arange(6.0)) and all(np.abs(mats[i] - a.pose(i)).max() < 5e-6 for i in range(6))
    a.close(); b.close()
    f5 = tmp_path / "empty"; shutil.copytree(folder, f5); open(f5 / "frame-000001.color.png", "wb").close()
    assert compare(f5) == (6, 1)                                        # so does an EMPTY file (loadFile reports size 0 as failure: the `continue` for empty buffers is never reached)


def test_fusion_application_equals_the_reference_code(oracle, R, tmp_path):
    """AppFusion::fuseSDF (apps/src/app_fusion.cpp:107-200) of the reference — keyframe filter, erosion, normals, integrate, correctSDF, clearInvalidVoxels,
    the .tsdf and the mesh — run on a dataset folder through its own SensorI3d, against the oracle's fusion of the same decoded frames (which the GPU
    suite holds apps/app_fusion to, byte for byte): records in file order bit-exact, with and without a keyframe file that drops frames.  Six cameras on
    the coordinate axes: their rotations are signed permutations, so the 4x4 pose inverse is exact whatever its operation order (Eigen's is unpinned)."""
    import pathlib
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    folder, vs, n = helpers.axis_camera_dataset(tmp_path); folder = pathlib.Path(folder)
    counts = []
    for use_kf in (False, True):
        kf_file = ""; keep = [True] * n
        if use_kf:
            keep = [True, False, True, True, False, True]; kf_file = str(tmp_path / "fusion" / "keyframes.txt")
            assert ref_py.keyframes_save(kf_file, 1, np.ones(n), keep)
        out_sdf = tmp_path / "fusion" / f"ref_{int(use_kf)}.tsdf"; out_ply = tmp_path / "fusion" / f"ref_{int(use_kf)}.ply"
        cfg = {"keyframes": kf_file, "voxel_size": repr(vs), "clip_x0": 0, "clip_x1": 0, "clip_y0": 0, "clip_y1": 0, "clip_z0": 0, "clip_z1": 0,
               "discont_window_size": 2, "output_sdf": str(out_sdf), "output_mesh": str(out_ply)}
        assert ref_py.app_fusion(folder, cfg, 0, 0.05, 10.0)
        vol = B.tsdf_read(out_sdf)
        s = B.Sensor(folder, 0, 0.05, 10.0)                             # (held to the reference's SensorI3d by test_dataset_folder_sensor_...)
        o = oracle.Fusion(np.float32(vs), 0.05, 10.0, np.zeros(6, np.float32))
        for i in range(s.num_frames):
            if keep[i]:
                o.integrate(s.depth(i), s.depth_intrinsics, s.color(i), s.color_intrinsics, s.pose(i), 2)
        o.finish(10); ref = o.export(); s.close()
        for k in ("keys", "sdf", "weight", "color"):
            assert np.array_equal(vol[k], ref[k]), (use_kf, k)
        counts.append(len(ref["sdf"]))
        assert open(out_ply, "rb").read(3) == b"ply" and out_ply.stat().st_size > 10000
    assert counts[0] > 3000 and counts[1] > 2000 and counts[0] != counts[1]


def test_keyframe_application_equals_the_reference_code(R, tmp_path):
    """apps/app_keyframes (host only) against the reference's AppKeyframes::selectKeyframes (apps/src/app_keyframes.cpp:101-144: the blur score of EVERY frame
    of the sensor, window selection, keyframes.txt) on the same dataset folder: the file, byte for byte."""
    import subprocess
    from PIL import Image, ImageFilter
    from oracle import ref_py
    exe = os.path.join(ROOT, "apps", "app_keyframes")
    if not os.path.exists(exe):
        pytest.skip("apps/app_keyframes has not been built")
    rng = np.random.default_rng(21); folder = tmp_path / "rgbd"; folder.mkdir(); (tmp_path / "fusion").mkdir()
    np.savetxt(folder / "colorIntrinsics.txt", np.eye(4)); np.savetxt(folder / "depthIntrinsics.txt", np.eye(4))
    yy, xx = np.mgrid[0:60, 0:80]
    for i in range(13):
        base = (127 + 90 * np.sin(xx / (2.0 + 0.3 * i)) * np.cos(yy / 3.0))[..., None] + rng.normal(0, 12, (60, 80, 3))
        im = Image.fromarray(np.clip(base, 0, 255).astype(np.uint8))
        if i % 3 == 1:
            im = im.filter(ImageFilter.GaussianBlur(1.0 + 0.2 * i))                 # some frames blurred: they must lose their window
        im.save(folder / f"frame-{i:06d}.color.png")
        Image.fromarray(rng.integers(500, 3000, (30, 40)).astype(np.uint16)).save(folder / f"frame-{i:06d}.depth.png")
        np.savetxt(folder / f"frame-{i:06d}.pose.txt", np.eye(4))
    (tmp_path / "sensor.yml").write_text('%YAML:1.0\n\ndataset: "./rgbd/"\nmax_frames: "0"\nmin_depth: "0.1"\nmax_depth: "10.0"\n')
    for win in (5, 1, 13):
        (tmp_path / "keyframes.yml").write_text(f'%YAML:1.0\n\nwindow_size: "{win}"\nfilename: "./fusion/keyframes.txt"\nshow_keyframes: "0"\n')
        r = subprocess.run([exe, "-s", str(tmp_path / "sensor.yml"), "-k", str(tmp_path / "keyframes.yml")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        ours = open(tmp_path / "fusion" / "keyframes.txt").read()
        assert ref_py.app_keyframes(folder, {"filename": str(tmp_path / "fusion" / "ref.txt"), "window_size": win, "show_keyframes": 0}, 0, 0.1, 10.0)
        assert ours == open(tmp_path / "fusion" / "ref.txt").read(), win
        assert len(ours.splitlines()) == 14 and ours.splitlines()[0] == str(win)


def test_yaml_configuration_equals_the_reference_loaders(R, tmp_path):
    """Intrinsic3D::Config::load + Optimizer::Config::load (intrinsic3d.cpp:58-80, optimizer.cpp:52-72, over the reference's Settings) vs i3d_config_load_yaml:
    every key of the shipped data/intrinsic3d.yml (incl. its `subvolume_sh_lamda_reg` spelling) lands in the same field with the same value, for the shipped
    values and for a second set that exercises fractions, exponents, negative and boolean entries."""
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    shipped = {"num_grid_levels": "3", "num_rgbd_levels": "3", "thin_shell_factor": "2.0", "thin_shell_factor_final": "1.0", "clear_distant_voxels": "1",
               "occlusion_distance": "0.02", "num_observations": "5", "subvolume_size_sh": "0.2", "subvolume_sh_lamda_reg": "10.0",
               "iterations": "10", "lm_steps": "50", "lambda_g": "0.2", "lambda_r0": "80.0", "lambda_r1": "10.0", "lambda_s0": "120.0", "lambda_s1": "10.0", "lambda_a": "0.1",
               "fix_poses": "0", "fix_intrinsics": "0", "fix_distortion": "0"}
    other = dict(shipped, num_grid_levels="1", num_rgbd_levels="2", thin_shell_factor="2.5", thin_shell_factor_final="0.75", clear_distant_voxels="0", occlusion_distance="1e-2",
                 num_observations="0", subvolume_size_sh="0.035", subvolume_sh_lamda_reg="1.5e1", iterations="3", lm_steps="7", lambda_g="1.25", lambda_r0="2e1", lambda_r1="160", lambda_a="-1.0",
                 fix_poses="1", fix_intrinsics="1", fix_distortion="1")
    for i, cfg in enumerate((shipped, other)):
        yml = tmp_path / f"intrinsic3d_{i}.yml"
        yml.write_text("%YAML:1.0\n\n# Intrinsic3D config\n" + "".join(f'# comment for {k}\n{k}: "{v}"\n' for k, v in cfg.items()) + 'keyframes: "./fusion/keyframes.txt"\noutput_mesh_prefix: "./intrinsic3d/mesh"\n')
        rc, oc = B.load_yaml_config(yml)
        ref = ref_py.config_load(cfg)
        ours = {k: float(getattr(rc, k)) for k, _ in B.RefineConfig._fields_}
        ours.update({k: float(getattr(oc, k)) for k in ("iterations", "lm_steps", "lambda_g", "lambda_r0", "lambda_r1", "lambda_s0", "lambda_s1", "lambda_a", "fix_poses", "fix_intrinsics", "fix_distortion")})
        for k, v in ref.items():
            assert ours[k] == v, (i, k, ours[k], v)
        assert float(oc.occlusion_distance) == ref["occlusion_distance"] and float(oc.num_observations) == ref["num_observations"]


def test_refinement_initialisation_equals_the_reference_code(oracle, R, tmp_path):
    """Intrinsic3D::init (intrinsic3d.cpp:151-203) of the reference on a dataset folder — the keyframe loop over its own SensorI3d / KeyframeSelection, resizeDepth
    into the colour camera, Pyramid(num_levels, colour, depth) and the pose inverse + math::poseMatToVecAA — against what the product's
    i3d_init_frames_from_sensor is built from and held to on the device: the oracle's pyramid primitives composed in the same order, and the host-side
    i3d_pose_mat_to_vec6.  Depth camera at half the colour resolution; three pyramid levels; a keyframe list that skips frames."""
    from PIL import Image
    from scipy.spatial.transform import Rotation
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    rng = np.random.default_rng(31); folder = tmp_path / "rgbd"; folder.mkdir()
    Kc = np.eye(4); Kc[0, 0] = 105.0; Kc[1, 1] = 104.0; Kc[0, 2] = 47.5; Kc[1, 2] = 35.5
    Kd = np.eye(4); Kd[0, 0] = 52.5; Kd[1, 1] = 52.0; Kd[0, 2] = 23.5; Kd[1, 2] = 17.5
    np.savetxt(folder / "colorIntrinsics.txt", Kc); np.savetxt(folder / "depthIntrinsics.txt", Kd)
    n = 5
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (72, 96, 3), np.uint8)).save(folder / f"frame-{i:06d}.color.png")
        d = rng.integers(300, 3500, (36, 48)).astype(np.uint16); d[rng.random(d.shape) < 0.2] = 0
        Image.fromarray(d).save(folder / f"frame-{i:06d}.depth.png")
        T = np.eye(4); T[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * (3.0 if i == 3 else 0.5)).as_matrix(); T[:3, 3] = rng.normal(size=3)
        np.savetxt(folder / f"frame-{i:06d}.pose.txt", T)
    keep = [True, False, True, True, False]; levels = 3
    m = ref_py.InitModel(folder, keep, levels, 0, 0.4, 3.0)
    s = B.Sensor(folder, 0, 0.4, 3.0)
    assert list(m.frame_ids) == [i for i in range(n) if keep[i]]
    assert np.array_equal(m.intrinsics, np.float32(s.color_intrinsics).astype(np.float64)) and not m.distortion.any()
    for k, f in enumerate(m.frame_ids):
        # (to round-off, not bit for bit: the 4x4 inverse in front of the conversion is Eigen's, whose operation order neither side reproduces)
        assert np.abs(m.poses[k] - B.pose_mat_to_vec6(s.pose(int(f)))).max() <= 1e-13, (k, m.poses[k] - B.pose_mat_to_vec6(s.pose(int(f))))
        bgr = s.color(int(f)); assert np.array_equal(m.image(k, 0, "bgr"), bgr)
        lum = oracle.lum_from_bgr(bgr)
        dep = oracle.resize_depth(s.depth(int(f)), s.depth_intrinsics, s.color_size[0], s.color_size[1], s.color_intrinsics)
        for l in range(levels):
            assert np.array_equal(m.image(k, l, "lum"), lum), (k, l)
            assert np.array_equal(m.image(k, l, "depth"), dep), (k, l)
            lum = oracle.pyr_down(lum); dep = oracle.depth_down(dep)
        assert m.image(k, levels, "lum") is None
    m.close(); s.close()


def _read_ply(path):
    raw = open(path, "rb").read(); end = raw.index(b"end_header\n") + 11
    head = raw[:end].decode().split("\n"); nv = int(head[2].split()[2]); nf = int([l for l in head if l.startswith("element face")][0].split()[2])
    vt = np.frombuffer(raw, np.dtype([("p", "<f4", 3), ("c", "u1", 3)]), nv, end)
    ft = np.frombuffer(raw, np.dtype([("n", "u1"), ("i", "<i4", 3)]), nf, end + nv * 15)
    assert end + nv * 15 + nf * 13 == len(raw) and (ft["n"] == 3).all()
    return vt["p"].copy(), vt["c"].copy(), ft["i"].copy()


APP_CASES = {
    # grid levels, pyramid levels, iterations, non-keyframes appended, half-resolution depth camera, intrinsic3d.yml overrides
    "two_levels": (2, 2, 1, 0, False, {}),
    "three_levels_fixed_distortion_skipped_frames": (3, 3, 2, 2, False, {"fix_distortion": 1}),
    "constant_albedo_no_clearing_half_res_depth": (2, 1, 1, 0, True, {"lambda_a": -1.0, "clear_distant_voxels": 0, "thin_shell_factor_final": 0.0, "num_observations": 3,
                                                                      "occlusion_distance": 0.01, "subvolume_size_sh": 0.03, "fix_poses": 1}),
}


@pytest.mark.parametrize("case", list(APP_CASES))
def test_refinement_application_equals_the_oracle_pipeline(oracle, R, tmp_path, case):
    """AppIntrinsic3D::run + onSDFRefined (apps/src/app_intrinsic3d.cpp:71-210) of the reference, compiled into oracle/_ref over its own SensorI3d, KeyframeSelection,
    SparseVoxelGrid::load / create(tsdf -> sbr), Intrinsic3D::init / refine, SDFVisualization and MarchingCubes, on a dataset folder in the reference's layout —
    against the oracle's pieces composed from the SAME files: which output files appear under which names, the final poses / intrinsics text, and the last level's
    mesh in both colour modes.  Cases: the level schedule (2 x (2, 1) and 3 x (3, 2, 1) stages), frames that are not keyframes, a depth camera at half the colour
    resolution (resizeDepth proper), fixed parameter groups, constant albedo, no voxel clearing with a constant shell factor, 3 observations, small subvolumes."""
    from PIL import Image
    from intrinsic3d_amd import binding as B, synthetic
    from oracle import ref_py
    import make_dataset
    GL, PL, iters, extra, half_depth, over = APP_CASES[case]
    sc = synthetic.make_scene(radius_vox=10, K=4, width=96, height=72, levels=1, seed=9, pose_noise=(0.0005, 0.001), lum_noise=0.003)
    s_yml, i_yml = make_dataset.write_dataset(str(tmp_path), sc, grid_levels=GL, rgbd_levels=PL, iterations=iters, extra_frames=extra, **over)
    if half_depth:                                                                      # depth maps at 48 x 36 with their own intrinsics
        K = np.loadtxt(tmp_path / "rgbd" / "depthIntrinsics.txt"); K[:2, :3] *= 0.5; np.savetxt(tmp_path / "rgbd" / "depthIntrinsics.txt", K, fmt="%.9g")
        for f in sorted((tmp_path / "rgbd").glob("*.depth.png")):
            Image.fromarray(np.asarray(Image.open(f))[::2, ::2].copy()).save(f)
    cfg = dict(re.findall(r'^(\w+): "(.*)"$', open(i_yml).read(), re.M))
    (tmp_path / "intrinsic3d").mkdir(exist_ok=True)
    cwd = os.getcwd(); os.chdir(tmp_path)                                              # (the reference application changes into the sensor config's directory)
    try:
        assert ref_py.app_intrinsic3d("./rgbd/", cfg, 0, 0.1, 10.0)
    finally:
        os.chdir(cwd)
    out = tmp_path / "intrinsic3d"
    # the coarsest grid level runs every pyramid level, the finer ones only the finest (intrinsic3d.cpp:243-246); level numbers count down to 0
    stages = [f"g{GL - 1}_p{p}" for p in range(PL - 1, -1, -1)] + [f"g{g}_p0" for g in range(GL - 2, -1, -1)]
    assert sorted(os.listdir(out)) == sorted(f"{p}_{s}{e}" for s in stages for p, e in (("intrinsics", ".txt"), ("poses", ".txt"), ("mesh", ".ply"), ("mesh", "_albedo.ply")))

    # the same run on the oracle, from the same files
    s = B.Sensor(tmp_path / "rgbd", 0, 0.1, 10.0)
    kf = B.keyframes_load(str(tmp_path / "fusion" / "keyframes.txt"))[2]; assert kf.sum() == 4 and len(kf) == 4 + extra
    vol = B.tsdf_read(str(tmp_path / "fusion" / os.path.basename(cfg["input_sdf"])))
    # (the start poses are taken from the reference's own init: this little scene leaves poses and distortion weakly determined, and the 1e-13 by which Eigen's 4x4
    #  inverse differs from the product's — test_refinement_initialisation — grows to 1e-2 through three optimisations)
    m = ref_py.InitModel(tmp_path / "rgbd", kf, PL, 0, 0.1, 10.0); poses = np.array(m.poses); m.close()
    frames = []
    for f in range(s.num_frames):
        if not kf[f]:
            continue
        bgr = s.color(f); lum = [oracle.lum_from_bgr(bgr)]
        dep = [oracle.resize_depth(s.depth(f), s.depth_intrinsics, s.color_size[0], s.color_size[1], s.color_intrinsics)]
        for _ in range(1, PL):
            lum.append(oracle.pyr_down(lum[-1])); dep.append(oracle.depth_down(dep[-1]))
        frames.append({"lum": lum, "depth": dep, "bgr": [bgr] + [bgr[::2 ** l, ::2 ** l] for l in range(1, PL)]})
    ci = np.float64(s.color_intrinsics); w, h = s.color_size; n = s.num_frames; s.close()
    g = oracle.Grid.from_voxels(vol["voxel_size"], vol["keys"], vol["sdf"], vol["weight"], vol["color"]); fr = oracle.Frames(frames, PL)
    ocfg = helpers.oracle_cfg(oracle, 0.0, iterations=iters, **{k: float(cfg[k]) for k in ("lambda_g", "lambda_r0", "lambda_r1", "lambda_s0", "lambda_s1", "lambda_a")}, occlusion_distance=float(np.float32(cfg["occlusion_distance"])),
                              lm_steps=int(cfg["lm_steps"]), num_observations=int(cfg["num_observations"]), **{k: int(cfg[k]) for k in ("fix_poses", "fix_intrinsics", "fix_distortion")})
    rc, intr, dist, pose6, done = oracle.refine(g, fr, ocfg, GL, PL, float(cfg["thin_shell_factor"]), float(cfg["thin_shell_factor_final"]), int(cfg["clear_distant_voxels"]),
                                                float(np.float32(cfg["subvolume_size_sh"])), float(cfg["subvolume_sh_lamda_reg"]), ci, np.zeros(5), np.array(poses))
    assert rc == 0 and done == len(stages)
    # poses / intrinsics files of the last stage (6 / default-precision decimals in the text); frames that are not keyframes keep the pose they were loaded with
    a = np.loadtxt(out / "poses_g0_p0.txt"); assert a.shape == (n, 8)
    ids = np.flatnonzero(kf)
    B.write_poses(str(tmp_path / "ours_poses.txt"), ids.astype(np.float64), pose6)
    b = np.loadtxt(tmp_path / "ours_poses.txt").reshape(len(ids), 8)
    assert np.abs(a[ids] - b).max() <= 2e-6, np.abs(a[ids] - b).max()
    first = np.loadtxt(out / f"poses_{stages[0]}.txt")
    assert (np.abs(a - first).max() > 1e-5) == (int(cfg["fix_poses"]) == 0)                 # ... and the poses did move between the stages, unless they are fixed
    if extra:
        assert np.array_equal(a[~kf.astype(bool)], first[~kf.astype(bool)])
    cam = ref_py.camera_load(str(out / "intrinsics_g0_p0.txt"))
    assert cam[0] and (cam[1], cam[2]) == (w, h) and np.allclose(cam[3], intr, rtol=1e-5, atol=0) and np.allclose(cam[4], dist, rtol=1e-5, atol=1e-9)   # (six significant digits in the file)
    assert (np.abs(dist).max() == 0.0) == (int(cfg["fix_distortion"]) == 1)
    # the last level's meshes: MarchingCubes over the refined distances, largest component, voxel colours / albedo colours
    rv, rcol, rf = _read_ply(out / "mesh_g0_p0.ply"); av, acol, af = _read_ply(out / "mesh_g0_p0_albedo.ply")
    ov, ocol, of = oracle.marching_cubes(g, True)
    ov2, ocol2, of2 = B.mesh_remove_loose_components(ov, ocol, of)
    assert np.array_equal(av, rv) and np.array_equal(af, rf)
    e = g.export(); g.import_fields(color=ref_py.albedo_colors(e["albedo"]))
    _, ac2, _ = B.mesh_remove_loose_components(*oracle.marching_cubes(g, True))
    if rv.shape == ov2.shape:
        assert np.array_equal(rf, of2) and np.abs(rv - ov2).max() <= 1e-6
        assert (np.abs(rcol.astype(int) - ocol2.astype(int)) > 1).mean() < 1e-3             # colours interpolate between voxels whose order of first use may differ in the last bit
        assert (np.abs(acol.astype(int) - ac2.astype(int)) > 1).mean() < 1e-3
    else:
        # five optimisations by two LM implementations (1e-10 apart) later, a voxel or two fall on the other side of a threshold (thin shell / sign of the distance):
        # a handful of vertices exist on one side only; everything else is the same surface
        from scipy.spatial import cKDTree
        assert abs(len(rv) - len(ov2)) <= 1e-3 * len(rv) and abs(len(rf) - len(of2)) <= 1e-3 * len(rf)
        d, j = cKDTree(ov2).query(rv)
        assert (d <= 1e-6).mean() > 0.999 and (np.abs(rcol.astype(int) - ocol2[j].astype(int)).max(1)[d <= 1e-6] > 1).mean() < 1e-3
    assert (np.ptp(acol) > 1) == (float(cfg["lambda_a"]) >= 0)                              # constant albedo: one grey (152 / 153 after the float interpolation along the edges)
    g.free(); fr.free()


@pytest.mark.parametrize("seed", [1, 7])
def test_debug_colour_modes_equal_the_reference_visualization(R, seed):
    """SDFVisualization::applyColorNormals / Laplacian / Intensity / IntensityGradient / Albedo / Shading (both) / Chromacity (sdf/visualization.cpp:228-373, compiled into
    oracle/_ref with SDFOperators, Shading::computeShading and the reference's own Subvolumes) against i3d_visualization_colors — the host instantiation of the very
    function the export kernel runs (device/vis_colors.hpp) — on a blob with holes, zero-weight voxels, black / saturated colours, albedos outside [0, 1] and
    several subvolumes with missing neighbours: every voxel's colour, byte for byte."""
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    rng = np.random.default_rng(seed)
    vs = 0.004; r = 9
    g = np.stack(np.meshgrid(*[np.arange(-r, r + 1)] * 3, indexing="ij"), -1).reshape(-1, 3)
    d = np.linalg.norm(g + 0.3, axis=1) - 6.2
    keep = (np.abs(d) < 2.6) & (rng.random(len(g)) > 0.04)                              # a shell with holes
    keys = (g[keep] + np.array([40, -3, 7])).astype(np.int32); n = len(keys)
    keys = keys[rng.permutation(n)]
    sdf = (np.linalg.norm(keys - np.array([40, -3, 7]) + 0.3, axis=1) - 6.2) * vs + rng.normal(0, 0.15 * vs, n)
    w = rng.uniform(0.5, 30, n).astype(np.float32); w[rng.random(n) < 0.05] = 0.0       # invalid voxels
    alb = rng.uniform(-0.1, 1.2, n); alb[rng.random(n) < 0.02] = 0.0
    col = rng.integers(0, 256, (n, 3)).astype(np.uint8); col[rng.random(n) < 0.03] = 0; col[rng.random(n) < 0.03] = 255
    size = 0.0131                                                                        # subvolumes of 3.3 voxels
    sub = np.unique(np.floor((keys.astype(np.float32) * np.float32(vs)) * (np.float32(1.0) / np.float32(size))).astype(np.int32), axis=0)
    sub = sub[rng.permutation(len(sub))]
    sh = np.concatenate([rng.uniform(0.4, 1.1, (len(sub), 1)), rng.normal(0, 0.35, (len(sub), 8))], 1)
    for mode in ("normals", "lap", "lum", "lum_grad", "albedo", "chroma", "shading_sv", "shading_sv_const"):
        ref, rank = ref_py.visualization_colors(mode, vs, keys, sdf, alb, w, col, size, sub, sh)
        got = B.visualization_colors(mode, vs, keys, sdf, alb, w, col, size, sub, sh, visit_rank=rank)
        assert np.array_equal(got, ref), (mode, int((got != ref).any(1).sum()), n)
        if mode == "lum_grad":      # painted in place while the grid is walked: a voxel reads its +x neighbour REPAINTED if the walk passed that one earlier — the walk's order matters
            assert not np.array_equal(B.visualization_colors(mode, vs, keys, sdf, alb, w, col, size, sub, sh), ref) and sorted(rank) == list(range(n))
        assert len(np.unique(ref)) > (2 if mode == "lum_grad" else 20), mode             # the mode painted something
    assert len(sub) > 20
    k2 = keys + np.array([0, 30, 10], np.int32)                                          # all coordinates positive: ONE subvolume of 10 m, whose coefficients are used as they are
    one, _ = ref_py.visualization_colors("shading_sv", vs, k2, sdf, alb, w, col, 10.0, np.zeros((1, 3), np.int32), sh[:1])
    assert np.array_equal(B.visualization_colors("shading_sv", vs, k2, sdf, alb, w, col, 10.0, np.zeros((1, 3), np.int32), sh[:1]), one) and one.any()
    assert np.array_equal(B.visualization_colors("", vs, keys, sdf, alb, w, col), col)


def test_sensor_yml_equals_the_reference_factory(R, tmp_path):
    """Sensor::create(Settings&) (rgbd/sensor.cpp:64-118 over Settings::get<T>, settings.cpp:86-109) against i3d_sensor_open_yaml — what the three applications open
    their sensor.yml with: key names, missing keys, the stream conversions (leading number of "2abc", exponent notation, a folder name cut at its first blank)."""
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    folder, vs, n = helpers.axis_camera_dataset(tmp_path)
    cases = [{"dataset": str(folder), "max_frames": "3", "min_depth": "0.45", "max_depth": "2.5"},
             {"dataset": str(folder) + "/"},                                              # nothing else: no frame limit, no depth range
             {"dataset": str(folder), "max_frames": "2abc", "min_depth": "1e-1", "max_depth": "0.7000001", "unrelated": "x"},
             {"dataset": str(folder), "max_frames": "0", "min_depth": ".3", "max_depth": "10"}]
    for i, cfg in enumerate(cases):
        yml = tmp_path / f"sensor{i}.yml"
        yml.write_text("%YAML:1.0\n\n# rgbd sensor config\n" + "".join(f'{k}: "{v}"\n' for k, v in cfg.items()))
        r = ref_py.Sensor(cfg=cfg); s = B.Sensor(yml=yml)
        assert (s.num_frames, s.num_loaded) == (r.num_frames, r.num_stored), (i, s.num_frames, r.num_frames)
        assert s.depth_range == r.depth_range, (i, s.depth_range, r.depth_range)
        assert np.array_equal(s.depth(0), r.depth(0)) and np.array_equal(s.color(s.num_loaded - 1), r.color(r.num_stored - 1))
        s.close(); r.close()
    assert ref_py.Sensor(cfg=cases[2]).num_stored == 2 and ref_py.Sensor(cfg=cases[1]).num_stored == n
    spaced = tmp_path / "with blank"; os.symlink(folder, spaced)
    bad = {"dataset": str(spaced), "max_frames": "0"}                                    # Settings::get<std::string> stops at the blank: a folder that does not exist
    (tmp_path / "bad.yml").write_text("%YAML:1.0\n" + "".join(f'{k}: "{v}"\n' for k, v in bad.items()))
    r = ref_py.Sensor(cfg=bad); s = B.Sensor(yml=tmp_path / "bad.yml")                    # ... which both sides open as a sensor without frames (the applications stop there)
    assert (r.num_frames, r.num_stored) == (0, 0) == (s.num_frames, s.num_loaded)
    (tmp_path / "empty.yml").write_text("%YAML:1.0\n")                                    # Settings::empty(): no sensor
    with pytest.raises(B.I3DError):
        B.Sensor(yml=tmp_path / "empty.yml")


def test_intensity_gradient_view_with_long_repaint_chains():
    """"lum_grad" is painted in place in the reference (visualization.cpp:273-305): with a walk that runs against +x every voxel of a row reads a neighbour that was repainted
    just before — chains as long as the rows.  i3d_visualization_colors (out along +x, back along -x) against a literal in-place loop over the same walk."""
    from intrinsic3d_amd import binding as B
    rng = np.random.default_rng(3)
    g = np.stack(np.meshgrid(np.arange(40), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    keys = g[rng.random(len(g)) > 0.02]; n = len(keys)
    w = np.ones(n, np.float32); w[rng.random(n) < 0.02] = 0.0
    col = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    order = np.lexsort((keys[:, 2], keys[:, 1], -keys[:, 0]))                            # the walk: x descending
    rank = np.empty(n, np.int64); rank[order] = np.arange(n)
    got = B.visualization_colors("lum_grad", 0.004, keys, np.zeros(n), np.full(n, 0.6), w, col, visit_rank=rank)
    at = {tuple(k): i for i, k in enumerate(keys)}; cur = col.copy()
    f = np.float32
    lum = lambda c: f(f(f(0.299) * f(c[0]) + f(0.587) * f(c[1])) + f(0.114) * f(c[2]))
    for i in order:                                                                      # the reference's loop, literally
        x, y, z = keys[i]
        nb = [at.get((x + 1, y, z)), at.get((x - 1, y, z)), at.get((x, y + 1, z)), at.get((x, y - 1, z)), at.get((x, y, z + 1)), at.get((x, y, z - 1))]
        dx = f(0)
        if all(j is not None and w[j] > 0 for j in nb):
            dx = f(lum(cur[nb[0]]) - lum(cur[i]))
        cur[i] = np.uint8(min(max(f(dx * f(0.5) + f(127.0)), f(0)), f(255)))
    assert np.array_equal(got, cur)
    assert (np.abs(got[:, 0].astype(int) - 127) > 20).sum() > 20 and not np.array_equal(got, B.visualization_colors("lum_grad", 0.004, keys, np.zeros(n), np.full(n, 0.6), w, col))


def test_pose_vectors_of_special_rotations_equal_the_reference_initialisation(R, tmp_path):
    """pose file -> SensorI3d::loadPose -> inverse -> math::poseMatToVecAA (intrinsic3d.cpp:186-190, math.cpp:166-179) in the reference's own init against
    i3d_pose_mat_to_vec6 for the rotations where an angle-axis conversion has its corners: identity, half turns about each axis and about a diagonal, angles of
    1e-9 and 1e-5, a hair below pi, and diagonal sign matrices."""
    from PIL import Image
    from scipy.spatial.transform import Rotation
    from intrinsic3d_amd import binding as B
    from oracle import ref_py
    folder = tmp_path / "rgbd"; folder.mkdir()
    K = np.eye(4); K[0, 0] = K[1, 1] = 50; K[0, 2] = 15.5; K[1, 2] = 11.5
    np.savetxt(folder / "colorIntrinsics.txt", K); np.savetxt(folder / "depthIntrinsics.txt", K)
    rots = [np.eye(3)] + [Rotation.from_rotvec(v).as_matrix() for v in ([np.pi, 0, 0], [0, np.pi, 0], [0, 0, np.pi], [1e-9, 0, 0], [1e-5, 2e-5, -1e-5],
                                                                          np.array([1, 1, 1]) / np.sqrt(3) * np.pi, np.array([1, 2, 3]) / np.sqrt(14) * (np.pi - 1e-4))]
    rots += [np.diag([1, -1, -1.0]), np.diag([-1, -1, 1.0])]
    rng = np.random.default_rng(0)
    for i, Rm in enumerate(rots):
        Image.fromarray(rng.integers(0, 256, (24, 32, 3), np.uint8)).save(folder / f"frame-{i:06d}.color.png")
        Image.fromarray(rng.integers(500, 900, (24, 32)).astype(np.uint16)).save(folder / f"frame-{i:06d}.depth.png")
        T = np.eye(4); T[:3, :3] = Rm; T[:3, 3] = rng.normal(size=3); np.savetxt(folder / f"frame-{i:06d}.pose.txt", T, fmt="%.17g")
    m = ref_py.InitModel(folder, [True] * len(rots), 1, 0, 0.1, 10.0); s = B.Sensor(folder, 0, 0.1, 10.0)
    for k in range(len(rots)):
        a = np.asarray(m.poses[k]); b = B.pose_mat_to_vec6(s.pose(k))
        assert np.abs(a - b).max() <= 1e-15 * max(1.0, np.abs(a).max()) * 4, (k, a, b)
    assert abs(np.linalg.norm(m.poses[1][:3]) - np.pi) < 1e-7 and not np.asarray(m.poses[0][:3]).any()
    m.close(); s.close()
