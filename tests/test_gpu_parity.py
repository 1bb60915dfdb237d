"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): voxel indexing / row structure bit-exact; residuals, Jacobians, SDF / albedo / camera
within 1e-4 relative.  Jacobians are stored in fp32 on the device, so per-entry checks use rtol 1e-4 with an atol tied
to the row's largest partial."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(oracle):
    sc = helpers.small_scene()
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    ctx = helpers.gpu_context(sc, arrays, vsh)
    yield dict(O=oracle, sc=sc, g=g, fr=fr, arrays=arrays, vsh=vsh, thres=thres, ctx=ctx)
    ctx.close()


def test_neighbor_table_bit_exact(setup):
    keys = setup["arrays"]["keys"]
    index = {tuple(k): i for i, k in enumerate(keys.tolist())}
    nb = setup["ctx"].debug_neighbors()
    offs = [(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1),(2,0,0),(0,2,0),(0,0,2),(1,1,0),(1,0,1),(0,1,1),
            (-2,0,0),(0,-2,0),(0,0,-2),(-1,-1,0),(-1,0,-1),(0,-1,-1)]
    exp = np.full_like(nb, -1)
    for j, o in enumerate(offs):
        q = keys + np.array(o, np.int32)
        exp[:, j] = [index.get(tuple(k), -1) for k in q.tolist()]
    assert np.array_equal(nb, exp)


def _assemble(setup, **kw):
    O = setup["O"]
    ocfg = helpers.oracle_cfg(O, setup["thres"], **kw)
    pv = O.ProblemView(setup["g"], setup["fr"], ocfg, setup["sc"]["intr"], setup["sc"]["dist"], setup["sc"]["poses"], setup["vsh"], 0)
    ctx = setup["ctx"]
    ctx.debug_assemble(helpers.gpu_cfg(ocfg), 0)
    return ocfg, pv, ctx


def test_flags_and_row_structure_bit_exact(setup):
    ocfg, pv, ctx = _assemble(setup)
    fl = pv.flags(); gf = ctx.debug_flags()
    assert np.array_equal((gf >> 1) & 1, fl["active"])
    assert np.array_equal((gf >> 2) & 1, fl["ring_ok"])
    assert np.array_equal(((gf >> 3) & 1) ^ 1, fl["fix_sdf"])
    assert np.array_equal(((gf >> 4) & 1) ^ 1, fl["fix_alb"])
    # Eg rows: same (voxel, frame) set
    v, f, w, r, _ = pv.eg(with_jacobian=False)
    gfr, gw, gr, _ = ctx.debug_eg_rows(jac=False)
    exp = set(zip(v.tolist(), f.tolist()))
    got = set((int(i), int(gfr[i, k])) for i, k in zip(*np.nonzero(gfr >= 0)))
    assert got == exp
    # Er / Es / Ea rows
    er, es, ea = ctx.debug_reg_rows()
    v1, _, _, _ = pv.reg(1); v2, _, _, _ = pv.reg(2); v3, d3, w3, _ = pv.reg(3)
    assert set(np.nonzero(er)[0].tolist()) == set(v1.tolist())
    assert set(np.nonzero(es)[0].tolist()) == set(v2.tolist())
    assert set(zip(*[a.tolist() for a in np.nonzero(ea)])) == set(zip(v3.tolist(), d3.tolist()))
    np.testing.assert_allclose(ea[v3, d3], w3, rtol=2e-6)
    assert pv.rows == [len(v), len(v1), len(v2), len(v3)]
    pv.free()


def test_eg_residual_and_jacobian(setup):
    ocfg, pv, ctx = _assemble(setup)
    v, f, w, r, J = pv.eg(with_jacobian=True)
    gfr, gw, gr, gJ = ctx.debug_eg_rows(jac=True)
    slot = np.array([int(np.nonzero(gfr[vi] == fi)[0][0]) for vi, fi in zip(v, f)])
    np.testing.assert_allclose(gw[v, slot], w, rtol=1e-5)
    np.testing.assert_allclose(gr[v, slot], r, rtol=1e-4, atol=1e-9)
    Jg = gJ[v, slot]
    # The residual value path is fp64 on the device; the 29 partials are computed and stored in fp32.  A partial is a c_j-weighted
    # sum over the 4 stencil points with sum_j c_j = 0, so rows whose terms cancel carry an ABSOLUTE error of ~1e-7 x (term size):
    # per entry 1e-4 relative, plus an absolute floor tied to the column's scale over all rows.
    colmax = np.abs(J).max(axis=0, keepdims=True)
    tol = 1e-4 * np.abs(J) + 2e-6 * colmax
    bad = np.abs(Jg - J) > tol
    assert not bad.any(), (np.argwhere(bad)[:5], np.abs(Jg - J)[bad][:5], J[bad][:5])
    # and the classic check on well-conditioned rows: relative to the row's largest partial in each column group
    for lo, hi in [(0, 10), (10, 14), (14, 20), (20, 24), (24, 29)]:
        sc = np.abs(J[:, lo:hi]).max(axis=1, keepdims=True) + 1e-30
        big = (sc[:, 0] > 1e-2 * colmax[0, lo:hi].max())
        err = np.abs(Jg[big, lo:hi] - J[big, lo:hi]) / sc[big]
        assert err.max() < 2e-4, (lo, hi, err.max())
    pv.free()


def test_normal_equations(setup):
    ocfg, pv, ctx = _assemble(setup)
    cost, g, dg, free = pv.normal_eq()
    gg, gd, gcost = ctx.debug_normal_eq()
    assert abs(gcost - cost) <= 1e-5 * cost
    np.testing.assert_allclose(gd, dg, rtol=2e-4, atol=1e-6 * dg.max())
    np.testing.assert_allclose(gg, g, rtol=2e-3, atol=2e-5 * np.abs(g).max())      # entry by entry (small entries are sums with cancellation: fp32 row partials against the oracle's fp64 duals)
    print(f"\n[normal equations] gradient: max-norm error {np.abs(gg - g).max() / np.abs(g).max():.2e} of max |g|; diagonal: {np.abs(gd - dg).max() / np.abs(dg).max():.2e}")
    assert np.abs(gg - g).max() <= 1e-4 * np.abs(g).max()                          # and in the max-norm convention of the field checks
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, g.shape) * free
    y = pv.jtj_apply(x); gy = ctx.debug_jtj_apply(x)
    np.testing.assert_allclose(gy, y, rtol=2e-3, atol=2e-5 * np.abs(y).max())
    pv.free()


def test_optimize_matches_oracle(setup):
    """3 outer iterations, every parameter group free; PCG iteration counts pinned to the oracle's."""
    O = setup["O"]; sc = setup["sc"]
    g2 = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
    g2.clear_outside_shell(setup["thres"])
    a0 = setup["arrays"]
    g2.import_fields(sdf_refined=a0["sdf_refined"], albedo=a0["albedo"])
    ocfg = helpers.oracle_cfg(O, setup["thres"], cg_fixed_iterations=5)
    rc, intr, dist, poses, stats = O.optimize(g2, setup["fr"], ocfg, sc["intr"], sc["dist"], sc["poses"], setup["vsh"])
    assert rc == 0
    ref = g2.export()
    ctx = helpers.gpu_context(sc, a0, setup["vsh"])
    gst = ctx.optimize(helpers.gpu_cfg(ocfg))
    sdf, alb = ctx.get_grid(); gi, gd, gp = ctx.get_camera()
    for so, sg in zip(stats, gst):
        assert list(so.rows) == list(sg.rows)
        assert abs(so.cost_initial - sg.cost_initial) <= 1e-4 * so.cost_initial
        assert abs(so.cost_final - sg.cost_final) <= 1e-4 * so.cost_final
        assert list(so.accepted[:so.n_attempts]) == list(sg.step_accepted[:sg.num_attempts])
    dsdf = np.abs(sdf - ref["sdf_refined"]).max(); dalb = np.abs(alb - ref["albedo"]).max()
    assert dsdf <= 1e-4 * np.abs(ref["sdf_refined"]).max(), dsdf
    assert dalb <= 1e-4 * np.abs(ref["albedo"]).max(), dalb
    np.testing.assert_allclose(gi, intr, rtol=1e-4)
    np.testing.assert_allclose(gp, poses, rtol=1e-4, atol=1e-6)
    # Distortion: k2, k3 are barely observable (r^4, r^6 with r < 0.5 in normalised image coordinates), so the 5 x 5 distortion block is ill-conditioned and the
    # bound cannot be a blanket number.  It is DERIVED: the oracle is run again on input fields perturbed by 1e-7 relative, which measures — component by component —
    # how far the answer moves per unit of relative perturbation of the problem (the amplification); the device reproduces residuals and Jacobian entries to 1e-4
    # (test_eg_residual_and_jacobian), i.e. solves a problem perturbed by at most 1e-4, so its distortion may differ by amplification x 1e-4, and never needs more
    # than the north star's 1e-4 where the amplification is below one.  Printed: the amplification and the perturbation the observed error corresponds to.
    g3 = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); g3.clear_outside_shell(setup["thres"])
    rng = np.random.default_rng(11); eps = 1e-7
    g3.import_fields(sdf_refined=a0["sdf_refined"] * (1.0 + eps * rng.standard_normal(len(a0["sdf_refined"]))), albedo=a0["albedo"] * (1.0 + eps * rng.standard_normal(len(a0["albedo"]))))
    rc3, _, dist3, _, _ = O.optimize(g3, setup["fr"], ocfg, sc["intr"], sc["dist"], sc["poses"], setup["vsh"]); g3.free()
    assert rc3 == 0
    amp = np.abs(np.asarray(dist3) - np.asarray(dist)) / eps; err = np.abs(np.asarray(gd) - np.asarray(dist))
    # capped by the blanket bound of rounds 1-3 (advisor finding of round 5: one flipped discrete decision in the perturbed oracle run would inflate `amp` without limit)
    tol = np.minimum(np.maximum(1e-4 * np.abs(np.asarray(dist)), 1e-4 * amp), 5e-3 * np.abs(np.asarray(dist)) + 5e-4) + 1e-12
    print(f"\n[distortion] oracle {np.asarray(dist)}\n  device error {err}\n  amplification (change per unit relative perturbation of the fields) {amp}\n  bound {tol}\n"
          f"  the device's error corresponds to a relative perturbation of {err / np.maximum(amp, 1e-30)} (bar: 1e-4)")
    assert np.all(err <= tol), (err, tol)
    ctx.close(); g2.free()


def test_rows_whose_spline_window_crosses_the_image_border(oracle):
    """The bicubic window of a projected point near the image border is clamped column by column (Grid2D, cost.h:108-127).  The device loads ONE 16-byte tap row
    from a clamped start column and folds the spline weights of the clamped taps (build.hip, bicubic_taps / fold_weights): residuals, Jacobians and the
    candidate cost of a scene that is larger than every image must still match the oracle."""
    from intrinsic3d_amd import synthetic
    sc = helpers.small_scene(seed=7, fx=300.0, cam_dist=0.17)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    ocfg = helpers.oracle_cfg(oracle, thres, cg_fixed_iterations=5, iterations=1)
    pv = oracle.ProblemView(g, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh, 0)
    ctx = helpers.gpu_context(sc, arrays, vsh)
    try:
        ctx.debug_assemble(helpers.gpu_cfg(ocfg), 0)
        v, f, w, r, J = pv.eg(with_jacobian=True)
        gfr, gw, gr, gJ = ctx.debug_eg_rows(jac=True)
        assert set(zip(v.tolist(), f.tolist())) == set((int(i), int(gfr[i, k])) for i, k in zip(*np.nonzero(gfr >= 0)))
        # how many of these rows sit within two pixels of the border (voxel centre through the keyframe's pose; the iso-point is a fraction of a voxel away)
        P = arrays["keys"][v].astype(np.float64) * float(sc["voxel_size"])
        poses = np.asarray(sc["poses"], np.float64).reshape(-1, 6)
        R = np.stack([synthetic.aa_to_rotmat(p6[:3]) for p6 in poses]); t = poses[:, 3:]
        q = np.einsum("nij,nj->ni", R[f], P) + t[f]
        u = sc["intr"][0] * q[:, 0] / q[:, 2] + sc["intr"][2]; vv = sc["intr"][1] * q[:, 1] / q[:, 2] + sc["intr"][3]
        W, H = sc["width"], sc["height"]
        near = (u < 2.0) | (u > W - 3.0) | (vv < 2.0) | (vv > H - 3.0)
        assert near.sum() >= 20, int(near.sum())
        slot = np.array([int(np.nonzero(gfr[vi] == fi)[0][0]) for vi, fi in zip(v, f)])
        np.testing.assert_allclose(gr[v, slot], r, rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(gr[v, slot][near], r[near], rtol=1e-4, atol=1e-9)
        colmax = np.abs(J).max(axis=0, keepdims=True)
        bad = np.abs(gJ[v, slot] - J) > 1e-4 * np.abs(J) + 2e-6 * colmax
        assert not bad.any(), (np.argwhere(bad)[:5], near[np.argwhere(bad)[:5, 0]])
        pv.free()
        # one iteration through the trust-region loop: the candidate-cost kernel (software-pipelined variant) sees the same border rows
        g2 = oracle.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        g2.clear_outside_shell(thres); g2.import_fields(sdf_refined=arrays["sdf_refined"], albedo=arrays["albedo"])
        rc, intr, dist, poses_o, stats = oracle.optimize(g2, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh)
        assert rc == 0
        gst = ctx.optimize(helpers.gpu_cfg(ocfg))
        for so, sg in zip(stats, gst):
            assert list(so.rows) == list(sg.rows)
            assert abs(so.cost_initial - sg.cost_initial) <= 1e-4 * so.cost_initial
            assert abs(so.cost_final - sg.cost_final) <= 1e-4 * so.cost_final
            assert list(so.accepted[:so.n_attempts]) == list(sg.step_accepted[:sg.num_attempts])
        g2.free()
    finally:
        ctx.close()


def test_estimate_sh_matches_oracle(setup):
    """LightingSVSH::estimate + computeVoxelShCoeffs: fp64 MFMA Gram blocks + host LM vs the oracle's Ceres-equivalent solve."""
    O = setup["O"]; sc = setup["sc"]; thres = setup["thres"]
    for size in (0.05, 0.035, 10.0):
        rc, sh, idx, vsh, has, st = O.estimate_sh(setup["g"], size, 10.0, thres)
        assert rc == 0
        ctx = setup["ctx"]
        gsh, gidx, gst = ctx.estimate_sh(size, 10.0, thres)
        assert gst.subvolumes == st.subvolumes and gst.data_rows == st.data_rows and gst.reg_rows == st.reg_rows
        assert gst.lm_iterations == st.lm_iterations
        order = {tuple(k): i for i, k in enumerate(gidx.tolist())}
        perm = np.array([order[tuple(k)] for k in idx.tolist()])
        np.testing.assert_allclose(gsh[perm], sh, rtol=1e-4, atol=1e-7)
        gv = ctx.get_voxel_sh()
        m = has.astype(bool)
        np.testing.assert_allclose(gv[m], vsh[m], rtol=1e-4, atol=1e-6)
    ctx.set_voxel_sh(setup["vsh"])     # restore the state the other tests expect


def test_estimate_sh_chunk_slabs_do_not_change_the_sums(setup, monkeypatch):
    """The Gram blocks of the SH data term are accumulated chunk by chunk of a subvolume's voxel run (one wave per chunk, added in chunk order).  The chunk dimension runs
    in slabs that bound the scratch and the grid (advisor finding of round 5: one large subvolume beside many small ones made the scratch S x longest-run, and a global
    volume of > 134 M voxels passed the gridDim.y limit, unchecked): forced here to ONE chunk per launch on the single-volume estimate of the scene (several chunks of
    2048 voxels), the coefficients must be bit for bit those of the default slab."""
    ctx = setup["ctx"]; thres = setup["thres"]
    a, ia, sa = ctx.estimate_sh(10.0, 10.0, thres)
    assert a.shape[0] == 1 and sa.data_rows > 3 * 2048              # one volume, several chunks
    monkeypatch.setenv("I3D_SH_SLAB", "1")
    b, ib, sb = ctx.estimate_sh(10.0, 10.0, thres)
    assert np.array_equal(a, b) and sa.cost_final == sb.cost_final
    monkeypatch.delenv("I3D_SH_SLAB")
    ctx.set_voxel_sh(setup["vsh"])


def test_gpu_matches_committed_golden(oracle):
    """HIP path vs tests/golden/optimize_small.json: the oracle's outputs on a seeded scene, committed with their generating script
    (tests/golden/make_golden.py) — regression vectors of the restatement, not reference outputs (the reference cannot be built here)."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optimize_small.json")))
    sc = helpers.small_scene(seed=11, radius_vox=12, K=4, width=128, height=96)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc, sh_size=0.04)
    ctx = helpers.gpu_context(sc, arrays, vsh)
    gsh, gidx, gst = ctx.estimate_sh(0.04, 10.0, thres)          # SH from the device path as well
    ocfg = helpers.oracle_cfg(oracle, thres, iterations=2, cg_fixed_iterations=4)
    stats = ctx.optimize(helpers.gpu_cfg(ocfg))
    sdf, alb = ctx.get_grid(); intr, dist, poses = ctx.get_camera()
    assert len(arrays["keys"]) == gold["num_voxels"]
    assert [list(s.rows) for s in stats] == gold["rows"]
    np.testing.assert_allclose([[s.cost_initial, s.cost_final] for s in stats], gold["cost"], rtol=1e-4)
    np.testing.assert_allclose(sdf[:16], gold["sdf_refined_head"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(alb[:16], gold["albedo_head"], rtol=1e-4)
    np.testing.assert_allclose(np.sum(sdf), gold["sdf_sum"], rtol=1e-4)
    np.testing.assert_allclose(np.sum(alb), gold["albedo_sum"], rtol=1e-5)
    np.testing.assert_allclose(intr, gold["intr"], rtol=1e-4)
    np.testing.assert_allclose(poses, gold["poses"], rtol=1e-4, atol=1e-6)
    ctx.close(); g.free(); fr.free()


def _run_sharded(setup, W, cfg):
    """W simulated ranks (host threads, one GPU) through one optimize call: (contexts, per-rank stats)"""
    import threading
    from intrinsic3d_amd import binding
    sc = setup["sc"]; a0 = setup["arrays"]; vsh = setup["vsh"]
    L = binding.load()
    shared = L.i3d_comm_sim_create(W)
    ctxs = [helpers.gpu_context(sc, a0, vsh) for _ in range(W)]
    for r, c in enumerate(ctxs):
        c.comm_init_sim(shared, r)
    out = [None] * W; err = [None] * W

    def run(r):
        try:
            out[r] = ctxs[r].optimize(cfg)
        except Exception as e:      # surface failures instead of deadlocking the other ranks silently
            err[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    [t.start() for t in th]; [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "sharded run hung"
    assert all(e is None for e in err), err
    return L, shared, ctxs, out


def test_sharded_run_survives_a_tile_halo_that_does_not_fit(setup, monkeypatch, capfd):
    """A sharded run has no untiled operator pass to fall back to.  When the halo of a 512-entry tile does not fit on ANY rank (forced here: the plan is
    told that such a tile has 48 halo slots), all ranks agree (max all-reduce of the overflow flag) and plan again with 1024-entry tiles — then the
    result must be the single-rank one.  Without the knob the same scene runs on 512-entry tiles."""
    O = setup["O"]
    cfg = helpers.gpu_cfg(helpers.oracle_cfg(O, setup["thres"], iterations=2, cg_fixed_iterations=12))
    ref = helpers.gpu_context(setup["sc"], setup["arrays"], setup["vsh"])
    rst = ref.optimize(cfg); rsdf, ralb = ref.get_grid(); ref.close()
    monkeypatch.setenv("I3D_EGT_HMAX_LIMIT", "48")
    L, shared, ctxs, out = _run_sharded(setup, 2, cfg)
    assert "planning again with 1024-entry tiles" in capfd.readouterr().err
    for r, c in enumerate(ctxs):
        sdf, alb = c.get_grid()
        for s1, s2 in zip(rst, out[r]):
            assert list(s1.rows) == list(s2.rows) and list(s1.step_accepted[:s1.num_attempts]) == list(s2.step_accepted[:s2.num_attempts])
        assert np.abs(sdf - rsdf).max() <= 1e-4 * np.abs(rsdf).max() and np.abs(alb - ralb).max() <= 1e-4 * np.abs(ralb).max()
        c.close()
    L.i3d_comm_sim_destroy(shared)


@pytest.mark.parametrize("ladder", ["6", "1"])
def test_single_rank_falls_back_to_the_other_tile_geometry(setup, monkeypatch, capfd, ladder):
    """Single rank.  With the damping ladder (the default) the plan has 512-entry tiles with 1536 halo slots — the geometry of the multi-system operator pass —, the
    serial loop (I3D_LADDER=1) plans 1024-entry tiles with 2048 halo slots.  When a tile's halo does not fit the first geometry (forced here: the plan is told it has
    48 slots) the run plans again with the other one before it would give up on the tiled pass (the ladder then steps back to the serial loop: its kernel exists
    for 512-entry tiles); with both geometries refused it takes the untiled pass.  All three must give the same result."""
    O = setup["O"]
    cfg = helpers.gpu_cfg(helpers.oracle_cfg(O, setup["thres"], iterations=2, cg_fixed_iterations=12))
    monkeypatch.setenv("I3D_LADDER", ladder)
    first, other = (("I3D_EGT_HMAX_LIMIT", "I3D_EGT_HMAX_LIMIT_1024") if ladder != "1" else ("I3D_EGT_HMAX_LIMIT_1024", "I3D_EGT_HMAX_LIMIT"))
    other_tiles = "1024" if ladder != "1" else "512"

    def run():
        c = helpers.gpu_context(setup["sc"], setup["arrays"], setup["vsh"])
        st = c.optimize(cfg); sdf, alb = c.get_grid(); c.close()
        return st, sdf, alb
    rst, rsdf, ralb = run()
    assert "does not fit" not in capfd.readouterr().err
    monkeypatch.setenv(first, "48")
    st1, sdf1, alb1 = run()
    err = capfd.readouterr().err
    assert f"planning again with {other_tiles}-entry tiles" in err and "untiled" not in err
    monkeypatch.setenv(other, "48")
    st2, sdf2, alb2 = run()
    assert "using the untiled pass" in capfd.readouterr().err
    for st, sdf, alb in ((st1, sdf1, alb1), (st2, sdf2, alb2)):
        for s1, s2 in zip(rst, st):
            assert list(s1.rows) == list(s2.rows) and list(s1.step_accepted[:s1.num_attempts]) == list(s2.step_accepted[:s2.num_attempts])
        assert np.abs(sdf - rsdf).max() <= 1e-4 * np.abs(rsdf).max() and np.abs(alb - ralb).max() <= 1e-4 * np.abs(ralb).max()


def test_sharded_ranks_match_single_rank(setup):
    """The SPMD path (tile-aligned owned ranges, compute lists with ghost entries, ghost tiles of the operator pass, rim exchange of the
    operator input, reduced PCG scalars / camera block) with W ranks simulated by W host threads on ONE GPU (i3d_comm_init_sim) must
    reproduce the single-rank result.  (The scene's work list spans a dozen ownership tiles: W = 8 leaves some ranks without rows.)"""
    import threading
    from intrinsic3d_amd import binding
    O = setup["O"]; sc = setup["sc"]; a0 = setup["arrays"]; vsh = setup["vsh"]
    ocfg = helpers.oracle_cfg(O, setup["thres"], iterations=2, cg_fixed_iterations=12)      # 12 > residual_reset_period: exercises r = b - A x too
    cfg = helpers.gpu_cfg(ocfg)
    ref = helpers.gpu_context(sc, a0, vsh)
    rst = ref.optimize(cfg); rsdf, ralb = ref.get_grid(); ri, rd, rp = ref.get_camera(); ref.close()
    L = binding.load()
    for W in (2, 3, 8):
        shared = L.i3d_comm_sim_create(W)
        ctxs = [helpers.gpu_context(sc, a0, vsh) for _ in range(W)]
        for r, c in enumerate(ctxs):
            c.comm_init_sim(shared, r)
        out = [None] * W; err = [None] * W

        def run(r):
            try:
                out[r] = ctxs[r].optimize(cfg)
            except Exception as e:      # surface failures instead of deadlocking the other ranks silently
                err[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
        [t.start() for t in th]; [t.join(timeout=120) for t in th]
        assert not any(t.is_alive() for t in th), "sharded run hung"
        assert all(e is None for e in err), err
        for r, c in enumerate(ctxs):
            sdf, alb = c.get_grid(); gi, gd, gp = c.get_camera()
            for k, (s1, s2) in enumerate(zip(rst, out[r])):
                what = (W, r, k, list(s1.step_accepted[:s1.num_attempts]), list(s2.step_accepted[:s2.num_attempts]), list(s1.pcg_iterations[:s1.num_attempts]), list(s2.pcg_iterations[:s2.num_attempts]),
                        s1.cost_initial, s2.cost_initial, s1.cost_final, s2.cost_final, s1.final_radius, s2.final_radius)
                assert list(s1.step_accepted[:s1.num_attempts]) == list(s2.step_accepted[:s2.num_attempts]), what
                assert list(s1.rows) == list(s2.rows) and s1.valid_voxels == s2.valid_voxels and s1.free_parameters == s2.free_parameters
                # iteration 0 starts from identical state: only the fp64 summation order differs; later ones inherit the fp32 PCG round-off
                # (fp32 atomics inside the operator make the PCG round-off run-to-run variable; the bar is the north-star 1e-4)
                assert abs(s1.cost_initial - s2.cost_initial) <= (1e-12 if k == 0 else 1e-4) * s1.cost_initial
                assert abs(s1.cost_final - s2.cost_final) <= 1e-4 * s1.cost_final, what
            assert np.abs(sdf - rsdf).max() <= 1e-4 * np.abs(rsdf).max()      # fp32 PCG round-off, different partial-sum order
            assert np.abs(alb - ralb).max() <= 1e-4 * np.abs(ralb).max()
            np.testing.assert_allclose(gi, ri, rtol=1e-4); np.testing.assert_allclose(gp, rp, rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(gd, rd, rtol=5e-3, atol=5e-4)   # k2,k3 barely observable: ill-conditioned block (see test_optimize_matches_oracle)
            c.close()
        L.i3d_comm_sim_destroy(shared)


def test_config_c1_dense_albedo_only(oracle):
    """BASELINE.json configs[0]: dense 64^3 SDF (sphere r = 24 voxels @ 4 mm), ONE keyframe 640x480, ONE global SH volume,
    albedo-only (sdf and camera blocks constant).  The whole level: sparsify -> SH estimate -> optimize, device vs oracle."""
    from intrinsic3d_amd import binding, synthetic
    O = oracle
    sc = synthetic.make_scene(radius_vox=24, voxel_size=0.004, K=1, width=640, height=480, dense_res=64, seed=11)
    assert sc["keys"].shape[0] == 64 ** 3
    thres = 2.0 * float(sc["voxel_size"])
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); fr = O.Frames(sc["frames"], 1)
    g.clear_outside_shell(thres)
    rc, osh, _, vsh, _, ost = O.estimate_sh(g, 0.0, 10.0, thres)                 # subvolume_size_sh 0 -> one global volume
    assert rc == 0 and osh.shape[0] == 1
    ocfg = helpers.oracle_cfg(O, thres, iterations=3, fix_poses=1, fix_intrinsics=1, fix_distortion=1, fix_sdf=1, num_observations=5)
    rc, ointr, odist, oposes, ostats = O.optimize(g, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh)
    assert rc == 0
    ref = g.export()
    with binding.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        ctx.set_frames(sc["frames"], 1); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        n = ctx.clear_outside_thin_shell(thres)
        assert n == len(g)
        gsh, _, gst = ctx.estimate_sh(0.0, 10.0, thres)
        np.testing.assert_allclose(gsh, osh, rtol=1e-4, atol=1e-6)
        gstats = ctx.optimize(helpers.gpu_cfg(ocfg))
        out = ctx.export_grid(); gi, gd, gp = ctx.get_camera()
    assert np.array_equal(out["keys"], ref["keys"])
    for so, sg in zip(ostats, gstats):
        assert list(so.rows) == list(sg.rows) and so.rows[0] > 1000
        assert abs(so.cost_final - sg.cost_final) <= 1e-4 * so.cost_final
    assert np.array_equal(out["sdf_refined"], ref["sdf_refined"])               # sdf blocks are constant
    assert np.abs(ref["albedo"] - 0.6).max() > 1e-3                              # the albedo did move
    assert np.abs(out["albedo"] - ref["albedo"]).max() <= 1e-4 * np.abs(ref["albedo"]).max()
    assert np.array_equal(gp, np.asarray(sc["poses"], np.float64)) and np.array_equal(gi, np.asarray(sc["intr"], np.float64))
    g.free(); fr.free()


def test_multi_tile_problem_matches_oracle(oracle):
    """~0.6 M work-list entries: more than 256 x 1024, so the persistent workgroups of the operator pass walk SEVERAL tiles each (the
    small scenes above never do), vectors span many workgroups, and the reductions go through thousands of partials"""
    from intrinsic3d_amd import synthetic
    O = oracle
    sc = synthetic.make_scene(radius_vox=112, voxel_size=0.004, K=6, width=320, height=240, levels=1, seed=21, pose_noise=(0.0005, 0.001))
    thres = 2.0 * float(sc["voxel_size"])
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); fr = O.Frames(sc["frames"], 1)
    g.clear_outside_shell(thres)
    rc, _, _, vsh, _, _ = O.estimate_sh(g, 0.2, 10.0, thres)
    assert rc == 0
    arrays = g.export()
    ocfg = helpers.oracle_cfg(O, thres, iterations=1, cg_fixed_iterations=4, lm_steps=3)
    rc, ointr, odist, oposes, ostats = O.optimize(g, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh)
    assert rc == 0
    ref = g.export()
    ctx = helpers.gpu_context(sc, arrays, vsh)
    gstats = ctx.optimize(helpers.gpu_cfg(ocfg))
    sdf, alb = ctx.get_grid(); gi, gd, gp = ctx.get_camera()
    sizes = ctx.problem_sizes(); ctx.close()
    assert sizes["active"] > 256 * 1024, sizes
    so, sg = ostats[0], gstats[0]
    assert list(so.rows) == list(sg.rows)
    assert abs(so.cost_initial - sg.cost_initial) <= 1e-5 * so.cost_initial and abs(so.cost_final - sg.cost_final) <= 1e-4 * so.cost_final
    assert list(so.accepted[:so.n_attempts]) == list(sg.step_accepted[:sg.num_attempts])
    assert np.abs(sdf - ref["sdf_refined"]).max() <= 1e-4 * np.abs(ref["sdf_refined"]).max()
    assert np.abs(alb - ref["albedo"]).max() <= 1e-4 * np.abs(ref["albedo"]).max()
    np.testing.assert_allclose(gi, ointr, rtol=1e-4); np.testing.assert_allclose(gp, oposes, rtol=1e-4, atol=1e-6)
    # the same problem under the SPMD path (2 and 8 simulated ranks): owned ranges of ~0.3 M / ~75 k entries, ghost rows, ghost tiles, rim exchange
    import threading
    from intrinsic3d_amd import binding
    L = binding.load()
    for W in (2, 8):
        shared = L.i3d_comm_sim_create(W)
        ctxs = [helpers.gpu_context(sc, arrays, vsh) for _ in range(W)]
        for r, c in enumerate(ctxs):
            c.comm_init_sim(shared, r)
        err = [None] * W

        def run(r):
            try:
                ctxs[r].optimize(helpers.gpu_cfg(ocfg))
            except Exception as e:
                err[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
        [t.start() for t in th]; [t.join(timeout=300) for t in th]
        assert not any(t.is_alive() for t in th) and all(e is None for e in err), err
        A = sizes["active"]
        for r, c in enumerate(ctxs):
            s2, a2 = c.get_grid()
            assert np.abs(s2 - sdf).max() <= 1e-4 * np.abs(sdf).max() and np.abs(a2 - alb).max() <= 1e-4
            cs = c.comm_stats()
            # what a rank exchanges per PCG pass is its RIM, never a vector: 8 bytes per rim entry (a 5-voxel-thick shell cut into 8 patches of
            # ~125 x 125 voxels here: ~16 % of what a rank owns; ~5 % for the 1-mm bench shell at 8 ranks)
            assert 0 < cs["halo_send"] < 0.25 * A / W and 0 < cs["halo_recv"] < 0.25 * A / W, cs
            assert cs["compute_list"] < 1.25 * A / W + 2048, cs                     # owned + ghost entries
            # a rim message carries 8 bytes per entry and SYSTEM: one system per message in the serial loop, the live systems of a ladder batch otherwise
            assert cs["halo_calls"] > 0 and cs["halo_bytes_sent"] % (8 * cs["halo_send"]) == 0 and cs["halo_calls"] <= cs["halo_bytes_sent"] // (8 * cs["halo_send"]) <= 6 * cs["halo_calls"], cs
            if r == 0:
                print(f"W={W}: rank 0 owns ~{A // W} entries, compute list {cs['compute_list']}, rim sent/received per pass {cs['halo_send']}/{cs['halo_recv']} entries "
                      f"({8 * cs['halo_send']} B), ghost tiles {cs['ghost_tiles']}, all-reduce bytes per call {cs['reduce_bytes'] // max(cs['reduce_calls'], 1)}")
            c.close()
        L.i3d_comm_sim_destroy(shared)
    g.free(); fr.free()
