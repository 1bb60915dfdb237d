"""-m gpu: edge cases of the path, device vs oracle through the C ABI — negative voxel coordinates (hash sign-extension, truncating
round), constant albedo (lambda_a < 0), fixed camera blocks, "all observations" (num_observations = 0), capacity errors, grids without
any active voxel, a keyframe that sees nothing."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _run_both(O, sc, thres, **cfgkw):
    from intrinsic3d_amd import binding
    g = O.Grid.from_voxels(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"]); fr = O.Frames(sc["frames"], sc["levels"])
    g.clear_outside_shell(thres)
    rc, _, _, vsh, _, _ = O.estimate_sh(g, 0.05, 10.0, thres)
    assert rc == 0
    ocfg = helpers.oracle_cfg(O, thres, **dict(dict(iterations=2, cg_fixed_iterations=6), **cfgkw))
    rc, ointr, odist, oposes, ostats = O.optimize(g, fr, ocfg, sc["intr"], sc["dist"], sc["poses"], vsh)
    ref = g.export()
    with binding.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        ctx.set_frames(sc["frames"], sc["levels"]); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        assert ctx.clear_outside_thin_shell(thres) == len(g)
        ctx.estimate_sh(0.05, 10.0, thres)
        gstats = ctx.optimize(helpers.gpu_cfg(ocfg))
        out = ctx.export_grid(); cam = ctx.get_camera()
    g.free(); fr.free()
    return rc, ref, (ointr, odist, oposes), ostats, out, cam, gstats


def _check(ref, ostats, out, gstats, tol=1e-4):
    assert np.array_equal(out["keys"], ref["keys"])
    for so, sg in zip(ostats, gstats):
        assert list(so.rows) == list(sg.rows)
        assert abs(so.cost_final - sg.cost_final) <= tol * max(so.cost_final, 1e-30)
    assert np.abs(out["sdf_refined"] - ref["sdf_refined"]).max() <= tol * np.abs(ref["sdf_refined"]).max()
    assert np.abs(out["albedo"] - ref["albedo"]).max() <= tol * np.abs(ref["albedo"]).max()


@pytest.mark.parametrize("where", ["negative_octant", "around_the_origin"])
def test_negative_coordinates(oracle, where):
    """the same scene translated into the negative octant: keys < 0 exercise the sign-extending hash (mat.h:117-124) in the visit order
    and the float voxel->world products; the surface must be found by the cameras all the same.  Around the origin (round 6): keys of every sign in every coordinate, and
    lighting subvolumes on both sides of every coordinate plane — their index is a FLOOR (subvolumes.cpp:281-295), unlike the truncating voxel index of row a2."""
    sc = dict(helpers.small_scene(seed=7, radius_vox=9, K=4, width=96, height=72))
    shift = np.array([-40, -33, -51], np.int32) if where == "negative_octant" else -np.round(np.asarray(sc["center"], np.float64) / float(sc["voxel_size"])).astype(np.int32)
    sc["keys"] = sc["keys"] + shift[None, :]
    t = shift.astype(np.float64) * float(sc["voxel_size"])
    poses = np.array(sc["poses"], np.float64)
    for f in range(len(poses)):                      # world shifted by t: x_cam = R (x_w' - t) + tr  ->  tr' = tr - R t
        R, _ = oracle.pose_to_mat(poses[f]); poses[f, 3:] = poses[f, 3:] - R.astype(np.float64).reshape(3, 3) @ t
    sc["poses"] = poses
    thres = 2.0 * float(sc["voxel_size"])
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres)
    assert rc == 0 and ostats[0].rows[0] > 500 and (ref["keys"] < 0).all(axis=1).any()
    if where == "around_the_origin":
        assert all((ref["keys"][:, a] < 0).any() and (ref["keys"][:, a] > 0).any() for a in range(3))
    _check(ref, ostats, out, gstats)


@pytest.mark.parametrize("kw", [dict(lambda_a=-1.0), dict(fix_poses=1, fix_intrinsics=1, fix_distortion=1), dict(num_observations=0),
                                dict(fix_poses=1, fix_distortion=1, num_observations=2), dict(num_observations=8, K=12)])
def test_parameter_group_switches(oracle, kw):
    kw = dict(kw); K = kw.pop("K", 4)                 # (num_observations = 8 of 12 keyframes: the largest number of row slots per voxel the row layout holds)
    sc = helpers.small_scene(seed=8, radius_vox=9, K=K, width=96, height=72)
    thres = 2.0 * float(sc["voxel_size"])
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres, **kw)
    assert rc == 0
    _check(ref, ostats, out, gstats)
    if kw.get("lambda_a", 0) < 0:
        assert np.all(out["albedo"] == 0.6) and np.all(ref["albedo"] == 0.6)       # constant albedo: every albedo block fixed, no Ea rows
        assert gstats[0].rows[3] == 0
    if kw.get("fix_poses"):
        assert np.array_equal(cam[2], np.asarray(sc["poses"], np.float64))
    if kw.get("num_observations", 5) == 2:
        assert gstats[0].rows[0] <= 2 * ostats[0].valid_voxels                     # at most the 2 best observations per voxel
    if kw.get("num_observations", 5) == 8:
        assert 4.2 * ostats[0].valid_voxels < gstats[0].rows[0] <= 8 * ostats[0].valid_voxels    # (4.34 rows per voxel on this scene: a voxel is seen by 4-6 of the 12 keyframes; many voxels hold 6-8 rows)


def test_capacity_and_state_errors(oracle):
    from intrinsic3d_amd import binding
    sc = helpers.small_scene(seed=9, radius_vox=6, K=10, width=48, height=36)
    with binding.Context(0) as ctx:
        with pytest.raises(binding.I3DError):
            ctx.optimize(binding.default_config(iterations=1))                     # nothing set: state error, not a crash
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        ctx.set_frames(sc["frames"], 1); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        ctx.set_voxel_sh(np.tile(np.asarray(sc["scene"].sh), (ctx.N, 1)))
        with pytest.raises(binding.I3DError):                                      # 10 keyframes, "all observations": > 8 rows per voxel
            ctx.optimize(binding.default_config(iterations=1, num_observations=0, thres_shell=0.01))
        with pytest.raises(binding.I3DError):
            ctx.optimize(binding.default_config(iterations=0, thres_shell=0.01))   # optimizer.cpp:113-114
        with pytest.raises(binding.I3DError):
            ctx.optimize(binding.default_config(iterations=1, thres_shell=0.01, rgbd_level=3))
        with pytest.raises(binding.I3DError):
            ctx.estimate_sh(0.05, 10.0, 0.0)                                       # LightingSVSH::estimate: thres_shell <= 0


def test_no_active_voxel_and_blind_keyframe(oracle):
    """(1) a shell threshold of ~0 leaves no in-shell voxel: zero rows, nothing moves, no error (the reference logs and continues);
    (2) a keyframe that looks away contributes no observation and its pose stays put"""
    from intrinsic3d_amd import binding
    sc = dict(helpers.small_scene(seed=10, radius_vox=8, K=3, width=96, height=72))
    with binding.Context(0) as ctx:
        ctx.set_grid_from_tsdf_records(sc["voxel_size"], sc["keys"], sc["sdf"], sc["weight"], sc["color"])
        ctx.set_frames(sc["frames"], 1); ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        ctx.set_voxel_sh(np.tile(np.asarray(sc["scene"].sh), (ctx.N, 1)))
        before = ctx.export_grid()
        st = ctx.optimize(binding.default_config(iterations=2, thres_shell=1e-12))
        after = ctx.export_grid()
        assert list(st[0].rows) == [0, 0, 0, 0] and np.array_equal(before["sdf_refined"], after["sdf_refined"]) and np.array_equal(before["albedo"], after["albedo"])
    poses = np.array(sc["poses"], np.float64)
    R, _ = oracle.pose_to_mat(poses[1])
    poses[1, :3] = 0.0; poses[1, 3:] = [0.0, 0.0, -5.0]            # camera 1: identity rotation, everything 5 m behind it
    sc["poses"] = poses
    thres = 2.0 * float(sc["voxel_size"])
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres)
    assert rc == 0
    _check(ref, ostats, out, gstats)
    assert np.array_equal(cam[2][1], poses[1]) and np.array_equal(ocam[2][1], poses[1])      # no row touches pose 1: the block never enters the problem


def test_one_call_host_entry_point_matches_resident_path(oracle):
    """i3d_optimize_host (the shim of INTEGRATION.md: host arrays in, unknowns written back) == the resident context path"""
    from intrinsic3d_amd import binding
    sc = helpers.small_scene(seed=12, radius_vox=9, K=4, width=96, height=72)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    cfg = helpers.gpu_cfg(helpers.oracle_cfg(oracle, thres, iterations=2, cg_fixed_iterations=6))
    ctx = helpers.gpu_context(sc, arrays, vsh)
    st1 = ctx.optimize(cfg); sdf1, alb1 = ctx.get_grid(); i1, d1, p1 = ctx.get_camera(); ctx.close()
    sdf2, alb2, i2, d2, p2, st2 = binding.optimize_host(cfg, sc["voxel_size"], arrays["keys"], arrays["sdf"], arrays["sdf_refined"], arrays["albedo"], arrays["weight"],
                                                        arrays["color"], sc["frames"], sc["levels"], sc["intr"], sc["dist"], sc["poses"], vsh)
    assert [list(s.rows) for s in st1] == [list(s.rows) for s in st2]
    smax = np.abs(sdf1).max()
    # two device runs of the same problem: they differ by the summation order of the fp32 atomics (north-star tolerance)
    assert np.abs(sdf1 - sdf2).max() <= 1e-4 * smax and np.abs(alb1 - alb2).max() <= 1e-4
    np.testing.assert_allclose(i2, i1, rtol=1e-4); np.testing.assert_allclose(p2, p1, rtol=1e-4, atol=1e-6)
    assert np.abs(sdf2 - arrays["sdf_refined"]).max() > 0                       # the unknowns did move and were written back
    g.free(); fr.free()


def test_many_keyframes(oracle):
    """350 keyframes: the per-keyframe constants no longer fit the build kernel's LDS staging (global path), the operator pass runs with
    fewer accumulator replicas, and the camera tail of every solver vector is 2109 entries long"""
    sc = helpers.small_scene(seed=13, radius_vox=7, K=350, width=40, height=30)
    thres = 2.0 * float(sc["voxel_size"])
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres, fix_intrinsics=1, fix_distortion=1)
    assert rc == 0 and ostats[0].rows[0] > 1000
    _check(ref, ostats, out, gstats)
    np.testing.assert_allclose(cam[2], ocam[2], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("transport", ["p2p", "rccl", "default"])
def test_real_rccl_collectives_with_one_rank(oracle, monkeypatch, transport):
    """the sharded code path driven through a REAL 1-rank RCCL communicator (I3D_FORCE_COLLECTIVES=1) — same answer as the plain path —
    with the mailbox transport (bootstrap over RCCL, start-up self-test incl. the in-kernel multi-workgroup exchanges; the pass is the three launches of the
    single-rank pass, both exchanges inside k_pcg_dir3 / k_pcg_step3; I3D_TRANSPORT=p2p) and with RCCL, which is also what an unset I3D_TRANSPORT selects
    (I3D_TRANSPORT=rccl: grouped send / receive for the rim, ncclAllReduce for the blocks, the six-launch pass with separate reduction launches)"""
    from intrinsic3d_amd import binding
    if transport == "default":
        monkeypatch.delenv("I3D_TRANSPORT", raising=False); transport = "rccl"      # the mailboxes are opt-in until a multi-GPU run has passed (comm.cpp bootstrap_p2p)
    else:
        monkeypatch.setenv("I3D_TRANSPORT", transport)
    sc = helpers.small_scene(seed=14, radius_vox=9, K=4, width=96, height=72)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    cfg = helpers.gpu_cfg(helpers.oracle_cfg(oracle, thres, iterations=2, cg_fixed_iterations=12))
    ref = helpers.gpu_context(sc, arrays, vsh); st0 = ref.optimize(cfg); s0, a0 = ref.get_grid(); cam0 = ref.get_camera(); ref.close()
    monkeypatch.setenv("I3D_FORCE_COLLECTIVES", "1")
    ctx = helpers.gpu_context(sc, arrays, vsh)
    ctx.comm_init(0, 1, binding.Context.comm_unique_id())
    assert ctx.comm_transport().startswith("p2p-mailbox" if transport == "p2p" else "rccl")
    st1 = ctx.optimize(cfg); s1, a1 = ctx.get_grid(); cam1 = ctx.get_camera(); ctx.close()
    assert [list(s.rows) for s in st0] == [list(s.rows) for s in st1]
    assert np.abs(s1 - s0).max() <= 1e-4 * np.abs(s0).max() and np.abs(a1 - a0).max() <= 1e-4
    np.testing.assert_allclose(cam1[2], cam0[2], rtol=1e-4, atol=1e-6)
    g.free(); fr.free()


def test_sharded_ranks_over_peer_to_peer_mailboxes(oracle, monkeypatch):
    """the per-pass exchanges of the sharded PCG through the mailbox kernels (host/p2p.cpp: peer stores + epoch flags, no host in the loop) with
    2 and 3 ranks simulated by host threads on ONE GPU (I3D_SIM_P2P=1): same answer as one rank, and no wait timed out"""
    import threading
    from intrinsic3d_amd import binding
    sc = helpers.small_scene(seed=17, radius_vox=16, K=4, width=96, height=72)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    cfg = helpers.gpu_cfg(helpers.oracle_cfg(oracle, thres, iterations=2, cg_fixed_iterations=12))
    ref = helpers.gpu_context(sc, arrays, vsh); st0 = ref.optimize(cfg); s0, a0 = ref.get_grid(); cam0 = ref.get_camera(); ref.close()
    monkeypatch.setenv("I3D_SIM_P2P", "1")
    L = binding.load()
    for W in (2, 3):
        shared = L.i3d_comm_sim_create(W)
        ctxs = [helpers.gpu_context(sc, arrays, vsh) for _ in range(W)]
        for r, c in enumerate(ctxs):
            c.comm_init_sim(shared, r)
        err = [None] * W; out = [None] * W

        def run(r):
            try:
                out[r] = ctxs[r].optimize(cfg)
            except Exception as e:
                err[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
        [t.start() for t in th]; [t.join(timeout=120) for t in th]
        assert not any(t.is_alive() for t in th), "a rank hung"
        assert all(e is None for e in err), err
        for r, c in enumerate(ctxs):
            s1, a1 = c.get_grid(); cam1 = c.get_camera()
            assert [list(x.rows) for x in out[r]] == [list(x.rows) for x in st0]
            assert np.abs(s1 - s0).max() <= 1e-4 * np.abs(s0).max() and np.abs(a1 - a0).max() <= 1e-4
            np.testing.assert_allclose(cam1[2], cam0[2], rtol=1e-4, atol=1e-6)
            cs = c.comm_stats(); assert cs["halo_calls"] > 0 and cs["halo_send"] > 0
            c.close()
        L.i3d_comm_sim_destroy(shared)
    g.free(); fr.free()


def test_untiled_operator_fallback(oracle, monkeypatch):
    """the PCG operator without the LDS tile plan (what a grid whose tile halos do not fit falls back to; I3D_NO_TILE=1 forces it):
    k_eg_jtjp + k_gather give the same answer as the tiled pass and the oracle"""
    sc = helpers.small_scene(seed=16, radius_vox=9, K=4, width=96, height=72)
    thres = 2.0 * float(sc["voxel_size"])
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres)
    monkeypatch.setenv("I3D_NO_TILE", "1")
    rc2, ref2, _, ostats2, out2, cam2, gstats2 = _run_both(oracle, sc, thres)
    assert rc == 0 and rc2 == 0
    _check(ref2, ostats2, out2, gstats2)
    assert np.abs(out2["sdf_refined"] - out["sdf_refined"]).max() <= 1e-4 * np.abs(out["sdf_refined"]).max() and np.abs(out2["albedo"] - out["albedo"]).max() <= 1e-4


def test_carry_trust_radius_extension(oracle):
    """opt-in extension (what nls_solver.cpp:322-323 intends): the radius survives from one outer iteration to the next, so later iterations
    need fewer LM attempts; device == oracle with the same switch, and the default still restarts at 1e4"""
    sc = helpers.small_scene(seed=15, radius_vox=9, K=4, width=96, height=72)
    thres = 2.0 * float(sc["voxel_size"])
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres, carry_trust_radius=1)
    assert rc == 0
    _check(ref, ostats, out, gstats)
    assert [s.num_attempts for s in gstats] == [s.n_attempts for s in ostats]
    rc0, ref0, _, ostats0, out0, _, gstats0 = _run_both(oracle, sc, thres)
    assert gstats[1].num_attempts <= gstats0[1].num_attempts and gstats[0].num_attempts == gstats0[0].num_attempts


@pytest.mark.parametrize("radius_vox, band_vox", [(3, 1.6), (2, 2.0)])
def test_grid_smaller_than_one_tile(oracle, radius_vox, band_vox):
    """364 / 251 stored voxels: ONE tile of the operator pass in either geometry, where the plan buffers must hold a 1024-entry tile's 2048 halo slots although the
    grid has fewer voxels than a 512-entry tile (the sizing the round-3 advisor flagged: max over both geometries, solver.cpp alloc_rows)."""
    sc = helpers.small_scene(seed=31, radius_vox=radius_vox, K=4, width=64, height=48, band_vox=band_vox, bump_amp_vox=0.2)
    assert len(sc["keys"]) <= 512
    thres = 2.0 * float(sc["voxel_size"])
    # ONE outer iteration: on a few hundred voxels seen by four small images the first step already moves the SDF by a quarter of its range, and a second
    # iteration (three rejected attempts, then a shrunken radius) amplifies the fp32 round-off of the first to 6e-4 — in EVERY variant of the operator pass
    # (tiled 1024 / 512, untiled, six-launch, deterministic: measured), i.e. the problem, not the plan.  After one iteration: 3e-6.
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres, iterations=1)
    assert rc == 0 and ostats[0].rows[0] > 100
    _check(ref, ostats, out, gstats)
    assert [s.num_attempts for s in gstats] == [s.n_attempts for s in ostats]
    assert [list(s.step_accepted[:s.num_attempts]) for s in gstats] == [list(s.accepted[:s.n_attempts]) for s in ostats]


def test_gradient_tolerance_is_a_max_norm_test(oracle):
    """Ceres ends a solve before the first step when the max-norm of the gradient over the free parameters is <= gradient_tolerance = 1e-10
    (trust_region_minimizer.cc; the call site nls_solver.cpp:296-337 keeps the default).  With every term switched off except a data term weighted 1e-13 the
    gradient is ~1e-12: the oracle takes no LM attempt and leaves the fields alone, and so must the device (until round 4 it tested |g|_2 == 0 and stepped)."""
    sc = helpers.small_scene(seed=21, radius_vox=9, K=4, width=96, height=72)
    thres = 2.0 * float(sc["voxel_size"])
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres, lambda_g=1e-13, lambda_r0=0.0, lambda_r1=0.0, lambda_s0=0.0, lambda_s1=0.0, lambda_a=0.0)
    assert rc == 0
    assert [s.n_attempts for s in ostats] == [0, 0] and [s.num_attempts for s in gstats] == [0, 0]
    for so, sg in zip(ostats, gstats):
        assert list(so.rows) == list(sg.rows)
        assert abs(so.cost_initial - sg.cost_initial) <= 1e-4 * so.cost_initial and abs(so.cost_final - sg.cost_final) <= 1e-4 * so.cost_final
    assert np.array_equal(out["sdf_refined"], ref["sdf_refined"]) and np.array_equal(out["albedo"], ref["albedo"])
    np.testing.assert_array_equal(np.asarray(cam[2]).ravel(), np.asarray(ocam[2]).ravel())
    # ... and a gradient of ~1e-8 (data term weighted 1e-9) is NOT converged: both step
    rc, ref, ocam, ostats, out, cam, gstats = _run_both(oracle, sc, thres, lambda_g=1e-9, lambda_r0=0.0, lambda_r1=0.0, lambda_s0=0.0, lambda_s1=0.0, lambda_a=0.0)
    assert rc == 0 and ostats[0].n_attempts >= 1 and gstats[0].num_attempts >= 1


@pytest.mark.parametrize("world", [2, 4, 8])
def test_mailbox_transport_across_processes(world):
    """tools/p2p_ipc_selftest: the mailbox transport between PROCESSES — IPC handles of the fine-grained mailboxes exchanged over pipes, no
    RCCL — all-reduces of 4 and 1210 doubles and rim pushes between every pair, 100 rounds, every value checked.  (One device here, so the
    ranks share it; on a multi-GPU node the same binary puts every rank on its own device.)"""
    import os, re, subprocess
    import torch
    if world == 8 and torch.cuda.device_count() < 8:
        # measured on a 1-GPU box: eight PROCESSES whose kernels poll each other are not co-scheduled on one device (ranks time out in round 0 / 1 after their
        # bounded waits, exit code 15) — four are.  The 8-rank case needs a device per rank, which is what a node gives it.
        pytest.skip("8 mutually polling processes need 8 devices (they are not co-scheduled on one)")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "p2p_ipc_selftest")
    assert os.path.exists(exe), "tools/p2p_ipc_selftest has not been built (run __graft_entry__.build())"
    # 8 ranks = a node's worth of processes (here sharing one device: their polling kernels time-slice, so fewer rounds; every wait is bounded)
    # Ranks that SHARE a device depend on the driver co-scheduling the polling kernels of several processes: measured 1 run in 12 at world 4 in which they
    # were not and every rank left through its bounded wait (exit code 15, rounds 0 / 1).  That is the environment, not the protocol: such a run is repeated
    # (at most twice); a wrong VALUE (any other failure code) fails at once, and with a device per rank there is no retry.
    shared = torch.cuda.device_count() < world
    for attempt in range(3 if shared else 1):
        r = subprocess.run([exe, str(world), "100" if world < 8 else "30"], capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            break
        codes = set(re.findall(r"failure (\d+) in round", r.stdout + r.stderr))
        if not (shared and codes == {"15"}):
            break
    assert r.returncode == 0, r.stdout + r.stderr


def test_device_hash_table_under_collisions():
    """The device's voxel table is open addressing with linear probing over a 64-bit finaliser of the packed key (grid_kernels.hip: pack_key, mix64); every neighbour of every stencil
    goes through it once per grid.  The finaliser is a bijection, so keys can be CRAFTED to land on one slot: 2000 voxels whose home slot is the same (a probe chain 2000 long — the
    table's worst case), each with its +x neighbour stored as well (ordinary slots), scattered over the whole +-2^20 coordinate range the packed key supports.  The neighbour table
    must be exactly what a dictionary gives."""
    from intrinsic3d_amd import binding
    M = (1 << 64) - 1
    C1, C2 = 0xff51afd7ed558ccd, 0xc4ceb9fe1a85ec53
    I1, I2 = pow(C1, -1, 1 << 64), pow(C2, -1, 1 << 64)
    def unmix(y):
        y ^= y >> 33; y = (y * I2) & M; y ^= y >> 33; y = (y * I1) & M; y ^= y >> 33
        return y
    def mix(k):
        k ^= k >> 33; k = (k * C1) & M; k ^= k >> 33; k = (k * C2) & M; k ^= k >> 33
        return k
    n_c = 2000; cap = 8192                                   # 2 x 4000 voxels -> 8192 slots
    rng = np.random.default_rng(17)
    keys = []
    seen = set()
    while len(keys) < n_c:
        y = (int(rng.integers(0, 1 << 62)) << 13 | 5) & M   # low 13 bits = 5: home slot 5 of an 8192-slot table; the rest random
        k = unmix(y)
        if k >> 63:
            continue
        x, yy, z = (k & 0x1fffff) - (1 << 20), ((k >> 21) & 0x1fffff) - (1 << 20), ((k >> 42) & 0x1fffff) - (1 << 20)
        if max(abs(x), abs(yy), abs(z)) > (1 << 20) - 8 or (x, yy, z) in seen or (x + 1, yy, z) in seen or (x - 1, yy, z) in seen:
            continue
        assert mix(k) & (cap - 1) == 5
        seen.add((x, yy, z)); keys.append((x, yy, z))
    allk = np.array(keys + [(x + 1, y, z) for x, y, z in keys], np.int32)
    allk = allk[rng.permutation(len(allk))]
    n = len(allk); assert n == 2 * n_c
    with binding.Context(0) as ctx:
        ctx.set_grid(0.004, allk, np.zeros(n), np.zeros(n), np.full(n, 0.6), np.ones(n, np.float32), np.full((n, 3), 128, np.uint8))
        nb = ctx.debug_neighbors()
    index = {tuple(k): i for i, k in enumerate(allk.tolist())}
    offs = [(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1),(2,0,0),(0,2,0),(0,0,2),(1,1,0),(1,0,1),(0,1,1),
            (-2,0,0),(0,-2,0),(0,0,-2),(-1,-1,0),(-1,0,-1),(0,-1,-1)]
    exp = np.full_like(nb, -1)
    for j, o in enumerate(offs):
        exp[:, j] = [index.get((k[0] + o[0], k[1] + o[1], k[2] + o[2]), -1) for k in allk.tolist()]
    assert np.array_equal(nb, exp) and (exp[:, 0] >= 0).sum() == n_c and (exp[:, 1] >= 0).sum() == n_c
