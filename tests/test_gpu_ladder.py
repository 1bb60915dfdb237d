"""-m gpu: the damping ladder of the trust-region loop (solver.cpp lm_solve, tile_pass_mr.hip).

The reference restarts its trust region at 1e4 in every outer iteration (optimizer.cpp:138 builds a fresh NLSSolver; nls_solver.cpp:322-323 never takes effect), so
most LM attempts are rejected, and after a rejection the next radius is known in advance (LevenbergMarquardtStrategy::StepRejected).  The library therefore SOLVES up to
I3D_LADDER consecutive attempts together — PCG systems that differ only in the LM diagonal, iterated in lock step, one stream of the stored rows for up to three of
them (k_eg_tile_mr) — and then DECIDES them one after the other with the same kernel as the serial loop.  Nothing about the result may change.  In the bit-reproducible
mode (I3D_DETERMINISTIC=1), fields, camera, costs, attempts, accept / reject sequence, PCG iteration counts and final radius must be equal BIT FOR BIT between

  (A) the serial loop of round 4 (I3D_LADDER=1, k_eg_tile) and the ladder driving that same kernel once per system (I3D_LADDER_MR=0): the lock-step solve, the
      speculation and the decision chain change nothing;
  (B) the serial loop streaming its rows through k_eg_tile_mr<1> (I3D_EGT_MR1=1) and the ladder at every depth and grouping (k_eg_tile_mr<1,2,3>): a system's result
      does not depend on which other systems share its stream of the rows.
k_eg_tile_mr against k_eg_tile itself is equality to fp32 round-off (its wave sums are DPP trees with another association): checked as such.  In the default mode
(fp32 LDS atomics inside k_eg_tile) everything agrees to round-off with the same attempts and accept sequence.  An invalid step (radius halved instead of divided)
puts a batch out of step: the attempt behind it must be solved again, alone, with the radius the trust region really reached — bit for bit the serial answer.
Oracle parity of the ladder itself is what the rest of the suite checks: the ladder is the default, every other GPU test runs through it."""
import numpy as np
import pytest

import helpers
from test_gpu_bench_parity import build_slice, _bench_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def slice_setup(oracle):
    return build_slice(oracle)


@pytest.fixture(autouse=True)
def same_tile_geometry(monkeypatch):
    """The multi-system pass exists in the 512-entry tile geometry; the serial loop it is compared with must tile the same way (the halo sums of a tile are partial
    sums: another tiling is another summation order)."""
    monkeypatch.setenv("I3D_EGT_TILE", "512")


def _run(S, iterations=2, cg_fixed=-1):
    O = S["O"]; sc = S["sc"]; a0 = S["arrays"]
    cfg = helpers.gpu_cfg(_bench_cfg(O, S["thres"], cg_fixed)); cfg.iterations = iterations
    ctx = helpers.gpu_context(sc, a0, S["vsh"])
    st = ctx.optimize(cfg); sdf, alb = ctx.get_grid(); cam = ctx.get_camera(); lad = ctx.debug_ladder_stats(); ctx.close()
    return st, sdf, alb, cam, lad


def _stats(st):
    return [(s.num_attempts, list(s.step_accepted[:s.num_attempts]), list(s.pcg_iterations[:s.num_attempts]), s.cost_initial, s.cost_final, s.final_radius, s.termination,
             list(s.rows)) for s in st]


def _same(a, b):
    st1, s1, a1, c1, _ = a; st2, s2, a2, c2, _ = b
    assert _stats(st1) == _stats(st2), (_stats(st1), _stats(st2))
    diffs = {"sdf": float(np.abs(s1 - s2).max()), "albedo": float(np.abs(a1 - a2).max()), "intr": float(np.abs(c1[0] - c2[0]).max()), "dist": float(np.abs(c1[1] - c2[1]).max()),
             "poses": float(np.abs(c1[2] - c2[2]).max())}
    assert np.array_equal(s1, s2) and np.array_equal(a1, a2) and all(np.array_equal(x, y) for x, y in zip(c1, c2)), diffs


_serial = {}


def _serial_det(S, monkeypatch, mr1):
    """two chained iterations of the serial loop in the bit-reproducible mode (computed once per module): mr1 = its rows through k_eg_tile_mr<1> instead of k_eg_tile"""
    key = "mr1" if mr1 else "egt"
    if key not in _serial:
        monkeypatch.setenv("I3D_DETERMINISTIC", "1"); monkeypatch.setenv("I3D_LADDER", "1"); monkeypatch.setenv("I3D_EGT_MR1", "1" if mr1 else "0")
        _serial[key] = _run(S)
        assert _serial[key][4]["batches"] == 0 and _serial[key][4]["depth"] == 1
        assert sum(s.num_attempts for s in _serial[key][0]) >= 6          # rejected attempts are what the ladder is about
    return _serial[key]


@pytest.mark.parametrize("depth", ["6", "2"])
def test_ladder_control_flow_is_the_serial_loop_bit_for_bit(slice_setup, monkeypatch, depth):
    """(A): the round-4 operator kernel once per system inside the lock-step solve"""
    serial = _serial_det(slice_setup, monkeypatch, mr1=False)
    monkeypatch.setenv("I3D_DETERMINISTIC", "1"); monkeypatch.setenv("I3D_LADDER", depth); monkeypatch.setenv("I3D_LADDER_MR", "0"); monkeypatch.setenv("I3D_EGT_MR1", "0")
    lad = _run(slice_setup)
    st = lad[4]
    assert st["batches"] >= 2 and st["depth"] == int(depth) and st["resyncs"] == 0 and st["row_streams"] == st["system_passes"], st
    _same(serial, lad)


# (depth, systems per row stream)
@pytest.mark.parametrize("depth,group", [("6", "3"), ("6", "2"), ("3", "3"), ("2", "3"), ("6", "1")])
def test_a_system_does_not_depend_on_the_systems_it_shares_the_rows_with(slice_setup, monkeypatch, depth, group):
    """(B): k_eg_tile_mr<1> alone in the serial loop vs k_eg_tile_mr<NB> in batches"""
    serial = _serial_det(slice_setup, monkeypatch, mr1=True)
    monkeypatch.setenv("I3D_DETERMINISTIC", "1"); monkeypatch.setenv("I3D_EGT_MR1", "0")
    monkeypatch.setenv("I3D_LADDER", depth); monkeypatch.setenv("I3D_LADDER_GROUP", group); monkeypatch.setenv("I3D_LADDER_MR", "1"); monkeypatch.setenv("I3D_LADDER_MR1", "1")
    lad = _run(slice_setup)
    st = lad[4]
    assert st["batches"] >= 2 and st["depth"] == int(depth) and st["resyncs"] == 0, st
    if group != "1":
        assert st["row_streams"] < st["system_passes"], st          # rows were shared
    _same(serial, lad)


def test_multi_system_kernel_against_the_round_4_kernel(slice_setup, monkeypatch):
    """k_eg_tile_mr<1> vs k_eg_tile in the serial loop: the same operator up to the association of its wave sums.  The outer iteration with the rejected attempts, from
    identical inputs (_second_iteration_start: two CHAINED iterations compared across kernels are bistable at the parity bar's scale — a near-tie of the reference's top-5 cut)."""
    monkeypatch.setenv("I3D_DETERMINISTIC", "1"); monkeypatch.setenv("I3D_LADDER", "1")
    monkeypatch.setenv("I3D_EGT_MR1", "0"); a = _run_second(slice_setup)
    monkeypatch.setenv("I3D_EGT_MR1", "1"); b = _run_second(slice_setup)
    st1, s1, a1, c1, _ = a; st2, s2, a2, c2, _ = b
    assert st1[0].num_attempts >= 4
    assert [(s.num_attempts, list(s.step_accepted[:s.num_attempts]), list(s.pcg_iterations[:s.num_attempts])) for s in st1] == \
           [(s.num_attempts, list(s.step_accepted[:s.num_attempts]), list(s.pcg_iterations[:s.num_attempts])) for s in st2]
    for x, y in zip(st1, st2):
        assert abs(x.cost_final - y.cost_final) <= 1e-7 * abs(x.cost_final)
    assert np.abs(s1 - s2).max() <= 1e-6 * np.abs(s1).max() and np.abs(a1 - a2).max() <= 1e-6 * np.abs(a1).max()
    np.testing.assert_allclose(c2[2], c1[2], rtol=1e-5, atol=1e-7)


def test_ladder_with_fixed_pcg_depth_and_residual_resets(slice_setup, monkeypatch):
    """30 PCG iterations per attempt: every system of a batch goes through the residual reset (r = b - A x) at iterations 10, 20, 30 in lock step."""
    monkeypatch.setenv("I3D_DETERMINISTIC", "1")
    monkeypatch.setenv("I3D_LADDER", "1"); monkeypatch.setenv("I3D_EGT_MR1", "1")
    serial = _run(slice_setup, iterations=1, cg_fixed=30)
    monkeypatch.setenv("I3D_LADDER", "6"); monkeypatch.setenv("I3D_EGT_MR1", "0"); monkeypatch.setenv("I3D_LADDER_MR1", "1")
    _same(serial, _run(slice_setup, iterations=1, cg_fixed=30))


def test_ladder_in_the_default_mode_agrees_to_round_off(slice_setup, monkeypatch):
    """the ladder (k_eg_tile_mr) against the serial loop (k_eg_tile) in the default mode, the outer iteration with the rejected attempts from identical inputs"""
    monkeypatch.setenv("I3D_LADDER", "1")
    st1, s1, a1, c1, _ = _run_second(slice_setup)
    monkeypatch.setenv("I3D_LADDER", "6")
    st2, s2, a2, c2, lad = _run_second(slice_setup)
    assert lad["row_streams"] < lad["system_passes"] and st1[0].num_attempts >= 4
    assert [(s.num_attempts, list(s.step_accepted[:s.num_attempts])) for s in st1] == [(s.num_attempts, list(s.step_accepted[:s.num_attempts])) for s in st2]
    for x, y in zip(st1, st2):       # Ceres' stop test compares i (Q1 - Q0) / Q1 with 0.1: summation-order noise may move a count by one
        assert all(abs(int(p) - int(q)) <= 1 for p, q in zip(x.pcg_iterations[:x.num_attempts], y.pcg_iterations[:y.num_attempts]))
        assert abs(x.cost_final - y.cost_final) <= 1e-6 * abs(x.cost_final)
    assert np.abs(s1 - s2).max() <= 1e-5 * np.abs(s1).max() and np.abs(a1 - a2).max() <= 1e-5 * np.abs(a1).max()
    np.testing.assert_allclose(c2[2], c1[2], rtol=1e-5, atol=1e-7)


def test_an_invalid_step_puts_the_batch_out_of_step_and_it_is_solved_again(slice_setup, monkeypatch):
    """TrustRegionMinimizer::HandleInvalidStep halves the radius and leaves the reduction factor alone: the systems behind the invalid attempt were solved for radii the
    trust region never reaches.  The step of attempt 1 is DECLARED invalid by a test switch (on real data model_cost_change <= 0 does not occur): in the second
    iteration attempt 0 is rejected (radius 5000, factor 4), attempt 1 — the first of the batch {1, 2} — is invalid (radius 2500), and system 2 was solved for 1250."""
    monkeypatch.setenv("I3D_DETERMINISTIC", "1")
    monkeypatch.setenv("I3D_DEBUG_INVALID_ATTEMPT", "1")
    monkeypatch.setenv("I3D_LADDER", "1"); monkeypatch.setenv("I3D_EGT_MR1", "1")
    serial = _run(slice_setup, iterations=2)
    assert serial[0][1].num_attempts >= 3 and serial[0][1].step_accepted[1] == 0
    monkeypatch.setenv("I3D_LADDER", "6"); monkeypatch.setenv("I3D_EGT_MR1", "0"); monkeypatch.setenv("I3D_LADDER_MR1", "1")
    lad = _run(slice_setup, iterations=2)
    _same(serial, lad)
    assert lad[4]["resyncs"] >= 1, lad[4]


_second = {}


def _second_iteration_start(S):
    """The state the SECOND outer iteration of the bench slice starts from (the first one, one accepted attempt, run once on a single rank in the bit-reproducible default
    mode): the iteration with the rejected attempts — the one the ladder is about — then starts from IDENTICAL inputs in every variant compared below.  (Chaining the two
    iterations inside every variant instead makes the comparison bistable: the variants' first results differ by round-off, which is enough to swap two keyframes of
    near-equal weight at the top-5 cut of one voxel, colorization.cpp:357-370 — a discrete decision of the reference algorithm — and the fields around it then end 5.3e-4
    apart in about half of the runs, sharded or not: tools/experiments/sharded_ladder_diag.py, and _run_both of test_gpu_bench_parity.py for the same voxel in round 1.)"""
    if "start" not in _second:
        O = S["O"]; sc = S["sc"]
        cfg = helpers.gpu_cfg(_bench_cfg(O, S["thres"], -1, second=False))
        ctx = helpers.gpu_context(sc, S["arrays"], S["vsh"])
        st = ctx.optimize(cfg); sdf, alb = ctx.get_grid(); cam = ctx.get_camera(); ctx.close()
        assert st[0].successful_steps == 1
        arrays = dict(S["arrays"]); arrays["sdf_refined"] = sdf; arrays["albedo"] = alb
        sc2 = dict(sc); sc2["intr"], sc2["dist"], sc2["poses"] = cam
        _second["start"] = (sc2, arrays)
    return _second["start"]


def _run_second(S, cg_fixed=-1):
    """the second outer iteration alone, single rank"""
    sc2, arrays = _second_iteration_start(S)
    cfg = helpers.gpu_cfg(_bench_cfg(S["O"], S["thres"], cg_fixed, second=True))
    ctx = helpers.gpu_context(sc2, arrays, S["vsh"])
    st = ctx.optimize(cfg); sdf, alb = ctx.get_grid(); cam = ctx.get_camera(); lad = ctx.debug_ladder_stats(); ctx.close()
    return st, sdf, alb, cam, lad


def _run_ranks(S, W, cg_fixed=-1):
    """W simulated ranks (host threads on one GPU, i3d_comm_init_sim) through the second outer iteration of the bench slice: per rank (stats, sdf, albedo, camera, ladder stats, comm stats)"""
    import threading
    from intrinsic3d_amd import binding
    sc2, arrays = _second_iteration_start(S)
    cfg = helpers.gpu_cfg(_bench_cfg(S["O"], S["thres"], cg_fixed, second=True))
    L = binding.load()
    shared = L.i3d_comm_sim_create(W)
    ctxs = [helpers.gpu_context(sc2, arrays, S["vsh"]) for _ in range(W)]
    for r, c in enumerate(ctxs):
        c.comm_init_sim(shared, r)
    out = [None] * W; err = [None] * W

    def run(r):
        try:
            out[r] = ctxs[r].optimize(cfg)
        except Exception as e:      # surface failures instead of deadlocking the other ranks silently
            err[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert not any(t.is_alive() for t in th), "sharded run hung"
    assert all(e is None for e in err), err
    res = []
    for r, c in enumerate(ctxs):
        sdf, alb = c.get_grid(); res.append((out[r], sdf, alb, c.get_camera(), c.debug_ladder_stats(), c.comm_stats())); c.close()
    L.i3d_comm_sim_destroy(shared)
    return res


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_ranks_run_the_ladder(slice_setup, world):
    """Round 6: the damping ladder in the SHARDED path (the shard key of the reference is subvolumes.cpp:281-295, here tile ranges of the brick-ordered
    work list).  W ranks — simulated by W host threads on one GPU, every exchange through the Comm interface the RCCL transport implements — solve the rejected attempts
    of an outer iteration together: k_eg_tile_mr<NB, GHOSTS> over own + ghost tiles, ONE all-reduce of the batch's slice sums and ONE of its [camera block | p.q] per
    pass, ONE rim message of B values per entry.  Against the single-rank ladder on the bench slice (every group free, Ceres' own PCG stop; the outer iteration with the
    rejected attempts, from identical inputs): same rows, same attempts, same accept / reject sequence; PCG counts equal (+-1 on rejected attempts: the ranks' partial
    sums are associated differently, and a stop test may sit on its threshold); costs, fields, intrinsics and poses to 1e-5 in the max-norm (measured: 1e-7 ... 4e-7).
    And the ladder did run on every rank (batches > 0, fewer row streams than system passes), with the batch's exchanges in ONE message each.  (8 ranks: shares of a dozen
    tiles, every tile next to a foreign one.)"""
    rst, rsdf, ralb, rcam, rlad = _run_second(slice_setup)
    assert rlad["batches"] > 0 and rst[0].num_attempts >= 4
    for rank, (st, sdf, alb, cam, lad, comm) in enumerate(_run_ranks(slice_setup, world)):
        assert lad["batches"] > 0 and lad["depth"] > 1 and lad["row_streams"] < lad["system_passes"], (rank, lad)
        s1, s2 = rst[0], st[0]
        what = (world, rank, _stats([s1]), _stats([s2]))
        assert list(s1.rows) == list(s2.rows) and s1.num_attempts == s2.num_attempts, what
        assert list(s1.step_accepted[:s1.num_attempts]) == list(s2.step_accepted[:s2.num_attempts]), what
        p1 = list(s1.pcg_iterations[:s1.num_attempts]); p2 = list(s2.pcg_iterations[:s2.num_attempts])
        assert all(abs(x - y) <= 1 for x, y in zip(p1, p2)) and p1[-1] == p2[-1], what
        assert abs(s1.cost_initial - s2.cost_initial) <= 1e-12 * s1.cost_initial and abs(s1.cost_final - s2.cost_final) <= 1e-5 * s1.cost_final, what
        e = (float(np.abs(sdf - rsdf).max() / np.abs(rsdf).max()), float(np.abs(alb - ralb).max() / np.abs(ralb).max()))
        print(f"\n[sharded ladder, {world} ranks, rank {rank}] sdf {e[0]:.1e} albedo {e[1]:.1e} of the field maximum against the single-rank ladder; exchanges {comm['reduce_calls']} all-reduces, {comm['halo_calls']} rim messages "
              f"for {lad['system_passes']} system passes in {lad['row_streams']} streams")
        assert max(e) <= 1e-5, e
        np.testing.assert_allclose(cam[0], rcam[0], rtol=1e-5); np.testing.assert_allclose(cam[2], rcam[2], rtol=1e-5, atol=1e-7)
        # exchanges of the PCG passes: two all-reduces and one rim message per pass of a BATCH, not per system
        assert comm["reduce_calls"] < 2 * lad["system_passes"] and comm["halo_calls"] < lad["system_passes"], (comm, lad)


def test_sharded_ladder_with_fixed_pcg_depth_equals_the_sharded_serial_loop(slice_setup, monkeypatch):
    """The sharded ladder against the sharded SERIAL loop (I3D_LADDER=1: k_eg_tile<GHOSTS>, the six-launch pass) at a fixed PCG depth of 12 (one residual reset), the outer
    iteration with the rejected attempts from identical inputs: with no stop test to sit on a threshold, attempts, accept sequence and PCG counts must be identical, and
    the fields agree to the round-off of two operator kernels whose wave sums are associated differently (k_eg_tile_mr against k_eg_tile: as
    test_multi_system_kernel_against_the_round_4_kernel) — far below the parity bar."""
    lad = _run_ranks(slice_setup, 2, cg_fixed=12)
    monkeypatch.setenv("I3D_LADDER", "1")
    ser = _run_ranks(slice_setup, 2, cg_fixed=12)
    for (st1, s1, a1, c1, l1, _), (st2, s2, a2, c2, l2, _) in zip(lad, ser):
        assert l1["batches"] > 0 and l2["batches"] == 0 and st1[0].num_attempts >= 4
        assert [x[:3] for x in _stats(st1)] == [x[:3] for x in _stats(st2)], (_stats(st1), _stats(st2))
        assert np.abs(s1 - s2).max() <= 1e-5 * np.abs(s2).max() and np.abs(a1 - a2).max() <= 1e-5 * np.abs(a2).max()
        np.testing.assert_allclose(c1[0], c2[0], rtol=1e-5); np.testing.assert_allclose(c1[2], c2[2], rtol=1e-5, atol=1e-7)


def test_sharded_ladder_without_the_multi_system_pass(slice_setup, monkeypatch):
    """I3D_LADDER_MR=0: the batch's exchanges stay one message each, the rows are streamed once per system through k_eg_tile<..., GHOSTS> (the control of round 5's ladder tests,
    now sharded): against the same switch on one rank — same attempts, accept sequence, PCG counts (+-1 on rejected attempts), fields to 1e-5."""
    monkeypatch.setenv("I3D_LADDER_MR", "0")
    rst, rsdf, ralb, rcam, rlad = _run_second(slice_setup)
    assert rlad["batches"] > 0 and rlad["row_streams"] == rlad["system_passes"]
    for rank, (st, sdf, alb, cam, lad, comm) in enumerate(_run_ranks(slice_setup, 2)):
        assert lad["batches"] > 0 and lad["row_streams"] == lad["system_passes"], (rank, lad)
        s1, s2 = rst[0], st[0]
        assert list(s1.rows) == list(s2.rows) and list(s1.step_accepted[:s1.num_attempts]) == list(s2.step_accepted[:s2.num_attempts]), (_stats([s1]), _stats([s2]))
        p1 = list(s1.pcg_iterations[:s1.num_attempts]); p2 = list(s2.pcg_iterations[:s2.num_attempts])
        assert all(abs(x - y) <= 1 for x, y in zip(p1, p2)) and p1[-1] == p2[-1], (p1, p2)
        assert np.abs(sdf - rsdf).max() <= 1e-5 * np.abs(rsdf).max() and np.abs(alb - ralb).max() <= 1e-5 * np.abs(ralb).max()
        assert comm["reduce_calls"] < 2 * lad["system_passes"]
