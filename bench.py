#!/usr/bin/env python3
"""bench.py — Gauss-Newton iteration throughput of the voxel-SDF shading optimisation (BASELINE.json metric).

A "step" is one outer iteration of Optimizer::optimize (optimizer.cpp:119-170): observation pass + residual/Jacobian
build + cost-term normalisation + one Levenberg-Marquardt solve (PCG on the normal equations, cost evaluation, step
acceptance).  The timed region is a sequence of i3d_optimize calls of (up to) 10 iterations each — the reference's own
call shape, lambda schedule included (intrinsic3d.yml: iterations 10) — that together run `--steps` iterations on inputs
that are already resident in HBM; `--warmup` iterations run in a separate untimed call first.

Workload (config.workload): BASELINE.json configs[3] shrunk to one node's worth of work per rank — a seeded synthetic
hashed grid of ~8M stored voxels at 1 mm (thin shell around a bumpy sphere, finest-level shell threshold), 200 keyframes
of 640x480 on a Fibonacci sphere, spatially-varying SH lighting estimated on the device (i3d_estimate_sh, untimed: it runs once
per level, not per iteration) on the ~512 subvolumes of 0.06 m the shell touches, all parameter groups free.  With --gpus N the SAME
problem is sharded across the ranks (strong scaling): tile-aligned ownership of the brick-ordered work list, rows of a thin rim
recomputed as ghosts, per PCG pass one neighbour exchange of the operator input on the rim + one small all-reduce.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
KERNEL_TAG = "r06-batched"      # bumped when k_build<true> / k_eg_tile change materially: PMC traffic / SQ counter files of older kernels are not attached
# issue cost of a VALU wave-instruction on gfx950 measured with tools/experiments/valu_rate.hip (profiles/r02_valu_rate.txt): cycles per instruction on one SIMD
# cycles per wave-instruction on one SIMD BY OCCUPANCY (waves per SIMD), measured (tools/experiments/valu_rate.hip, profiles/r03_valu_rate.txt); pk = packed fp32 (v_pk_*).
# A kernel is priced at the occupancy it actually runs at (k_build<true>: 247 VGPRs = 2 waves per SIMD), not at the 4-wave rates.
VALU_CYCLES_BY_OCC = {1: {"f64": 9.8, "f32": 7.3, "pk": 8.6}, 2: {"f64": 6.45, "f32": 4.27, "pk": 6.30}, 3: {"f64": 5.85, "f32": 3.74, "pk": 5.75}, 4: {"f64": 5.24, "f32": 3.21, "pk": 5.20}}      # (3: mean of the measured 2- and 4-wave rates)
KERNEL_OCCUPANCY = {"build": 2, "cost": 3, "eg_pass": 4, "eg_mr2": 2, "eg_mr3": 2, "observe": 4}      # waves per SIMD from the register counts (tools/kernel_resources.sh)
GPU_CLOCK_HZ = 2.4e9; NUM_SIMD = 1024


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--verbose", action="store_true", help="print every trust-region attempt (candidate cost, model change, rho, radius) on stderr")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--voxels", type=float, default=8.0e6, help="stored voxels of the synthetic grid")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--voxel-size", type=float, default=0.001)
    ap.add_argument("--band", type=float, default=3.5, help="stored half-thickness of the shell in voxels")
    ap.add_argument("--shell", type=float, default=1.0, help="thin-shell factor (thin_shell_factor_final)")
    ap.add_argument("--subvolume", type=float, default=0.06, help="SH subvolume size in metres (chosen so that the shell of the 0.6 m object touches ~512 subvolumes, BASELINE.json configs[3])")
    ap.add_argument("--cpu-sample", type=float, default=1.0e6, help="stored voxels of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--spin-up", type=float, default=0.0, help="seconds of device copies before the warm-up steps (the device idles while the host generates the scene; 0 = none)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--carry-radius", action="store_true",
                    help="NOT the reference's behaviour (and not the headline): carry the trust-region radius across outer iterations, as nls_solver.cpp:322-323 intends")
    ap.add_argument("--force-collectives", action="store_true",
                    help="experiments only (1 GPU): run the sharded code path through a real 1-rank communicator, to see what the exchange launches cost per PCG pass (comm_us_per_pass)")
    ap.add_argument("--all-kernel-timing", action="store_true", help="HIP events around every launch (kernel_ms_total for all categories; ~8 % slower)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="experiments only: no per-launch HIP events (no roofline in the output)")
    ap.add_argument("--pcg-fixed", type=int, default=-1, help="experiments only: pin the PCG iterations per LM attempt (-1 = Ceres' stopping rule)")
    ap.add_argument("--band2-steps", type=int, default=4,
                    help="timed steps of the second leg on SURVEY.md 8(d)'s own C4 shape (the 4-voxel stored shell, --band 2: 3.2 Eg rows per voxel) whose numbers ride in the "
                         "same JSON line as value_band2 / roofline_band2 (0 = skip; skipped when --band is 2 already, when sharded, and in experiment modes)")
    ap.add_argument("--pmc-calibrate", action="store_true",
                    help="also launch a known-size device copy (1 GiB read + 1 GiB write) so that a rocprofv3 --pmc pass over this command "
                         "can calibrate FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM section); see tools/pmc_traffic.py")
    return ap.parse_args()


def build_workload(args, log):
    from intrinsic3d_amd import synthetic
    t0 = time.time()
    radius_vox = int(round(np.sqrt(args.voxels / (4.0 * np.pi * 2.0 * args.band))))
    sc = synthetic.make_scene(radius_vox=radius_vox, voxel_size=args.voxel_size, K=args.frames, width=args.width, height=args.height,
                              levels=1, band_vox=args.band, seed=args.seed, cam_dist=2.6 * radius_vox * args.voxel_size,
                              pose_noise=(0.002, 0.0035), lum_noise=0.005, bump_amp_vox=0.5, bump_freq=40.0)
    log(f"scene: {sc['keys'].shape[0]} voxels, radius {radius_vox} vox, {args.frames} frames {args.width}x{args.height} in {time.time() - t0:.1f}s")
    return sc


def grid_arrays(sc):
    n = sc["keys"].shape[0]
    sdf = sc["sdf"].astype(np.float64)
    return dict(keys=sc["keys"], sdf=sdf, sdf_refined=sdf.copy(), albedo=np.full(n, 0.6), weight=sc["weight"], color=sc["color"])


CALL_ITERATIONS = 10          # Optimizer::optimize runs `iterations: 10` per call (intrinsic3d.yml); the lambda schedule spans ONE call


def make_cfg(binding, args, iterations, thres):
    # shipped data/intrinsic3d.yml values (lambda schedule, 10 iterations, 50 LM steps, 5 observations, 2 cm occlusion)
    return binding.default_config(iterations=iterations, lm_steps=50, lambda_g=0.2, lambda_r0=80.0, lambda_r1=10.0, lambda_s0=120.0,
                                  lambda_s1=10.0, lambda_a=0.1, fix_poses=0, fix_intrinsics=0, fix_distortion=0,
                                  occlusion_distance=0.02, num_observations=5, thres_shell=thres, grid_level=0, rgbd_level=0,
                                  pcg_fixed_iterations=args.pcg_fixed, verbose=1 if getattr(args, 'verbose', False) else 0, carry_trust_radius=1 if args.carry_radius else 0)


def cpu_baseline(args, sc, thres, log, device=0, threaded_leg=True):
    """The restated CPU reference (oracle, fp64, 8 OpenMP threads in the solve like options.num_threads = 8) on a bounded
    spatial sample of the SAME workload: the voxels of a cap of the sphere, same keyframes, same configuration."""
    from oracle import oracle_py as O
    O.build()
    keys = sc["keys"]
    n_target = int(args.cpu_sample)
    if n_target <= 0 or keys.shape[0] == 0:
        return None
    x = keys[:, 0]
    cut = np.partition(x, keys.shape[0] - min(n_target, keys.shape[0]))[keys.shape[0] - min(n_target, keys.shape[0])]
    sel = x >= cut
    g = O.Grid.from_voxels(sc["voxel_size"], keys[sel], sc["sdf"][sel], sc["weight"][sel], sc["color"][sel])
    fr = O.Frames(sc["frames"], 1)
    n = len(g)
    rc_sh, _, _, vsh, _, _ = O.estimate_sh(g, args.subvolume, 10.0, thres)       # same lighting model as the timed run (untimed here too)
    if rc_sh != 0:
        vsh = np.tile(np.asarray(sc["scene"].sh), (n, 1))
    iters = 2
    cfg = O.OptConfig(iterations=iters, lm_steps=50, lambda_g=0.2, lambda_r0=80.0, lambda_r1=10.0, lambda_s0=120.0, lambda_s1=10.0, lambda_a=0.1,
                      fix_poses=0, fix_intrinsics=0, fix_distortion=0, occlusion_distance=0.02, num_observations=5, thres_shell=thres,
                      grid_level=0, rgbd_level=0, cg_fixed_iterations=-1, verbose=0)
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    threads = int(os.environ["OMP_NUM_THREADS"])
    before = g.export()
    O.set_collect_threads(1); O.phase_seconds()                     # residual collection on one thread, as in the reference
    t0 = time.time()
    rc, o_intr, o_dist, o_poses, stats = O.optimize(g, fr, cfg, sc["intr"], sc["dist"], sc["poses"], vsh)
    dt = time.time() - t0
    phases = O.phase_seconds()
    after = g.export() if rc == 0 else None
    # ... and once more with the collection threaded (section 8(d) asks for both): ONE iteration from the same start, same rows in the same order
    threaded = None
    if rc == 0 and threaded_leg:
        g.import_fields(sdf_refined=before["sdf_refined"], albedo=before["albedo"], color=before["color"])
        cfg1 = O.OptConfig(iterations=1, lm_steps=50, lambda_g=0.2, lambda_r0=80.0, lambda_r1=10.0, lambda_s0=120.0, lambda_s1=10.0, lambda_a=0.1,
                           fix_poses=0, fix_intrinsics=0, fix_distortion=0, occlusion_distance=0.02, num_observations=5, thres_shell=thres,
                           grid_level=0, rgbd_level=0, cg_fixed_iterations=-1, verbose=0)
        O.set_collect_threads(threads); O.phase_seconds()
        t1 = time.time(); rc1, _, _, _, st1 = O.optimize(g, fr, cfg1, sc["intr"], sc["dist"], sc["poses"], vsh); dt1 = time.time() - t1
        ph1 = O.phase_seconds(); O.set_collect_threads(1)
        if rc1 == 0:
            threaded = {"threads": threads, "seconds_per_iteration_sample": dt1, "collect_s": float(ph1[0]), "build_and_solve_s": float(ph1[2]),
                        "rows_equal_single_thread": [int(x) for x in st1[0].rows] == [int(x) for x in stats[0].rows]}
    g.free(); fr.free()
    if rc != 0:
        return None
    # checker, not product: the device path on the SAME sample, same two iterations (native PCG stop, all groups free) — the problem it
    # assembles (row counts), the cost before and after every iteration, and the fields / camera it ends with
    parity = None
    try:
        from intrinsic3d_amd import binding
        with binding.Context(device) as c2:
            c2.set_grid(sc["voxel_size"], before["keys"], before["sdf"], before["sdf_refined"], before["albedo"], before["weight"], before["color"])
            c2.set_frames(sc["frames"], 1); c2.set_camera(sc["intr"], sc["dist"], sc["poses"]); c2.set_voxel_sh(vsh)
            gst = c2.optimize(binding.default_config(iterations=iters, lm_steps=50, lambda_g=0.2, lambda_r0=80.0, lambda_r1=10.0, lambda_s0=120.0, lambda_s1=10.0,
                                                     lambda_a=0.1, occlusion_distance=0.02, num_observations=5, thres_shell=thres))
            d_sdf, d_alb = c2.get_grid(); d_intr, d_dist, d_poses = c2.get_camera()

        def rel_l2(dev, ref, start):          # error of the accumulated update, not of the state it is added to
            den = float(np.linalg.norm(ref - start)); return float(np.linalg.norm(dev - ref)) / den if den > 0 else float(np.linalg.norm(dev - ref))
        parity = {"iterations": iters,
                  "rows_oracle": [[int(x) for x in st.rows] for st in stats], "rows_device": [[int(x) for x in st.rows] for st in gst],
                  "rows_equal": [int(x) for x in stats[0].rows] == [int(x) for x in gst[0].rows],
                  "cost_initial_rel_diff": abs(gst[0].cost_initial - stats[0].cost_initial) / stats[0].cost_initial,
                  "cost_final_rel_diff": [abs(a.cost_final - b.cost_final) / b.cost_final for a, b in zip(gst, stats)],
                  "lm_attempts": {"oracle": [int(st.n_attempts) for st in stats], "device": [int(st.num_attempts) for st in gst]},
                  "pcg_iterations": {"oracle": [[int(x) for x in st.cg_iters[:st.n_attempts]] for st in stats],
                                     "device": [[int(x) for x in st.pcg_iterations[:st.num_attempts]] for st in gst]},
                  "sdf_update_rel_l2_err": rel_l2(d_sdf, after["sdf_refined"], before["sdf_refined"]),
                  "albedo_update_rel_l2_err": rel_l2(d_alb, after["albedo"], before["albedo"]),
                  "sdf_max_rel_err": float(np.abs(d_sdf - after["sdf_refined"]).max() / np.abs(after["sdf_refined"]).max()),
                  "albedo_max_rel_err": float(np.abs(d_alb - after["albedo"]).max() / np.abs(after["albedo"]).max()),
                  "poses_max_abs_err": float(np.abs(np.asarray(d_poses) - np.asarray(o_poses)).max()),
                  "intrinsics_max_rel_err": float(np.abs((np.asarray(d_intr) - np.asarray(o_intr)) / np.asarray(o_intr)).max())}
        log(f"cpu baseline parity on the sample: rows equal {parity['rows_equal']}, cost rel. diff initial {parity['cost_initial_rel_diff']:.2e} "
            f"final {parity['cost_final_rel_diff']}, update rel. L2 err sdf {parity['sdf_update_rel_l2_err']:.2e} albedo {parity['albedo_update_rel_l2_err']:.2e}")
    except Exception as e:
        log(f"parity check on the sample failed to run: {e}")
    sec_per_iter_sample = dt / iters
    scale = keys.shape[0] / float(n)
    value = 1.0 / (sec_per_iter_sample * scale)
    log(f"cpu baseline: {n} voxels, {iters} iterations in {dt:.1f}s -> {sec_per_iter_sample:.2f} s/iter on the sample, x{scale:.1f} voxels")
    return {"value": value, "unit": "GN iterations/s", "cores": threads, "host_cores": os.cpu_count(), "threads": threads, "kind": "port",      # cores = the threads actually used (the solve; the residual collection runs on one, as in the reference)
           
            # `value` is measured on the sample and scaled linearly by the voxel ratio unless the sample IS the workload (tools/c4_full_parity.py: profiles/r05_c4_full_parity.json)
            "extrapolated": bool(scale > 1.001), "voxel_ratio": scale, "sample_voxels": int(n),
            "sample": f"restated CPU reference (Ceres-2.1.0-equivalent, fp64) on a {n}-voxel cap of the same grid with all {sc['K']} keyframes, "
                      f"{iters} GN iterations in {dt:.1f}s; per-iteration time scaled linearly by the voxel ratio {scale:.1f} to the full workload "
                      f"(residual collection single-threaded as in the reference, solve on {threads} threads like options.num_threads = 8; the host has {os.cpu_count()} cores)",
            "seconds_per_iteration_sample": sec_per_iter_sample,
            "phases_s_per_iteration_sample": {"collect_single_thread": float(phases[0]) / iters, "build_and_solve": float(phases[2]) / iters},
            "threaded_collection": None if threaded is None else dict(threaded, value=1.0 / (threaded["seconds_per_iteration_sample"] * scale)),
            "parity_on_sample": parity}


def kernel_table(args, sizes, world, timing_work, timing):
    """Average launch time of the two roofline kernels against SURVEY.md section 8(d)'s byte model.
       build (K2+K3):   68 A + 132 Rg + 36 Rr + 12 Rs + 16 Ra + B_img
       operator (K6):   4 nnz, nnz = 29 Rg + 7 Rr + Rs + 2 Ra  -- 8(d)'s fused single-pass J^T J p ("4 nnz + 12*4 n" is the whole PCG iteration; the 48 n of
                        vector passes belong to the vector kernels, not to this one)
       `design_GB` beside it is what THIS implementation must move per launch by construction: 120 B per stored Eg row (the 29 partials with the row weight folded in
       + the keyframe id) + per work-list entry the operator input 8, flags 6, local stencil slots 28 (18 x 12 bits), symmetric Ea weights 24, accumulators out 8
       + ~20 B per tile-halo slot (~1 per entry); Er / Es rows are not stored (constant coefficients), so their 4 nnz bytes are never read."""
    A, Rg, Rr, Rs, Ra = sizes["active"], sizes["eg"], sizes["er"], sizes["es"], sizes["ea"]
    nnz = 29.0 * Rg + 7.0 * Rr + Rs + 2.0 * Ra
    img_bytes = min(4.0 * args.frames * args.width * args.height, 256.0 * Rg)
    b_build = 68.0 * A + 132.0 * Rg + 36.0 * Rr + 12.0 * Rs + 16.0 * Ra + img_bytes
    b_egpass = 4.0 * nnz
    d_build = b_build
    d_egpass = 120.0 * Rg + (8 + 6 + 28 + 24 + 8 + 20) * float(A)
    if world > 1:                                           # a rank streams its own share of the rows (its ghost rows are not counted: conservative)
        b_build /= world; b_egpass /= world; d_build /= world; d_egpass /= world
    kernels = {}
    # ladder batches: one launch of k_eg_tile_mr<NB> streams the rows ONCE for NB systems (strict bytes: still 4 nnz); per system it stages one more input and writes one more output
    per_sys = (8 + 8) * float(A) / world
    for name, bytes_per_launch, design in (("build", b_build, d_build), ("eg_pass", b_egpass, d_egpass), ("eg_mr2", b_egpass, d_egpass + per_sys), ("eg_mr3", b_egpass, d_egpass + 2 * per_sys)):
        if name not in timing_work:
            continue
        ms, n, slow_ms, slow_n = timing_work[name]
        if n > 0:
            avg = ms / n                     # HIP events around each launch on the library's stream, no-op launches excluded
            kernels[name] = {"launches": n, "launches_incl_noop": timing[name][1], "avg_ms": avg, "algorithmic_GB": bytes_per_launch / 1e9,
                             "design_GB": design / 1e9, "achieved_GBs": bytes_per_launch / 1e9 / (avg * 1e-3),
                             # launches slower than 4x the 90th percentile are not in avg_ms (a launch that straddles a device hiccup); reported, not hidden:
                             "excluded_slow_launches": slow_n, "excluded_slow_ms": slow_ms, "avg_ms_incl_slow": (ms + slow_ms) / (n + slow_n)}
    return kernels


def roofline_of(name, kernels, Rg, A, world, attach_counters=True):
    """The `roofline` object of one kernel.  `bound` comes from COUNTERS when a committed SQ pass of the same kernel tag and workload exists:
         wait_share          = SQ_WAIT_ANY / SQ_WAVE_CYCLES         (share of resident wave-cycles spent in s_waitcnt)
         valu_issue_share    = VALU instructions x measured cycles per instruction AT THE KERNEL'S OCCUPANCY / launch time
       "latency" when the waves mostly wait and neither the memory system nor the VALU is saturated, "valu-issue" when the issue time fills the launch,
       "hbm" otherwise (the byte model against the 8 TB/s peak is the figure of merit either way)."""
    if name not in kernels:
        return None
    k = kernels[name]
    tr = pmc_traffic(name, Rg, A) if (world == 1 and attach_counters) else None
    nsys = {"eg_mr2": 2, "eg_mr3": 3}.get(name, 1)
    out = {"kernel": {"build": "k_build<true>", "eg_pass": "k_eg_tile", "eg_mr2": "k_eg_tile_mr<2>", "eg_mr3": "k_eg_tile_mr<3>"}[name], "bound": "hbm", "achieved": k["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": k["achieved_GBs"] / HBM_PEAK_GBS, "traffic": tr[0] if tr else None, "traffic_measured_in_run": False,
           "traffic_source": (f"committed PMC passes of this command, profiles/{tr[1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, calibrated; not measured in this run)" if tr else None)}
    if nsys > 1:
        # one launch streams the rows ONCE (strict bytes: 4 nnz, what `achieved` / `frac` price) and applies them to `systems_per_launch` PCG systems of a ladder batch:
        # `serial_equivalent_*` = what the serial loop would stream for the same work (4 nnz per system) per second of this launch, over the HBM peak: a RATIO that
        # exceeds 1 by construction when systems share bytes — NOT a roofline fraction (round-5 review).  The roofline figure is `frac` (strict bytes).
        out.update(systems_per_launch=nsys, serial_equivalent_GBs=k["achieved_GBs"] * nsys, serial_equivalent_bandwidth_ratio=k["achieved_GBs"] * nsys / HBM_PEAK_GBS,
                   bound=f"issue + latency at 2 waves per SIMD ({nsys} systems share every byte of the rows; no counters attached)")
    sq = sq_valu(name, Rg) if (world == 1 and attach_counters) else None
    if sq:
        occ = KERNEL_OCCUPANCY.get(name, 4); cyc = VALU_CYCLES_BY_OCC[occ]
        pk = sq.get("valu_pk", 0.0)
        issue_ms = (sq["valu_f64"] * cyc["f64"] + pk * cyc["pk"] + (sq["valu"] - sq["valu_f64"] - pk) * cyc["f32"]) / (NUM_SIMD * GPU_CLOCK_HZ) * 1e3
        cn = sq.get("counters", {})
        wait_share = cn["SQ_WAIT_ANY"] / cn["SQ_WAVE_CYCLES"] if cn.get("SQ_WAVE_CYCLES") else None
        out.update(valu_instructions=sq["valu"], valu_issue_ms=issue_ms, valu_frac=issue_ms / k["avg_ms"], occupancy_waves_per_simd=occ,
                   wait_share=wait_share, valu_active_per_busy_cycle=(cn["SQ_ACTIVE_INST_VALU"] / cn["SQ_BUSY_CYCLES"] if cn.get("SQ_BUSY_CYCLES") else None),
                   valu_source=f"profiles/{sq['source']} (SQ_INSTS_VALU, SQ_WAIT_ANY / SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES; fp64 share from the ISA; cycles per instruction at {occ} waves per SIMD)")
        if out["valu_frac"] >= 0.8:
            out["bound"] = "valu-issue" + (f" ({nsys} systems share every byte of the rows)" if nsys > 1 else "")
        elif wait_share is not None and wait_share >= 0.45 and out["frac"] < 0.55:
            out["bound"] = (f"latency (s_waitcnt {100 * wait_share:.0f} % of the wave-cycles at {occ} waves per SIMD; VALU issue {100 * out['valu_frac']:.0f} % of the launch"
                            + (f"; {nsys} systems share every byte of the rows" if nsys > 1 else "") + ")")
    return out


def band2_leg(args, binding, log, device):
    """SURVEY.md section 8(d)'s own C4 shape on the driver's record: the 4-voxel stored shell (--band 2).  The voxels on the rim of the stored band cannot own Eg
    rows (their forward stencil leaves the band), so the work list averages ~3.2 rows per entry instead of 5.0 — the case the operator pass must be robust to
    (per-group row counts bound a wave's row stream, tile_pass.hip).  Same voxel count / keyframes / configuration as the headline leg, fewer steps."""
    import copy
    a2 = copy.copy(args); a2.band = 2.0
    sc = build_workload(a2, log)
    thres = a2.shell * float(sc["voxel_size"])
    arrays = grid_arrays(sc)
    ctx = binding.Context(device)
    try:
        ctx.set_grid(sc["voxel_size"], arrays["keys"], arrays["sdf"], arrays["sdf_refined"], arrays["albedo"], arrays["weight"], arrays["color"])
        ctx.set_frames(sc["frames"], 1)
        ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
        ctx.estimate_sh(a2.subvolume, 10.0, thres)
        ctx.optimize(make_cfg(binding, a2, 1, thres))                 # warm-up
        ctx.timing_enable(True); ctx.timing_select(["eg_pass", "eg_mr2", "eg_mr3", "build"]); ctx.timing_get(reset=True)
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = ctx.optimize(make_cfg(binding, a2, a2.band2_steps, thres))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timing_work = ctx.timing_get_work_ex(); timing = ctx.timing_get(reset=True); sizes = ctx.problem_sizes()
        kernels = kernel_table(a2, sizes, 1, timing_work, timing)
        A, Rg = sizes["active"], sizes["eg"]
        return {"value": a2.band2_steps / dt, "unit": "GN iterations/s", "steps": a2.band2_steps, "warmup": 1, "ms_per_step": dt / a2.band2_steps * 1e3,
                "workload": f"as config.workload with the 4-voxel stored shell of SURVEY.md 8(d) (--band 2): {arrays['keys'].shape[0]} stored voxels, {A} active, "
                            f"{Rg} Eg rows = {Rg / max(1, A):.2f} per active voxel",
                "stored_voxels": int(arrays["keys"].shape[0]), "active_voxels": A, "rows": {"Eg": Rg, "Er": sizes["er"], "Es": sizes["es"], "Ea": sizes["ea"]},
                "lm_attempts": [int(s.num_attempts) for s in stats],
                # the dominant operator kernel of this leg, with the PMC / SQ counters of a committed pass over THIS workload when there is one (matched by row count)
                "roofline": roofline_of(max((k for k in kernels if k != "build"), key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"]), kernels, Rg, A, 1),
                "roofline_build": roofline_of("build", kernels, Rg, A, 1), "kernels": kernels, "ladder": ctx.debug_ladder_stats()}
    finally:
        ctx.close()


def sq_valu(kernel, eg_rows):
    """VALU wave-instructions per launch of `kernel` from the committed SQ-counter pass (profiles/*_sq_counters.json, SQ_INSTS_VALU), same tag / workload rule
    as pmc_traffic: the issue-time floor of the kernel = instructions x cycles-per-instruction / (SIMDs x clock)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel_tag") != KERNEL_TAG or kernel not in d.get("kernels", {}):
            continue
        if abs(d.get("eg_rows", 0) - eg_rows) <= 0.01 * eg_rows:
            best = dict(d["kernels"][kernel], source=os.path.basename(f))
    return best


def pmc_traffic(kernel, eg_rows, active):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*_pmc_traffic.json, written by tools/pmc_traffic.py from
    separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command, calibrated on a known-size copy).  Counters cannot
    be collected inside the timed run, so the figure is only attached when it was measured on the same workload (row count within 1 %; scaled by the row ratio)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        # the row count drifts by ~1e-4 between GN iterations (rows appear / vanish as the surface moves): same workload = within 1 %
        if d.get("kernel_tag") != KERNEL_TAG:
            continue
        if abs(d.get("eg_rows", 0) - eg_rows) <= 0.01 * eg_rows and abs(d.get("active_voxels", 0) - active) <= 0.01 * active and kernel in d.get("kernels", {}):
            best = (d["kernels"][kernel]["traffic_bytes_per_launch"] * (eg_rows / float(d["eg_rows"])), os.path.basename(f))
    return best


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves — one process per GPU through
    torch.distributed.run on 127.0.0.1 — and pass rank 0's JSON line through.  Never silently fewer ranks than asked for."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} visible devices, this box has {have}; refusing to run fewer ranks than asked for", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    if "WORLD_SIZE" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False); pre.add_argument("--gpus", type=int, default=1)
        n = pre.parse_known_args()[0].gpus
        if n > 1:
            sys.exit(spawn_ranks(n))
    # stdout carries exactly ONE line (the JSON): anything native libraries print on fd 1 meanwhile (RCCL prints a version banner when a
    # communicator is created, partly buffered until exit) goes to stderr; the JSON line is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1); os.dup2(2, 1); RESULT_FD[0] = real_stdout
    _main()          # fd 1 stays on stderr for the rest of the process: C stdio buffers of native libraries are flushed at exit


def _emit(line):
    """the JSON line goes to the process's real stdout (saved in fd RESULT_FD)"""
    sys.stdout.flush()
    os.write(RESULT_FD[0], (line + "\n").encode())


RESULT_FD = [1]


def _main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world > 1:
            args.gpus = world                 # the launcher's world size wins (the driver passes both, consistently)
        elif args.gpus > 1:
            raise SystemExit(f"bench.py: --gpus {args.gpus} under a launcher that set WORLD_SIZE=1; refusing to report a 1-rank run as {args.gpus} GPUs")

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # ONE device communicator per job: the library's own (RCCL, bootstrapped below from a unique id).  The launcher-side group only carries that id, the
        # barriers around the timed region and the max over the ranks' clocks — host-side traffic, so gloo (round 3 opened a second NCCL communicator for it).
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from intrinsic3d_amd import binding
    sc = build_workload(args, log)
    thres = args.shell * float(sc["voxel_size"])
    arrays = grid_arrays(sc)

    ctx = binding.Context(local_rank)
    if world == 1 and args.force_collectives:
        os.environ["I3D_FORCE_COLLECTIVES"] = "1"
        ctx.comm_init(0, 1, binding.Context.comm_unique_id())
    if world > 1:
        # one process per GPU: RCCL communicator of the library, bootstrapped through torch.distributed (unique id from rank 0)
        box = [binding.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(rank, world, box[0])
    t0 = time.time()
    ctx.set_grid(sc["voxel_size"], arrays["keys"], arrays["sdf"], arrays["sdf_refined"], arrays["albedo"], arrays["weight"], arrays["color"])
    ctx.set_frames(sc["frames"], 1)
    ctx.set_camera(sc["intr"], sc["dist"], sc["poses"])
    torch.cuda.synchronize(); t_upload = time.time() - t0
    log(f"upload + hash/neighbour build: {t_upload:.2f}s")
    t0 = time.time()
    sh_sub, _, sh_stats = ctx.estimate_sh(args.subvolume, 10.0, thres)       # LightingSVSH::estimate + computeVoxelShCoeffs (intrinsic3d.cpp:255-264)
    torch.cuda.synchronize(); sh_first_s = time.time() - t0
    t0 = time.time(); ctx.estimate_sh(args.subvolume, 10.0, thres); torch.cuda.synchronize(); sh_estimate_ms = (time.time() - t0) * 1e3      # once more, warm (buffers allocated): what a level pays
    log(f"SH estimate: {sh_sub.shape[0]} subvolumes, {sh_stats.data_rows} data rows, {sh_stats.lm_iterations} LM iterations in {sh_first_s:.2f}s (warm: {sh_estimate_ms:.1f} ms)")

    if args.pmc_calibrate:
        # a KERNEL that reads 2^30 B and writes 2^30 B, three launches (tools/pmc_traffic.py identifies it by name and launch count).  Not a.clone(): a contiguous clone is a
        # device-to-device memcpy of the runtime — no kernel, no counters (round 6: the tool found no copy kernel; rounds 3-4 had in fact calibrated on normal_()'s writes)
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_(); b = torch.empty_like(a)
        for _ in range(3):
            torch.mul(a, 1.0, out=b)          # vectorized_elementwise_kernel: 16 B/lane streaming read + write
        torch.cuda.synchronize(); del a, b
    if args.spin_up > 0:
        # optional (default off): seconds of device copies before the driver's own warm-up steps, untimed, reported as `spin_up_s`.  Tried against the
        # suspicion that the first bench run on a fresh box is slow because the device idled during scene generation: no effect (26.59 with, 26.47
        # without; the SAME box then ran the same command at 25.09 — the 5-9 % spread of the memory-bound kernels moves in time on one box).
        t_spin = time.time(); a = torch.empty(1 << 26, dtype=torch.float32, device="cuda").normal_()
        while time.time() - t_spin < args.spin_up:
            for _ in range(20):
                b = a.clone()
            torch.cuda.synchronize()
        del a, b
    if args.warmup > 0:
        ctx.optimize(make_cfg(binding, args, args.warmup, thres))
    ctx.timing_enable(not args.no_kernel_timing)
    sharded_run = world > 1 or args.force_collectives
    if not args.all_kernel_timing:
        # the roofline kernels only: an event pair around EVERY launch costs ~8 % of the wall clock (+ the exchange launches when sharded)
        ctx.timing_select(["eg_pass", "eg_mr2", "eg_mr3", "build"] + (["comm"] if sharded_run else []))
    ctx.timing_get(reset=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    syncs0 = ctx.debug_counters()["stream_syncs"]; ladder0 = ctx.debug_ladder_stats()
    t0 = time.perf_counter()
    stats = []
    while len(stats) < args.steps:          # the reference's call shape: 10 iterations per Optimizer::optimize, lambda schedule per call
        stats += ctx.optimize(make_cfg(binding, args, min(CALL_ITERATIONS, args.steps - len(stats)), thres))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    stream_syncs = ctx.debug_counters()["stream_syncs"] - syncs0
    ladder = ctx.debug_ladder_stats()
    ladder_d = {k: ladder[k] - ladder0[k] for k in ("batches", "row_streams", "system_passes", "resyncs", "unused_systems")}
    cull_pairs, cull_skipped = ctx.debug_cull_stats()       # observation pass: (64-voxel group, keyframe) pairs of the last iteration, and how many were culled
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    t0 = time.time(); _ = ctx.get_grid(); _ = ctx.get_camera(); t_download = time.time() - t0          # what a host-buffer caller reads back
    timing_work = ctx.timing_get_work_ex()   # launches that did work (PCG launches queued behind the convergence flag return at once); + what the upper cut-off removed
    timing = ctx.timing_get(reset=True)
    sizes = ctx.problem_sizes()
    ctx.timing_enable(False)

    comm_stats = ctx.comm_stats() if sharded_run else None
    transport = ctx.comm_transport() if sharded_run else ""

    A, Rg, Rr, Rs, Ra = sizes["active"], sizes["eg"], sizes["er"], sizes["es"], sizes["ea"]
    kernels = kernel_table(args, sizes, world, timing_work, timing)
    def roof(name):
        return roofline_of(name, kernels, Rg, A, world)
    dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"]) if kernels else None
    roofline = roof(dominant) if dominant else None
    roofline_build = roof("build")           # the kernel the north star names, whichever one dominates
    # all operator launches of the timed region together (one, two or three systems per stream of the rows): strict bytes (4 nnz per launch) and the bytes the serial
    # loop streams for the same system passes, over the summed launch time
    ops = [(kernels[k], n) for k, n in (("eg_pass", 1), ("eg_mr2", 2), ("eg_mr3", 3)) if k in kernels]
    roofline_operator = None
    if ops:
        t_ms = sum(k["avg_ms"] * k["launches"] for k, _ in ops); gb = sum(k["algorithmic_GB"] * k["launches"] for k, _ in ops); ugb = sum(k["algorithmic_GB"] * k["launches"] * n for k, n in ops)
        roofline_operator = {"kernels": "k_eg_tile + k_eg_tile_mr<2> + k_eg_tile_mr<3>", "launches": sum(k["launches"] for k, _ in ops), "ms_total": t_ms, "achieved": gb / (t_ms * 1e-3), "frac": gb / (t_ms * 1e-3) / HBM_PEAK_GBS,
                             "serial_equivalent_GBs": ugb / (t_ms * 1e-3), "serial_equivalent_bandwidth_ratio": ugb / (t_ms * 1e-3) / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "systems_per_launch": ugb / gb}
    comm = None
    if sharded_run:
        # HIP events around the exchange launches that still are launches of their own (the rim push; with RCCL also the all-reduces).  Over the
        # peer-to-peer transport the two reductions of a pass run INSIDE the PCG's boundary kernels, so the whole price of the sharded path on one
        # GPU is the difference of ms_per_step between a plain run and a --force-collectives run (profiles/README.md quotes both).
        ms, n = timing["comm"]; passes = max(1, sum(timing[k][1] for k in ("eg_pass", "eg_mr2", "eg_mr3") if k in timing))      # streams of the rows (a ladder batch shares the exchanges of a pass)
        comm = {"transport": transport, "separate_launch_ms_total": ms, "separate_launches": n, "operator_passes": passes,
                "separate_launch_us_per_pass": 1e3 * ms / passes, "stats_rank0": comm_stats}

    band2 = None
    if rank == 0 and world == 1 and args.band2_steps > 0 and abs(args.band - 2.0) > 1e-6 and not (args.force_collectives or args.pmc_calibrate or args.no_kernel_timing or args.all_kernel_timing or args.carry_radius):
        try:
            band2 = band2_leg(args, binding, log, local_rank)
        except Exception as e:      # reported beside the headline; never lets it down
            log(f"band-2 leg failed: {e}")
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        try:
            cpu = cpu_baseline(args, sc, thres, log, local_rank)
        except Exception as e:      # the baseline is informative; never let it take the measurement down
            log(f"cpu baseline failed: {e}")

    if rank == 0:
        pcg = [int(s.pcg_iterations[i]) for s in stats for i in range(s.num_attempts)]
        out = {
            "metric": "Gauss-Newton iterations/s at the finest SDF level", "value": args.steps / dt, "unit": "GN iterations/s",
            # what the timed steps were (the schedule's iterations differ in weight: later ones take more PCG passes — the reason a 20-step run behind 5 warm-up steps
            # reads lower than a 10-step run): LM attempts and PCG iterations of every attempt, per timed step
            "lm_attempts": [int(s.num_attempts) for s in stats], "pcg_iterations": [[int(s.pcg_iterations[i]) for i in range(s.num_attempts)] for s in stats],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spin_up_s": args.spin_up, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"parallelism": f"{world} rank(s), one per GPU: replicated voxel state; tile-aligned ownership of the brick-ordered work list, rim rows recomputed as ghosts; "
                                       f"per PCG pass three launches and two exchanges that run inside them over the peer-to-peer mailboxes ([4 slice sums + the rim of z] in k_pcg_dir3, "
                                       f"[p.q | camera block] in k_pcg_step3); over RCCL (the default) the damping ladder with ONE neighbour exchange + two all-reduce launches per pass of a BATCH "
                                       f"(k_eg_tile_mr over own + ghost tiles)"
                                       + (f"; this run: {transport}" if transport else ""),
                       "workload": f"synthetic hashed SDF grid, {arrays['keys'].shape[0]} stored voxels @ {args.voxel_size * 1e3:g} mm "
                                   f"({A} in the thin shell), {args.frames} keyframes {args.width}x{args.height}, {sh_sub.shape[0]} SH subvolumes of {args.subvolume} m (estimated on the device, untimed), "
                                   f"joint SDF+albedo+pose+intrinsics+distortion, 5 observations/voxel (BASELINE.json configs[3] on one node)",
                       "stored_voxels": int(arrays["keys"].shape[0]), "active_voxels": A, "active_fraction": A / float(arrays["keys"].shape[0]),
                       "stored_shell_half_thickness_voxels": args.band,
                       "note": "SURVEY section 8(d) prices its worked C4 example with every stored voxel active (N_a = 8e6, R_g = 4e7); here the stored band is "
                               f"{2 * args.band:g} voxels thick and the finest-level thin shell (factor {args.shell:g}) activates {100.0 * A / float(arrays['keys'].shape[0]):.0f} % of it, so one iteration "
                               f"carries {Rg / 4.0e7:.2f}x the rows of that example (--band 2 stores the 4-voxel shell of section 8(d): profiles/r03_bench_band2.json)",
                       "rows": {"Eg": Rg, "Er": Rr, "Es": Rs, "Ea": Ra},
                       "free_parameters": sizes["free"], "keyframes": args.frames, "image": [args.width, args.height],
                       "pcg_iterations_per_step": pcg, "lm_attempts": [int(s.num_attempts) for s in stats]},
            "carry_trust_radius": bool(args.carry_radius), "kernel_tag": KERNEL_TAG,
            "optimize_calls": (args.steps + CALL_ITERATIONS - 1) // CALL_ITERATIONS, "iterations_per_call": min(CALL_ITERATIONS, args.steps),
            "roofline": roofline, "roofline_build": roofline_build, "roofline_operator": roofline_operator, "kernels": kernels, "comm": comm,
            # SURVEY.md 8(d)'s own C4 shape (4-voxel stored shell, ~3.2 Eg rows per voxel) beside the headline workload (7-voxel band, 5.0 rows per voxel)
            "value_band2": band2["value"] if band2 else None, "roofline_band2": band2["roofline"] if band2 else None, "band2": band2,
            "kernel_ms_total": {k: v[0] for k, v in timing.items()}, "kernel_launches": {k: v[1] for k, v in timing.items()},
            "time_split_ms_per_step": {"time_add": float(np.mean([s.time_add for s in stats]) * 1e3), "time_build": float(np.mean([s.time_build for s in stats]) * 1e3),
                                       "time_solve": float(np.mean([s.time_solve for s in stats]) * 1e3)},      # nls_solver.cpp:66-67,101
            "cost": [float(stats[0].cost_initial), float(stats[-1].cost_final)],
            # host <-> device round trips: the trust-region loop runs on the device (lm_kernels.hip), the host polls mapped memory instead of draining the stream
            "stream_syncs_per_step": stream_syncs / float(args.steps),
            # LightingSVSH::estimate + computeVoxelShCoeffs on the device (keys, sorts, MFMA Gram blocks, the 9 S-unknown LM on the host, interpolation): once per level, untimed above
            "sh_estimate_ms": sh_estimate_ms,
            # the damping ladder (whole context, warm-up included): LM attempts solved together share the streams of the stored rows
            "ladder": dict(ladder_d, depth=ladder["depth"]),
            # streams of the stored rows per Gauss-Newton iteration (operator launches: every one reads 4 nnz bytes once) against the system passes they serve
            # (= what the serial trust-region loop streams: one per PCG iteration of every LM attempt)
            "operator_passes_per_step": (sum(kernels[k]["launches"] for k in ("eg_pass", "eg_mr2", "eg_mr3") if k in kernels)) / float(args.steps),
            "system_passes_per_step": (ladder_d["system_passes"] / float(args.steps)) if ladder["depth"] > 1 else (kernels["eg_pass"]["launches"] / float(args.steps) if "eg_pass" in kernels else None),
            "operator_bytes_per_pass_GB": kernels[dominant]["algorithmic_GB"] if dominant in ("eg_pass", "eg_mr2", "eg_mr3") else None,
            "observe_culling": {"group_keyframe_pairs": cull_pairs, "culled": cull_skipped, "fraction": (cull_skipped / float(cull_pairs)) if cull_skipped >= 0 and cull_pairs else None},
            # the boundary also accepts host buffers (i3d_set_grid / i3d_set_frames / i3d_optimize_host): the same run with the one-off upload
            # (voxels + keyframe pyramids over PCIe, hash / neighbour-table build) and the read-back of the refined fields counted in.  Never `value`.
            "host_buffers_inclusive": {"upload_and_grid_build_s": t_upload, "download_s": t_download,
                                       "iterations_per_s": args.steps / (dt + t_upload + t_download)},
            "cpu_baseline": cpu,
        }
        _emit(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
