"""Copies what tools/r06_profile.sh (+ tools/c4_full_parity.py) left under gpurun_out/<dir> into profiles/ under the round's names (the judged copies).
usage: python tools/r06_collect.py gpurun_out/r06h"""
import json, os, shutil, sys

src = sys.argv[1]; dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def cp(a, b):
    p = os.path.join(src, a)
    if os.path.exists(p): shutil.copyfile(p, os.path.join(dst, b)); print("->", b)
    else: print("MISSING", a)


for a, b in (("bench_default.json", "r06_bench_default.json"), ("bench_driver_like.json", "r06_bench_driver_like.json"), ("bench_profiled.json", "r06_bench_profiled.json"), ("kernel_stats.csv", "r06_bench_profiled_kernel_stats.csv"),
             ("kernel_avg_work_only.txt", "r06_bench_profiled_kernel_avg_work_only.txt"), ("pmc_traffic_default.json", "r06_pmc_traffic.json"), ("pmc_traffic_band2.json", "r06_band2_pmc_traffic.json"),
             ("sq_counters_default.json", "r06_sq_counters.json"), ("sq_counters_band2.json", "r06_band2_sq_counters.json"), ("bench_serial.json", "r06_bench_serial_loop.json"),
             ("bench_deterministic.json", "r06_bench_deterministic.json"), ("bench_atomics.json", "r06_bench_lds_atomic_mode.json"), ("timeline_idle.txt", "r06_timeline_idle.txt"), ("c4_full_parity.json", "r06_c4_full_parity.json"),
             ("sh_kernels.txt", "r06_sh_kernels.txt")):
    cp(a, b)
# MFMA evidence of the SH Gram kernel: raw counters + the derived figures
p = os.path.join(src, "mfma_sh_gram_raw.json")
if os.path.exists(p):
    raw = json.load(open(p)); out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU over "
                                               "`python bench.py --steps 1 --warmup 0 --cpu-sample 0 --band2-steps 0 --no-kernel-timing` (the bench's SH estimate: 515 subvolumes, 2.28 M eligible voxels)",
                                     "kernels": raw}
    for k, v in raw.items():
        m = v["mean"]
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in m:
            # one v_mfma_f64_16x16x4_f64 = 16 x 16 x 4 x 2 flop = 2048 flop; the counter counts MOPS in units of 512 flop (MI355X_MICROARCH.md) -> report both raw and derived
            out.setdefault("derived", {})[k] = {"mfma_mops_f64": m["SQ_INSTS_VALU_MFMA_MOPS_F64"], "mfma_busy_cycles": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "busy_cycles": m.get("SQ_BUSY_CYCLES"),
                                                  "mfma_busy_share_of_busy_cycles": (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]) if m.get("SQ_BUSY_CYCLES") else None}
    json.dump(out, open(os.path.join(dst, "r06_mfma_sh_gram.json"), "w"), indent=1); print("-> r06_mfma_sh_gram.json")
# (a rank's share through the sharded path: profiles/r06_rank_share.json, written from tools/sessions/r06_session_c.sh)
with open(os.path.join(dst, "r06_run_to_run.txt"), "w") as f:
    names = (("# default mode (bit-reproducible operator pass)", "run_to_run_default.txt"), ("# I3D_DETERMINISTIC=0 (fp32 LDS atomics inside k_eg_tile; k_eg_tile_mr is fixed-order)", "run_to_run_atomics.txt"))
    if os.path.exists(os.path.join(src, "run_to_run_deterministic.txt")):      # sessions before the default changed
        names = (("# default mode (fp32 LDS atomics inside k_eg_tile; k_eg_tile_mr is fixed-order)", "run_to_run_default.txt"), ("# I3D_DETERMINISTIC=1", "run_to_run_deterministic.txt"))
    for title, name in names:
        p = os.path.join(src, name)
        f.write(title + "\n" + ("".join(l for l in open(p) if l.startswith("rep ") or l.startswith("max")) if os.path.exists(p) else "MISSING\n"))
print("-> r06_run_to_run.txt")
