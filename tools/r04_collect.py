"""Copies what tools/r04_profile.sh left under gpurun_out/<dir> into profiles/ under the round's names (the judged copies).  usage: python tools/r04_collect.py gpurun_out/r4p"""
import json, os, shutil, sys

src = sys.argv[1]; dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
def cp(a, b):
    p = os.path.join(src, a)
    if os.path.exists(p): shutil.copyfile(p, os.path.join(dst, b)); print("->", b)
    else: print("MISSING", a)
for a, b in (("bench_default.json", "r04_bench_default.json"), ("bench_profiled.json", "r04_bench_profiled.json"), ("kernel_stats.csv", "r04_bench_profiled_kernel_stats.csv"),
             ("kernel_avg_work_only.txt", "r04_bench_profiled_kernel_avg_work_only.txt"), ("pmc_traffic.json", "r04_pmc_traffic.json"), ("sq_counters.json", "r04_sq_counters.json"),
             ("bench_band2.json", "r04_bench_band2.json"), ("bench_deterministic.json", "r04_bench_deterministic.json"), ("timeline_idle.txt", "r04_timeline_idle.txt")):
    cp(a, b)
runs = {}
for name in ("share_plain", "share_fc", "share_fc_rccl"):
    p = os.path.join(src, name + ".json")
    if not os.path.exists(p): continue
    d = json.load(open(p)); c = d.get("comm")
    runs[name] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "eg_ms": d["kernels"]["eg_pass"]["avg_ms"], "build_ms": d["kernels"]["build"]["avg_ms"],
                  "transport": (c or {}).get("transport"), "pcg_iterations": float(sum(d["config"]["pcg_iterations_per_step"])) / d["steps"],
                  "active_voxels": d["config"]["active_voxels"], "rows": d["config"]["rows"], "stream_syncs_per_step": d.get("stream_syncs_per_step"), "comm": c}
if runs:
    json.dump({"what": "a rank's share of the bench problem (8 ranks: 1/8 of the voxels) on ONE GPU: plain single-rank path, the sharded path through a 1-rank communicator over the "
                       "mailbox transport (three launches per pass, exchanges inside the kernels) and over RCCL (six launches + rim push + two all-reduce launches); "
                       "python bench.py --cpu-sample 0 --voxels 1e6 [--force-collectives] (I3D_TRANSPORT=rccl for the last); one session", "runs": runs},
              open(os.path.join(dst, "r04_rank_share.json"), "w"), indent=1)
    print("-> r04_rank_share.json")
with open(os.path.join(dst, "r04_run_to_run.txt"), "w") as f:
    for title, name in (("# default mode (fp32 LDS atomics inside k_eg_tile)", "run_to_run_default.txt"), ("# I3D_DETERMINISTIC=1", "run_to_run_deterministic.txt")):
        p = os.path.join(src, name)
        f.write(title + "\n" + ("".join(l for l in open(p) if l.startswith("rep ") or l.startswith("max")) if os.path.exists(p) else "MISSING\n"))
print("-> r04_run_to_run.txt")
