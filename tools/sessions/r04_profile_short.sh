#!/bin/bash
# GPU box: the round-4 profile set of the final kernels.  usage: tools/r04_profile.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo skipping suite
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
$B > $out/bench_default.json 2> $out/bench_default.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B --cpu-sample 0 --band2-steps 0 > $out/bench_profiled.json 2> $out/bench_profiled.log
S="--steps 2 --warmup 1 --cpu-sample 0 --band2-steps 0 --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B $S --pmc-calibrate > $out/bench_pmc.json 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B $S --pmc-calibrate > /dev/null 2> $out/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq1 -- $B $S > /dev/null 2> $out/sq1.log
$B --band 2 --cpu-sample 0 --steps 6 > $out/bench_band2.json 2> $out/bench_band2.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_avg.py $(find $out/kt -name '*kernel_trace.csv' | head -1) 'i3d::' > $out/kernel_avg_work_only.txt
python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/bench_pmc.json $out/pmc_traffic.json > /dev/null 2> $out/pmc_traffic.err
python tools/pmc_summary.py $out/sq1 'k_build|k_eg_tile|k_observe|k_pcg_step|k_pcg_dir' --json $out/sq1.json > $out/sq1.txt 2>&1
python tools/sq_valu.py $out/sq1.json $out/bench_pmc.json $out/sq_counters.json > $out/sq_valu.txt 2>&1
cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
python tools/timeline_idle.py $(find $out/kt -name '*kernel_trace.csv' | head -1) > $out/timeline_idle.txt 2>&1; cat $out/timeline_idle.txt | head -12
rm -rf $out/kt/*/*kernel_trace.csv $out/pmc_fetch $out/pmc_write $out/sq1
head -14 $out/kernel_avg_work_only.txt; head -12 $out/sq_valu.txt; tail -2 $out/run_to_run_default.txt | cut -c1-300; tail -1 $out/run_to_run_deterministic.txt | cut -c1-300
python - <<PY
import json
for f in ("bench_default", "bench_profiled", "bench_band2", "share_plain", "share_fc", "share_fc_rccl", "bench_deterministic"):
    try: d = json.load(open("$out/" + f + ".json"))
    except Exception as e: print(f, "MISSING", e); continue
    k = d["kernels"]
    print(f, "it/s %.2f ms %.3f syncs %.1f eg %.4f (%.3f) build %.4f (%.3f)" % (d["value"], d["ms_per_step"], d.get("stream_syncs_per_step", -1), k["eg_pass"]["avg_ms"], k["eg_pass"]["achieved_GBs"] / 8000.0, k["build"]["avg_ms"], d["roofline_build"]["frac"]),
          (d.get("comm") or {}).get("transport"), "band2:", d.get("value_band2"), (d.get("roofline_band2") or {}).get("frac"), "cpu:", (d.get("cpu_baseline") or {}).get("value"))
t = json.load(open("$out/pmc_traffic.json")); print({k: (round(v["traffic_bytes_per_launch"] / 1e9, 3), round((v.get("algorithmic_bytes_per_launch") or 0) / 1e9, 3)) for k, v in t["kernels"].items()})
PY
