#!/bin/bash
# GPU box, round 4 session b: full GPU suite, default bench (with the band-2 leg), a rank's share plain vs through the sharded path, partition A/B on the 4-voxel shell,
# kernel trace of the default bench (device idle = wall - sum of kernel time)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -12 $out/gputest.log
B="python $GRAFT_REPO_ROOT/bench.py"
$B --cpu-sample 0 > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err
$B --cpu-sample 0 --voxels 1e6 --band2-steps 0 > $out/share_plain.json 2> $out/share_plain.err
$B --cpu-sample 0 --voxels 1e6 --force-collectives > $out/share_fc.json 2> $out/share_fc.err
I3D_TRANSPORT=rccl $B --cpu-sample 0 --voxels 1e6 --force-collectives > $out/share_fc_rccl.json 2> $out/share_fc_rccl.err
$B --cpu-sample 0 --band 2 --steps 5 > $out/band2.json 2> $out/band2.err
I3D_NO_PARTITION=1 $B --cpu-sample 0 --band 2 --steps 5 > $out/band2_nopart.json 2> $out/band2_nopart.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B --cpu-sample 0 --band2-steps 0 > $out/bench_profiled.json 2> $out/bench_profiled.log
cd $GRAFT_REPO_ROOT
python - <<PY
import json, glob
def L(f):
    try: return json.load(open("$out/" + f + ".json"))
    except Exception as e: return None
for f in ("bench", "share_plain", "share_fc", "share_fc_rccl", "band2", "band2_nopart", "bench_profiled"):
    d = L(f)
    if not d: print(f, "MISSING"); continue
    k = d["kernels"]
    print(f, "it/s %.2f ms %.3f syncs %.1f eg %.4f (%.3f) build %.4f (%.3f) attempts %s" % (d["value"], d["ms_per_step"], d.get("stream_syncs_per_step", -1), k["eg_pass"]["avg_ms"], d["roofline"]["frac"] if d["roofline"]["kernel"] == "k_eg_tile" else -1,
          k["build"]["avg_ms"], d["roofline_build"]["frac"], d["config"]["lm_attempts"][:4]), (d.get("comm") or {}).get("transport"), "band2:", d.get("value_band2"), (d.get("roofline_band2") or {}).get("frac"))
PY
python tools/kernel_trace_avg.py $(find $out/kt -name '*kernel_trace.csv' | head -1) 'i3d::' > $out/kernel_avg_work_only.txt 2>&1; head -30 $out/kernel_avg_work_only.txt
find $out/kt -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/kt -name '*kernel_trace.csv' | head -1 | xargs -I{} python - {} <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
# the timed region: last 10 GN iterations = from the 3rd k_classify-from-the-end...: report the whole trace's busy fraction over the span of i3d kernels after warm-up
i3d = [k for k in ks if "i3d::" in k[2]]
builds = [k for k in i3d if "k_build<true" in k[2]]
t0 = builds[-10][0] if len(builds) >= 10 else i3d[0][0]
sel = [k for k in i3d if k[0] >= t0 - 5_000_000]
span = sel[-1][1] - sel[0][0]; busy = sum(e - s for s, e, _ in sel)
gaps = [sel[i + 1][0] - sel[i][1] for i in range(len(sel) - 1)]
import statistics
print("timeline over the last 10 iterations: span %.2f ms, kernels %.2f ms (%d launches), idle %.2f ms per iteration; median gap %.2f us, gaps > 20 us: %d (%.2f ms)" % (
    span / 1e6, busy / 1e6, len(sel), (span - busy) / 1e7, statistics.median(gaps) / 1e3, sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
PY
