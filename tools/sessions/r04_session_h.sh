#!/bin/bash
# round 4, session h: where the observation pass's time goes with culling (kernel trace) and how much is culled
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_cull.py -q -x > $out/cull.log 2>&1; tail -5 $out/cull.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0 --steps 4 --warmup 1 --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o cull -- $B > $out/bench_prof.json 2> $out/bench_prof.err
python - <<PY
import json, glob, csv
d = json.load(open("$out/bench_prof.json")); print(d["value"], d["observe_culling"])
for f in glob.glob("$out/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("observe", "cull", "depth_blocks", "group_bounds", "k_build", "classify")): print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
for m in 0 1; do I3D_NO_CULL=$m python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing > $out/allk_$m.json 2> /dev/null; done
python - <<PY
import json
for m in range(2):
    d = json.load(open("$out/allk_%d.json" % m)); print(m, "it/s %.2f" % d["value"], {k: round(v / d["steps"], 3) for k, v in d["kernel_ms_total"].items() if k in ("observe", "classify", "build")}, d["config"]["rows"], d["cost"])
PY
