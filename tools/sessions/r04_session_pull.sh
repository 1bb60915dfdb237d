#!/bin/bash
# round 4: halo sums pulled over plan lists instead of pushed with LDS atomics — correctness in both modes, then the four variants of the operator pass
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
I3D_HALO_PULL=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py -q -x -k "optimize_matches or normal_equations or multi_tile or sharded or falls_back or native_pcg" 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-300
python -m pytest tests/test_gpu_bench_parity.py -q -x -k "deterministic" 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-300
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0"
$B > $out/v0.json 2> /dev/null
I3D_HALO_PULL=1 $B > $out/v4.json 2> /dev/null
I3D_DETERMINISTIC=1 $B > $out/v6.json 2> /dev/null
I3D_DETERMINISTIC=1 I3D_EGT_ORDERED=1 $B > $out/v3.json 2> /dev/null
$B > $out/v0b.json 2> /dev/null
I3D_HALO_PULL=1 $B --band 2 --steps 6 > $out/v4_band2.json 2> /dev/null
$B --band 2 --steps 6 > $out/v0_band2.json 2> /dev/null
python - <<PY
import json
for f in ("v0", "v4", "v6", "v3", "v0b", "v4_band2", "v0_band2"):
    try: d = json.load(open("$out/" + f + ".json"))
    except Exception as e: print(f, "MISSING", e); continue
    k = d["kernels"]["eg_pass"]
    print(f, "it/s %.2f ms %.3f eg %.4f (%.3f)" % (d["value"], d["ms_per_step"], k["avg_ms"], d["roofline"]["frac"]), d["cost"], sum(d["config"]["pcg_iterations_per_step"]), d["time_split_ms_per_step"])
PY
