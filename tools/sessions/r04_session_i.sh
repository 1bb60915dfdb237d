#!/bin/bash
# round 4, session i: software-pipelined candidate cost + branch-free taps — parity subset, A/B, then the whole GPU suite
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_edge_cases.py -q -x > $out/parity.log 2>&1; tail -6 $out/parity.log | cut -c1-400
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing"
$B > $out/allk_pipe.json 2> /dev/null
I3D_COST_NOPIPE=1 $B > $out/allk_nopipe.json 2> /dev/null
$B > $out/allk_pipe2.json 2> /dev/null
python - <<PY
import json
for f in ("allk_pipe", "allk_nopipe", "allk_pipe2"):
    d = json.load(open("$out/" + f + ".json")); L = d["kernel_launches"]
    print(f, "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), {k: (round(v / d["steps"], 3), round(v / max(L[k], 1), 4)) for k, v in d["kernel_ms_total"].items() if v}, d["cost"])
PY
python -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -8 $out/gputest.log | cut -c1-400
