#!/bin/bash
# round 6, session i: the ladder drops a stopped system one pass earlier (the boundary of the pass just queued instead of the one before): tests, then the driver protocol + the default command
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py tests/test_gpu_edge_cases.py -x -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log | cut -c1-300
python bench.py --cpu-sample 0 > $O/default.json 2> /dev/null
python bench.py --cpu-sample 0 --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2> /dev/null
python - <<PY
import json
for f in ("default", "driver"):
    d = json.loads(open("$O/" + f + ".json").read().strip().splitlines()[-1])
    print(f, "it/s %.2f ms %.3f band2 %s" % (d["value"], d["ms_per_step"], d.get("value_band2")), {n: (round(v["avg_ms"], 4), v["launches"]) for n, v in d["kernels"].items() if n != "eg_pass"},
          "op passes/step", d["operator_passes_per_step"], "system passes/step", d["system_passes_per_step"], "pcg sum/step", sum(sum(x) for x in d["pcg_iterations"]) / d["steps"], d["ladder"])
PY
