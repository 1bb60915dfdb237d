#!/bin/bash
# round 4, session g: observation-pass culling — with / without test, the GPU suite, default bench and the kernel split with and without culling
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_cull.py -q -x > $out/cull.log 2>&1; tail -25 $out/cull.log | cut -c1-400
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0"
$B --all-kernel-timing > $out/bench_allk.json 2> $out/bench_allk.err
I3D_NO_CULL=1 $B --all-kernel-timing > $out/bench_allk_nocull.json 2> /dev/null
$B > $out/bench.json 2> /dev/null
I3D_NO_CULL=1 $B > $out/bench_nocull.json 2> /dev/null
python - <<PY
import json
for f in ("bench", "bench_nocull", "bench_allk", "bench_allk_nocull"):
    d = json.load(open("$out/" + f + ".json"))
    print(f, "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), d["time_split_ms_per_step"], {k: round(v / d["steps"], 3) for k, v in d["kernel_ms_total"].items() if v})
PY
python -m pytest tests -m gpu -q --deselect tests/test_gpu_cull.py > $out/gputest.log 2>&1; tail -8 $out/gputest.log | cut -c1-400
