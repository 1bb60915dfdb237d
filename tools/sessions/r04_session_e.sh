#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -q -k "c5" -s > $out/c5.log 2>&1; tail -12 $out/c5.log | cut -c1-600; grep -n "C5\]" $out/c5.log | cut -c1-700
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0 --steps 6 --warmup 1"
i=0
for m in 0 2 3 7 10 11 0 2 3; do i=$((i+1)); I3D_EGT_DET=$m $B > $out/detm_${m}_$i.json 2> /dev/null; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/detm_*.json")):
    d = json.load(open(f)); k = d["kernels"]
    print(os.path.basename(f), "it/s %.2f ms %.3f eg %.4f (%.3f)" % (d["value"], d["ms_per_step"], k["eg_pass"]["avg_ms"], k["eg_pass"]["achieved_GBs"] / 8000.0))
PY
I3D_EGT_DET=3 timeout 200 python tools/flake_hunt.py 3 2>&1 | tail -2 | cut -c1-300
