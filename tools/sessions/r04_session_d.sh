#!/bin/bash
# GPU box, round 4 session d: the sharded parity test three times (diagnostics), C5, and the fixed-order variants of k_eg_tile against each other
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests/test_gpu_parity.py -q -k "sharded_ranks_match" > $out/sharded_$i.log 2>&1; tail -3 $out/sharded_$i.log | cut -c1-300; grep -n "AssertionError\|^E  " $out/sharded_$i.log | head -8 | cut -c1-700; done
python -m pytest tests/test_gpu_configs.py -q -k "c5" -s > $out/c5.log 2>&1; tail -8 $out/c5.log | cut -c1-500; grep -n "C5\]" $out/c5.log | cut -c1-600
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0 --steps 6 --warmup 1"
for m in 0 2 3 7 0 3; do I3D_EGT_DET=$m $B > $out/detm_${m}_$RANDOM.json 2> /dev/null; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/detm_*.json")):
    d = json.load(open(f)); k = d["kernels"]
    print(os.path.basename(f), "it/s %.2f ms %.3f eg %.4f (%.3f)" % (d["value"], d["ms_per_step"], k["eg_pass"]["avg_ms"], k["eg_pass"]["achieved_GBs"] / 8000.0))
PY
