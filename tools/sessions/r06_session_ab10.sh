#!/bin/bash
# round 6, session ab10: k_eg_tile_mr with the camera part of the operator inputs staged by a non-inlined helper in batches of four loads (tree) against the previous commit's kernel
# (gpurun_ab/lib_head.so): default bench command and --band 2, builds interleaved, three repetitions; then the ladder tests (bit-identity of a system across launch widths)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab10; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --cpu-sample 0 --band2-steps 0"
for rep in 1 2 3; do
  I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so $B > $O/old_$rep.json 2> /dev/null
  $B > $O/new_$rep.json 2> /dev/null
done
I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so $B --band 2 > $O/old_band2.json 2> /dev/null
$B --band 2 > $O/new_band2.json 2> /dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels"]
    print("%-10s it/s %.2f  " % (os.path.basename(f)[:-5], d["value"]) + "  ".join("%s %.4f x %d" % (n, v["avg_ms"], v["launches"]) for n, v in k.items()) + "  cost_final %s" % (d["cost"],))
PY
timeout 1500 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
