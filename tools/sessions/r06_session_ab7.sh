#!/bin/bash
# round 6, session ab7: the candidate-cost kernel k_build<false> (and k_build<true>) with their gathers requested in batches (tree) against round 5's build.hip (gpurun_ab/lib_build_r5.so):
# the default bench command with every kernel category timed, builds interleaved, then the tests that hold rows, costs and the trust-region sequence
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab7; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing"
for rep in 1 2; do
  I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_build_r5.so $B > $O/old_$rep.json 2> /dev/null
  $B > $O/new_$rep.json 2> /dev/null
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); kt = d["kernel_ms_total"]; kl = d["kernel_launches"]
    print("%-8s it/s %.2f  cost %.4f ms x %d  build %.4f  mr2 %.4f  cost_final %s attempts %s" % (os.path.basename(f)[:-5], d["value"], kt["cost"] / kl["cost"], kl["cost"], kt["build"] / kl["build"], kt["eg_mr2"] / kl["eg_mr2"], d["cost"], d["lm_attempts"]))
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_ladder.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
