#!/bin/bash
# round 6, session ab8: k_eg_gradcol and k_gather with their regulariser / stencil gathers requested in batches and unconditionally (tree) against the previous commit's kernels
# (gpurun_ab/lib_head.so): the default bench command with every kernel category timed, builds interleaved, then the tests that hold gradient, column norms and the trust-region sequence
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab8; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing"
for rep in 1 2; do
  I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so $B > $O/old_$rep.json 2> /dev/null
  $B > $O/new_$rep.json 2> /dev/null
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); kt = d["kernel_ms_total"]; kl = d["kernel_launches"]
    print("%-8s it/s %.2f  " % (os.path.basename(f)[:-5], d["value"]) + "  ".join("%s %.4f x %d" % (k, kt[k] / max(1, kl[k]), kl[k]) for k in ("eg_aux", "gather", "cost", "build", "eg_mr2")) + "  cost_final %s attempts %s" % (d["cost"], d["lm_attempts"]))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_ladder.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
