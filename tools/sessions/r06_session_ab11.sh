#!/bin/bash
# round 6, session ab11: k_candidate with a thread's entries four at a time (tree) against the previous commit's kernel (gpurun_ab/lib_head.so): kernel trace of two short runs (the
# kernel's own average), default bench command interleaved, then the tests that hold the trust-region sequence
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab11; mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0"
cd /tmp
for v in old new; do
  L=""; [ $v = old ] && L=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so
  I3D_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -- $B --steps 4 --warmup 1 --no-kernel-timing > /dev/null 2> $O/kt_$v.log
  python $GRAFT_REPO_ROOT/tools/kernel_trace_avg.py $(find $O/kt_$v -name '*kernel_trace.csv' | head -1) 'k_candidate|k_accept|k_gather|k_eg_gradcol|k_tile_plan|k_eaw_sym' > $O/avg_$v.txt 2>&1
  rm -rf $O/kt_$v
done
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so $B > $O/old_$rep.json 2> /dev/null
  $B > $O/new_$rep.json 2> /dev/null
done
echo "--- old"; cat $O/avg_old.txt | head -8; echo "--- new"; cat $O/avg_new.txt | head -8
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-8s it/s %.2f ms %.3f cost_final %s attempts %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["cost"], d["lm_attempts"]))
PY
timeout 1500 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
