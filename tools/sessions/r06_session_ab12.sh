#!/bin/bash
# round 6, session ab12: the plumbing of an iteration — halo-pair sort on 22 / 23 key bits instead of 32, the one-workgroup partial reductions with batched loads, cost / step-norm slots
# assigned instead of zeroed + added (two fills per trust-region attempt gone) — tree against the previous commit (gpurun_ab/lib_head.so): default bench command, builds interleaved,
# three repetitions, the launch census of one iteration, then the full ladder / parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab12; mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0"
for rep in 1 2 3; do
  I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so $B > $O/old_$rep.json 2> /dev/null
  $B > $O/new_$rep.json 2> /dev/null
done
cd /tmp
for v in old new; do
  L=""; [ $v = old ] && L=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so
  I3D_LIB=$L rocprofv3 --kernel-trace --output-format csv -d $O/kt_$v -- $B --no-kernel-timing > /dev/null 2> $O/kt_$v.log
  python $GRAFT_REPO_ROOT/tools/experiments/iteration_sequence.py $(find $O/kt_$v -name '*kernel_trace.csv' | head -1) 6 > $O/sequence_$v.txt 2>&1
  rm -rf $O/kt_$v
done
cd $GRAFT_REPO_ROOT
for v in old new; do echo "--- $v"; head -1 $O/sequence_$v.txt; grep -E "radix|fillBuffer|copyBuffer|k_reduce_partials|^busy" $O/sequence_$v.txt | cut -c1-110; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-8s it/s %.2f ms %.3f add %.3f build %.3f solve %.3f cost_final %s attempts %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["time_split_ms_per_step"]["time_add"], d["time_split_ms_per_step"]["time_build"], d["time_split_ms_per_step"]["time_solve"], d["cost"], d["lm_attempts"]))
PY
timeout 1500 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
