#!/bin/bash
# round 5, session u: the remaining GPU-minutes — the tests that run the refine schedule and the bench slice on the final library (ladder + parity ran in session t)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05u; mkdir -p $O
export TMPDIR=/tmp
timeout 175 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py -q -m gpu -p no:cacheprovider -x > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED" $O/tests.log | cut -c1-300 | tail -4
