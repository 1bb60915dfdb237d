#!/bin/bash
# round 6, session x: soak — the GPU suite twice more on one box (flake estimate: sessions a, k, p, r, w were green)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06x; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/tests_$i.log 2>&1; echo "run $i rc=$?" | tee -a $O/summary.txt
  grep -n "passed\|failed" $O/tests_$i.log | tail -1
done
