#!/bin/bash
# round 6, session h: the ladder tests after the sharded tests were re-based on identical second-iteration inputs
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06h; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ladder.py -x -q -m gpu -p no:cacheprovider -s > $O/tests.log 2>&1; echo "ladder rc=$?" | tee -a $O/summary.txt
grep -n "sharded ladder,\|passed\|failed" $O/tests.log | cut -c1-400
