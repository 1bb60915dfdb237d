#!/bin/bash
# round 6, session v: the sharded ladder tests incl. the fallback without the multi-system pass
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06v; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ladder.py -x -q -m gpu -p no:cacheprovider -k sharded > $O/tests.log 2>&1; echo "rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log | cut -c1-400
