#!/bin/bash
# round 6, session k: the whole GPU suite as the driver runs it (after the sharded ladder, the prefetching multi-system pass, the one-pass stop lag, the slab changes)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 > $O/tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -14 $O/tests.log | cut -c1-300
