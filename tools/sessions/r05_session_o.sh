#!/bin/bash
# round 5, session o: the whole GPU suite on the final library + the sharded-simulation test ten times (its host-mediated copies are stream-ordered now)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05o; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -s > $O/tests.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED\|C5, " $O/tests.log | cut -c1-400 | tail -12
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "sharded_ranks_match_single_rank" > $O/sharded_$i.log 2>&1; echo "sharded rep $i rc=$?" | tee -a $O/summary.txt; done
