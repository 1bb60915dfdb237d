#!/bin/bash
# round 5, session r: the entry point's smoke test and the whole GPU suite on the round's final library
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05r; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log | cut -c1-300
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED" $O/tests.log | cut -c1-300 | tail -5
