#!/bin/bash
# round 6, session a: the driver's two commands on the re-ordered suite (durations recorded), smoke, and the driver-protocol bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=25 -s > $O/tests.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED\|\[C5" $O/tests.log | cut -c1-1200 | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 1500 $O/bench.json
