#!/bin/bash
# round 6, last sessions: the driver's two commands (GPU suite with -x, smoke()) three times in a row on one box, then the bench as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06zz3; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/tests_$rep.log 2>&1; echo "rep $rep: gpu suite rc=$? $(grep -E 'passed|failed' $O/tests_$rep.log | tail -1)" | tee -a $O/summary.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$rep.log 2>&1; echo "rep $rep: smoke rc=$?" | tee -a $O/summary.txt
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('it/s %.2f ms %.3f band2 %.2f' % (d['value'], d['ms_per_step'], d['value_band2']), d['roofline']['kernel'], round(d['roofline']['frac'],3), 'build', round(d['roofline_build']['frac'],3))"
