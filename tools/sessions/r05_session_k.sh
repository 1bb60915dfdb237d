#!/bin/bash
# round 5, session k: the whole GPU suite on the round's final library + the bench line as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log
echo "bench rc=$?" | tee -a $O/summary.txt
python - <<P | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('value_band2'), d['roofline'], d.get('cpu_baseline'))
P
