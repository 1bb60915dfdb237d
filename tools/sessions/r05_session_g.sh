#!/bin/bash
# round 5, session g: checkpoint — the default bench command as the driver runs it, the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
python - <<P | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'band2', d.get('value_band2'), 'roofline', {k:d['roofline'].get(k) for k in ('kernel','frac','useful_frac','bound')}, 'operator', d.get('roofline_operator'))
print('cpu', {k:(d.get('cpu_baseline') or {}).get(k) for k in ('value','extrapolated','seconds_per_iteration_sample')}, 'parity', (d.get('cpu_baseline') or {}).get('parity_on_sample'))
print('ladder', d.get('ladder'), 'passes', d.get('operator_passes_per_step'), d.get('system_passes_per_step'), 'sh ms', d.get('sh_estimate_ms'))
P
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed" $O/gpu_suite.log | tail -3
