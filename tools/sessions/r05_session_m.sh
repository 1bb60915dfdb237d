#!/bin/bash
# round 5, session m: k_eg_gradcol with quad wave sums + the initial cost riding on it, bit-reproducible pass by default on one rank — whole GPU suite, A/B bench, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -s > $O/tests.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED\|C5, " $O/tests.log | cut -c1-400 | tail -12
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.log; echo "$name rc=$? $(python - <<P
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    k=d['kernels']; print(round(d['value'],2), round(d['ms_per_step'],2), {n:(round(v['avg_ms'],4), v['launches']) for n,v in k.items()}, {a:round(b,2) for a,b in d['time_split_ms_per_step'].items()}, {a:round(b/10,2) for a,b in d['kernel_ms_total'].items() if b})
except Exception as e: print('no json', e)
P
)" | tee -a $O/summary.txt; }
run new timeout 600 $B
run old I3D_GRADCOL=0 I3D_COST0=0 timeout 600 $B
run new_all timeout 600 $B --all-kernel-timing
run atomics I3D_DETERMINISTIC=0 timeout 600 $B
run new2 timeout 600 $B
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench_profiled.json 2> $O/bench_profiled.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_avg.py $(find $O/kt -name '*kernel_trace.csv' | head -1) 'i3d::' > $O/kernel_avg_work_only.txt
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/kt
head -30 $O/kernel_avg_work_only.txt
