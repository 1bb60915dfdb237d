#!/bin/bash
# round 5, session j: candidate cost through shared image samples — parity tests, A/B bench, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_parity.py tests/test_gpu_bench_parity.py -q -m gpu -p no:cacheprovider -s > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED\|\[distortion\]\|perturbation of\|candidate cost" $O/tests.log | cut -c1-400 | tail -12
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.log; echo "$name rc=$? $(python - <<P
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    k=d['kernels']; print(round(d['value'],2), round(d['ms_per_step'],2), {n:(round(v['avg_ms'],4), v['launches']) for n,v in k.items()}, {a:round(b,2) for a,b in d['time_split_ms_per_step'].items()}, {a:round(b/10,2) for a,b in d['kernel_ms_total'].items() if b})
except Exception as e: print('no json', e)
P
)" | tee -a $O/summary.txt; grep -h "candidate cost" $O/bench_$name.log | head -2; }
run shared_all timeout 600 $B --all-kernel-timing
run rowwise_all I3D_COST_SHARED=0 timeout 600 $B --all-kernel-timing
run shared timeout 600 $B
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench_profiled.json 2> $O/bench_profiled.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_avg.py $(find $O/kt -name '*kernel_trace.csv' | head -1) 'i3d::' > $O/kernel_avg_work_only.txt
rm -rf $O/kt
head -24 $O/kernel_avg_work_only.txt
