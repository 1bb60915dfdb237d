#!/bin/bash
# round 5, session i: early row reload in k_eg_tile_mr + the wave-skew experiment; the tightened parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -s > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED\|\[distortion\]\|\[normal equations\]" $O/tests.log | cut -c1-600 | tail -12
B="python bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.log; echo "$name rc=$? $(python - <<P
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    k=d['kernels']; print(round(d['value'],2), round(d['ms_per_step'],2), {n:(round(v['avg_ms'],4), v['launches']) for n,v in k.items()}, {a:round(b,2) for a,b in d['time_split_ms_per_step'].items()})
except Exception as e: print('no json', e)
P
)" | tee -a $O/summary.txt; }
run skew0 timeout 600 $B
run skew1 I3D_MR_SKEW=1 timeout 600 $B
run skew2 I3D_MR_SKEW=2 timeout 600 $B
run skew4 I3D_MR_SKEW=4 timeout 600 $B
run skew8 I3D_MR_SKEW=8 timeout 600 $B
run skew0b timeout 600 $B
