#!/bin/bash
# round 5, session d: after the scratch fix of the batch vector kernels and the alive flags of k_eg_tile_mr
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ladder.py -q -m gpu -p no:cacheprovider > $O/ladder_tests.log 2>&1
echo "ladder tests rc=$?" | tee -a $O/summary.txt
tail -12 $O/ladder_tests.log | cut -c1-400
B="python bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.log; echo "$name rc=$? $(python - <<P
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    k=d['kernels']; print(round(d['value'],2), round(d['ms_per_step'],2), {n:(round(v['avg_ms'],4), v['launches']) for n,v in k.items()}, {a:round(b,2) for a,b in d['time_split_ms_per_step'].items()}, 'ms_total/step', {a:round(b/10,2) for a,b in d['kernel_ms_total'].items() if b}, 'launches/step', {a:round(b/10,1) for a,b in d['kernel_launches'].items() if b})
except Exception as e: print('no json', e)
P
)" | tee -a $O/summary.txt; }
run ladder6_all I3D_LADDER=6 timeout 600 $B --all-kernel-timing
run serial_all I3D_LADDER=1 timeout 600 $B --all-kernel-timing
run ladder6 I3D_LADDER=6 timeout 600 $B
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x --deselect tests/test_gpu_ladder.py > $O/gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -8 $O/gpu_suite.log | cut -c1-600
