#!/bin/bash
# round 5, session b: k_eg_tile_mr v2 (batched LDS reads, DPP wave sums) — ladder tests, A/B bench runs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ladder.py -q -m gpu -p no:cacheprovider > $O/ladder_tests.log 2>&1
echo "ladder tests rc=$?" | tee -a $O/summary.txt
tail -25 $O/ladder_tests.log
B="python bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.log; echo "$name rc=$? $(python - <<P
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    k=d['kernels']; print(round(d['value'],2), round(d['ms_per_step'],2), {n:(round(v['avg_ms'],4), v['launches']) for n,v in k.items()}, {a:round(b,2) for a,b in d['time_split_ms_per_step'].items()}, d['config']['lm_attempts'], d.get('ladder'))
except Exception as e: print('no json', e)
P
)" | tee -a $O/summary.txt; }
run ladder6 I3D_LADDER=6
run serial I3D_LADDER=1
run serial_mr1 I3D_LADDER=1 I3D_EGT_MR1=1
run ladder6_g2 I3D_LADDER=6 I3D_LADDER_GROUP=2
run ladder6_mr1 I3D_LADDER=6 I3D_LADDER_MR1=1
