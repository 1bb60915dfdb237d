#!/bin/bash
# round 5, session a: the damping ladder — bit-identity tests, then A/B bench runs (serial loop / ladder / groupings), then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ladder.py -q -m gpu -p no:cacheprovider > gpurun_out/r05a/ladder_tests.log 2>&1
echo "ladder tests rc=$?" | tee -a gpurun_out/r05a/summary.txt
tail -40 gpurun_out/r05a/ladder_tests.log
B="python bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
run() { name=$1; shift; env "$@" timeout 600 $B > gpurun_out/r05a/bench_$name.json 2> gpurun_out/r05a/bench_$name.log; echo "$name rc=$? $(python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r05a/bench_$name.json').read().strip().splitlines()[-1])
    k=d['kernels']; print(d['value'], d['ms_per_step'], {n:(round(v['avg_ms'],4), v['launches']) for n,v in k.items()}, d['time_split_ms_per_step'], d['config']['lm_attempts'], d.get('ladder'))
except Exception as e: print('no json', e)
P
)" | tee -a gpurun_out/r05a/summary.txt; }
run serial I3D_LADDER=1
run ladder6 I3D_LADDER=6
run ladder6_g2 I3D_LADDER=6 I3D_LADDER_GROUP=2
run ladder6_mr1 I3D_LADDER=6 I3D_LADDER_MR1=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r05a/gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a gpurun_out/r05a/summary.txt
tail -15 gpurun_out/r05a/gpu_suite.log
