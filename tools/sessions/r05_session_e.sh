#!/bin/bash
# round 5, session e: wave_sum_quads (permlane swaps) in k_eg_tile_mr — micro-test, ladder tests, bench A/B, parity-critical tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -I intrinsic3d_amd/csrc/device tools/experiments/wave_reduce_test.hip -o /tmp/wave_reduce_test > /dev/null 2>&1 && /tmp/wave_reduce_test | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py -q -m gpu -p no:cacheprovider > $O/ladder_tests.log 2>&1
echo "ladder + bench parity tests rc=$?" | tee -a $O/summary.txt
tail -6 $O/ladder_tests.log | cut -c1-400
B="python bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.log; echo "$name rc=$? $(python - <<P
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    k=d['kernels']; print(round(d['value'],2), round(d['ms_per_step'],2), {n:(round(v['avg_ms'],4), v['launches']) for n,v in k.items()}, {a:round(b,2) for a,b in d['time_split_ms_per_step'].items()}, d['config']['lm_attempts'], d.get('ladder'))
except Exception as e: print('no json', e)
P
)" | tee -a $O/summary.txt; }
run ladder6 I3D_LADDER=6 timeout 600 $B
run ladder6_mr1 I3D_LADDER=6 I3D_LADDER_MR1=1 timeout 600 $B
run serial_mr1 I3D_LADDER=1 I3D_EGT_MR1=1 timeout 600 $B
run band2 I3D_LADDER=6 timeout 600 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --band2-steps 0 --band 2
run band2_serial I3D_LADDER=1 timeout 600 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --band2-steps 0 --band 2
