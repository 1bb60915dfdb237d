#!/bin/bash
# round 5, session q: quad wave sums in the two-pass gradient / column-norm kernels (the sharded path, the debug entry points) — the tests that run them; then the bench as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi_device.py tests/test_gpu_edge_cases.py tests/test_gpu_bench_parity.py -q -m gpu -p no:cacheprovider -s > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED\|normal equations" $O/tests.log | cut -c1-300 | tail -8
I3D_GRADCOL=0 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --band2-steps 0 --all-kernel-timing > $O/bench_twopass.json 2> $O/bench_twopass.log
python - <<P | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_twopass.json').read().strip().splitlines()[-1]); print('two-pass', d['value'], {a:round(b/5,3) for a,b in d['kernel_ms_total'].items() if b})
P
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.log
echo "bench rc=$?" | tee -a $O/summary.txt
python - <<P | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_driver_like.json').read().strip().splitlines()[-1]); print('driver-like', d['value'], d['ms_per_step'], d.get('value_band2'), d['roofline']['frac'], d['roofline'].get('useful_frac'), d['kernels'])
P
