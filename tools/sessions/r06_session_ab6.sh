#!/bin/bash
# round 6, session ab6: k_build<true> requesting the observation slot of row k + 1 while row k is evaluated (tree) against the kernel without it (gpurun_ab/lib_build_r5.so), the kernel
# alone on one unchanged state of the bench workload, builds interleaved; then the tests that hold the rows and the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab6; mkdir -p $O
export TMPDIR=/tmp
P="python tools/experiments/build_value_probe.py"
for rep in 1 2 3; do
  I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_build_r5.so timeout 300 $P > $O/old_$rep.json 2> $O/old_$rep.err
  timeout 300 $P > $O/new_$rep.json 2> $O/new_$rep.err
done
for f in $O/old_*.json $O/new_*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['build_ms'],4), d['eg_rows'])")"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log | cut -c1-200
python bench.py --cpu-sample 0 --band2-steps 0 > $O/bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('it/s', round(d['value'],2), {n:(round(v['avg_ms'],4)) for n,v in d['kernels'].items()})"
