#!/bin/bash
# round 6, session ab5: k_build<true> with the taps of point j + 2 requested behind the evaluation of point j (within one row) against the batched form (I3D_BUILD_NOPIPE=1), the kernel alone
# on one unchanged state of the bench workload (tools/experiments/build_value_probe.py), then the parity tests that hold the rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab5; mkdir -p $O
export TMPDIR=/tmp
P="python tools/experiments/build_value_probe.py"
for rep in 1 2 3; do
  I3D_BUILD_NOPIPE=1 timeout 300 $P > $O/batched_$rep.json 2> $O/batched_$rep.err
  timeout 300 $P > $O/pipelined_$rep.json 2> $O/pipelined_$rep.err
done
for f in $O/batched_*.json $O/pipelined_*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['build_ms'],4), d['eg_rows'])")"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log | cut -c1-200
