#!/bin/bash
# round 5, session t: k_eg_tile_mr<1>, <2> hold the 14 voxel inputs of an entry in registers across its rows — ladder + parity tests, then a same-session A/B against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05t; mkdir -p $O
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -n "passed\|failed\|FAILED" $O/tests.log | cut -c1-300 | tail -4
AB_ARGS="--steps 10 --warmup 2 --band2-steps 0" timeout 260 bash tools/ab_libs.sh r05t/ab gpurun_ab/lib_base.so - 2>&1 | tail -6 | tee -a $O/summary.txt
