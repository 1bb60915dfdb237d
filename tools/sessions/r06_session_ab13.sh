#!/bin/bash
# round 6, session ab13: TSDF fusion (row f4) with correctSDF's 26 neighbours requested in three batches (tree) against the previous kernel (gpurun_ab/lib_head.so):
# tools/fusion_bench.py, builds interleaved, then the fusion tests (records bit-identical to the oracle's)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab13; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_head.so timeout 600 python tools/fusion_bench.py --frames 30 > $O/old_$rep.json 2> $O/old_$rep.err
  timeout 600 python tools/fusion_bench.py --frames 30 > $O/new_$rep.json 2> $O/new_$rep.err
done
for f in $O/old_*.json $O/new_*.json; do echo "$(basename $f) $(tail -1 $f | cut -c1-300)"; done
timeout 900 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu > $O/tests.log 2>&1; echo "fusion tests rc=$?"; tail -2 $O/tests.log | cut -c1-200
