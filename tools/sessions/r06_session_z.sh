#!/bin/bash
# round 6, session z: the device hash table under crafted collisions (2000 voxels on one home slot); 8 observation slots
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06z; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_edge_cases.py -x -q -m gpu -p no:cacheprovider -k "collisions or parameter_group" > $O/tests.log 2>&1; echo "rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log | cut -c1-400
