#!/bin/bash
# round 6, session s: full-size parity of the round's final library on the headline workload (8.02 M voxels, two Gauss-Newton iterations, device vs oracle; the CPU leg is the unextrapolated baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06s; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/c4_full_parity.py --out $O/c4_full_parity.json > /dev/null 2> $O/c4_full_parity.log
tail -3 $O/c4_full_parity.log | cut -c1-1500
