#!/bin/bash
# round 5, session n: why test_sharded_ranks_match_single_rank moved (tests/diag_sharded_mismatch.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $O
timeout 600 python tests/diag_sharded_mismatch.py > $O/diag.log 2>&1
echo "rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/diag.log | cut -c1-900
