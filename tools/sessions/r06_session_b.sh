#!/bin/bash
# round 6, session b: first run of the ladder in the sharded path (rank simulation, 1-rank RCCL) + the sharded tests that now go through it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ladder.py -x -q -m gpu -p no:cacheprovider -k "sharded" > $O/ladder_sharded.log 2>&1; echo "sharded ladder rc=$?" | tee -a $O/summary.txt
tail -30 $O/ladder_sharded.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_levels.py -x -q -m gpu -p no:cacheprovider -k "sharded or rccl or mailbox or rank" > $O/sharded_others.log 2>&1; echo "other sharded rc=$?" | tee -a $O/summary.txt
tail -15 $O/sharded_others.log | cut -c1-600
