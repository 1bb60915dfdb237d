#!/bin/bash
# round 4, session j: the new border-row test, then the sharded (rank-simulation) tests repeated — looking for the one-off mismatch of session r4b
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "border" > $out/border.log 2>&1; tail -5 $out/border.log | cut -c1-600
fails=0
for i in $(seq 1 $2); do
  timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -q -x -k "sharded or rank or transport or halo" > $out/sharded_$i.log 2>&1 || { fails=$((fails+1)); echo "run $i failed"; tail -30 $out/sharded_$i.log | cut -c1-400; }
done
echo "sharded repeats: $2 runs, $fails failed"; tail -3 $out/sharded_1.log
