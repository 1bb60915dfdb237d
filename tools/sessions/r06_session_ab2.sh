#!/bin/bash
# round 6, session ab2: the value half of a two-kernel split of k_build<true>, measured (variant builds -DI3D_BUILD_VALUE_PROBE=2|3|4) beside the shipped kernel and a streaming copy
# of the bytes the derivative half would move at least (tools/experiments/build_value_probe.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab2; mkdir -p $O
export TMPDIR=/tmp
P="python tools/experiments/build_value_probe.py"
for rep in 1 2; do
  timeout 300 $P > $O/tree_$rep.json 2> $O/tree_$rep.err
  for w in 2 3 4; do I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_vprobe$w.so timeout 300 $P > $O/vprobe${w}_$rep.json 2> $O/vprobe${w}_$rep.err; done
done
tail -n 1 $O/*.json | cut -c1-400
