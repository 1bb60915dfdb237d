#!/bin/bash
# round 6, session ab3: the launch sequence of one outer iteration (kernel trace of the default bench command, fills / copies / library kernels included)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab3; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0 --no-kernel-timing > $O/bench.json 2> $O/bench.log
cd $GRAFT_REPO_ROOT
KT=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python tools/experiments/iteration_sequence.py $KT 6 > $O/iteration_sequence.txt 2>&1
rm -rf $O/kt
cat $O/iteration_sequence.txt | cut -c1-3000
