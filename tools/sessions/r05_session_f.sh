#!/bin/bash
# round 5, session f: kernel trace of the ladder bench (per-kernel averages, device idle)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 --band2-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench_profiled.json 2> $O/bench_profiled.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_avg.py $(find $O/kt -name '*kernel_trace.csv' | head -1) 'i3d::|rocprim' > $O/kernel_avg_work_only.txt
python tools/timeline_idle.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/timeline_idle.txt 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/kt
head -45 $O/kernel_avg_work_only.txt; head -14 $O/timeline_idle.txt
