#!/bin/bash
# Dev tool (GPU box): kernel-trace averages of the bench for a list of env-var variants.  usage: tools/prof_variants.sh <outdir> "<VAR=val ...>" ...
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  d=$out/v$i
  ( cd /tmp && env $v rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$d -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-kernel-timing > $GRAFT_REPO_ROOT/$d.json 2> $GRAFT_REPO_ROOT/$d.log )
  f=$(find $d -name '*kernel_trace.csv' | head -1)
  echo "=== variant $i: $v  ($(python -c "import json;print(json.load(open('$d.json'))['ms_per_step'])" 2>/dev/null) ms/step)"
  python tools/kernel_trace_avg.py $f 'i3d::' | head -14
done
