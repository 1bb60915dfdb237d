#!/bin/bash
# round 6, session ab4: what ONE cost launch for the B candidates of a batch could take — variant builds whose cost kernel does B times the work in one launch (gpurun_ab/lib_cprobe<B>.so:
# the B workgroups of a voxel block on one XCD at about the same time; lib_cprobe3far.so: a whole candidate apart), the run itself unchanged
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab4; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing"
for rep in 1 2; do
  $B > $O/tree_$rep.json 2> /dev/null
  for v in cprobe1 cprobe2 cprobe3 cprobe3far cprobe6; do I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_$v.so $B > $O/${v}_$rep.json 2> /dev/null; done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels"]
    print("%-14s it/s %.2f  cost %.4f ms x %d  build %.4f  mr2 %.4f  attempts %s" % (os.path.basename(f)[:-5], d["value"], k["cost"]["avg_ms"], k["cost"]["launches"], k["build"]["avg_ms"], k["eg_mr2"]["avg_ms"], d["lm_attempts"]))
PY
