#!/bin/bash
# round 6, session d: where a tile of k_eg_tile_mr spends its time (s_memtime phase marks, variant build gpurun_ab/lib_phases.so), default workload and --band 2
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --cpu-sample 0 --band2-steps 0"
$B > $O/tree.json 2> /dev/null
I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_phases.so $B > $O/phases.json 2> $O/phases.err
I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_phases.so $B --band 2 > $O/phases_band2.json 2> $O/phases_band2.err
grep "mr phases" $O/phases.err | tail -42 > $O/phases_default.txt
grep "mr phases" $O/phases_band2.err | tail -42 > $O/phases_band2.txt
cat $O/phases_default.txt; echo; cat $O/phases_band2.txt
python - <<PY
import json
for f in ("tree", "phases", "phases_band2"):
    d = json.loads(open("$O/" + f + ".json").read().strip().splitlines()[-1])
    print(f, "it/s %.2f" % d["value"], {n: (round(v["avg_ms"], 4), v["launches"]) for n, v in d["kernels"].items()})
PY
