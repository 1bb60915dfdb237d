#!/bin/bash
# round 6, session ab1: how evenly the workgroups of k_eg_tile_mr finish (two s_memtime reads per workgroup, variant build gpurun_ab/lib_blocktime.so), default workload and --band 2
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab1; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --cpu-sample 0 --band2-steps 0"
$B > $O/tree.json 2> /dev/null
I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_blocktime.so $B > $O/bt.json 2> $O/bt.err
I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_blocktime.so $B --band 2 > $O/bt_band2.json 2> $O/bt_band2.err
grep "mr blocktime" $O/bt.err | tail -3 > $O/blocktime_default.txt
grep "mr blocktime" $O/bt_band2.err | tail -3 > $O/blocktime_band2.txt
cat $O/blocktime_default.txt; echo; cat $O/blocktime_band2.txt
python - <<PY
import json
for f in ("tree", "bt", "bt_band2"):
    d = json.loads(open("$O/" + f + ".json").read().strip().splitlines()[-1])
    print(f, "it/s %.2f" % d["value"], {n: (round(v["avg_ms"], 4), v["launches"]) for n, v in d["kernels"].items()})
PY
