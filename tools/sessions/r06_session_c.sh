#!/bin/bash
# round 6, session c: the sharded ladder with the batched rim push (tests), then a rank's share at 8 ranks through the sharded path (1-rank RCCL): ladder vs serial loop vs the plain single-rank path
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_levels.py -x -q -m gpu -p no:cacheprovider -k "sharded or rccl or mailbox or rank" > $O/sharded.log 2>&1; echo "sharded tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/sharded.log | cut -c1-400
B="python bench.py --cpu-sample 0 --voxels 1e6 --band2-steps 0"
$B > $O/share_plain.json 2> /dev/null
$B --force-collectives > $O/share_fc_ladder.json 2> $O/share_fc_ladder.err
I3D_LADDER=1 $B --force-collectives > $O/share_fc_serial.json 2> /dev/null
$B --force-collectives --all-kernel-timing > $O/share_fc_ladder_allk.json 2> /dev/null
python - <<PY
import json
for f in ("share_plain", "share_fc_ladder", "share_fc_serial", "share_fc_ladder_allk"):
    try: d = json.loads(open("$O/" + f + ".json").read().strip().splitlines()[-1])
    except Exception as e: print(f, "MISSING", e); continue
    print(f, "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), {n: (round(v["avg_ms"], 4), v["launches"]) for n, v in d["kernels"].items()}, d.get("ladder"), (d.get("comm") or {}).get("separate_launch_us_per_pass"), (d.get("comm") or {}).get("transport"))
    if f.endswith("allk"): print({k: (round(v, 2), d["kernel_launches"][k]) for k, v in d["kernel_ms_total"].items()})
PY
