#!/bin/bash
# GPU box, round 4 session c: full suite with the fixed-order kernels, run-to-run differences (flake_hunt), DET A/B, 512-entry tiles on the 4-voxel shell
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -25 $out/gputest.log | cut -c1-600
timeout 300 python tools/flake_hunt.py 4 > $out/flake_det.txt 2>&1; cat $out/flake_det.txt | cut -c1-400
I3D_EGT_DET=0 timeout 300 python tools/flake_hunt.py 3 > $out/flake_nodet.txt 2>&1; tail -2 $out/flake_nodet.txt | cut -c1-400
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --band2-steps 0"
for r in 1 2; do
  $B > $out/det_$r.json 2> $out/det_$r.err
  I3D_EGT_DET=0 $B > $out/nodet_$r.json 2> $out/nodet_$r.err
done
$B --band 2 --steps 5 > $out/band2_1024.json 2> /dev/null
I3D_EGT_TILE=512 $B --band 2 --steps 5 > $out/band2_512.json 2> /dev/null
I3D_EGT_TILE=512 $B > $out/det_512.json 2> /dev/null
python - <<PY
import json
for f in ("det_1", "nodet_1", "det_2", "nodet_2", "band2_1024", "band2_512", "det_512"):
    try: d = json.load(open("$out/" + f + ".json"))
    except Exception as e: print(f, "MISSING", e); continue
    k = d["kernels"]
    print(f, "it/s %.2f ms %.3f eg %.4f (%.3f) build %.4f split %s" % (d["value"], d["ms_per_step"], k["eg_pass"]["avg_ms"], k["eg_pass"]["achieved_GBs"] / 8000.0, k["build"]["avg_ms"], {a: round(b, 2) for a, b in d["time_split_ms_per_step"].items()}))
PY
