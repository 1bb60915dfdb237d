#!/bin/bash
# round 6, session o: k_eg_gradcol with the next row slot in flight: 2 waves per SIMD without spills (tree) / 3 waves per SIMD with 128 B of scratch (lib_gc3) / the kernel as it was (lib_prev)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06o; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "normal_equations or optimize_matches or golden" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/tests.log
B="python bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing"
for round in 1 2; do
  for v in prev gc3 tree; do
    if [ $v = tree ]; then unset I3D_LIB; else export I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_$v.so; fi
    $B > $O/${v}_$round.json 2> /dev/null
  done
done
unset I3D_LIB
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[12].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), "eg_aux ms %.3f in %d launches" % (d["kernel_ms_total"]["eg_aux"], d["kernel_launches"]["eg_aux"]))
PY
