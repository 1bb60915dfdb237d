#!/bin/bash
# round 6, session y: k_pcg_step3 with per-segment loads (86 registers, 5 waves per SIMD: gpurun_ab/lib_split.so) against the batched loads (114 registers, 4 waves per SIMD: tree)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06y; mkdir -p $O
export TMPDIR=/tmp
I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_split.so timeout 600 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py -x -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "split lib: ladder + bench parity rc=$?" | tee -a $O/summary.txt
tail -2 $O/tests.log | cut -c1-300
B="python bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing"
for round in 1 2; do
  for v in tree split; do
    if [ $v = tree ]; then unset I3D_LIB; else export I3D_LIB=$GRAFT_REPO_ROOT/gpurun_ab/lib_$v.so; fi
    $B > $O/${v}_$round.json 2> /dev/null
    $B --band 2 > $O/${v}_band2_$round.json 2> /dev/null
  done
done
unset I3D_LIB
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[12].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), "vector ms %.2f / %d" % (d["kernel_ms_total"]["vector"], d["kernel_launches"]["vector"]))
PY
