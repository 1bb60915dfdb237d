#!/bin/bash
# round 6, session u: halo sums stored in pair-rank order (the fold of k_pcg_step3 reads an entry's pairs as one contiguous run) against the build before it (gpurun_ab/lib_prev.so):
# bit-identity + parity tests, then an interleaved A/B with every kernel category timed
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ladder.py tests/test_gpu_bench_parity.py tests/test_gpu_edge_cases.py -x -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log | cut -c1-300
B="python bench.py --cpu-sample 0 --band2-steps 0 --all-kernel-timing"
for round in 1 2; do
  for v in prev tree; do
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
    print(os.path.basename(f), "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), "vector ms %.2f / %d" % (d["kernel_ms_total"]["vector"], d["kernel_launches"]["vector"]), "classify ms %.2f" % d["kernel_ms_total"]["classify"], {n: (round(v["avg_ms"], 4), v["launches"]) for n, v in d["kernels"].items() if n != "eg_pass"})
PY
