#!/bin/bash
# GPU box: same-session A/B of library builds (boxes differ by +-5 %, so variants are only comparable inside one call).
# AB_ARGS: extra bench.py arguments (e.g. --all-kernel-timing for the per-category kernel time)
# usage: tools/ab_libs.sh <outdir under gpurun_out> <lib1.so> <lib2.so> ...   ("-" = the in-tree build); every library is run twice, interleaved
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $out
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so); [ "$lib" = "-" ] && tag=tree
    if [ "$lib" = "-" ]; then unset I3D_LIB; else export I3D_LIB=$GRAFT_REPO_ROOT/$lib; fi
    python bench.py --cpu-sample 0 $AB_ARGS > $out/${tag}_$round.json 2> $out/${tag}_$round.log
  done
done
unset I3D_LIB
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/*.json")):
    d = json.load(open(f)); k = d["kernels"]
    t = d.get("kernel_ms_total", {}); n = d.get("kernel_launches", {})
    print(os.path.basename(f), "it/s %.2f  ms %.2f  eg %.4f  build %.4f  split %s  vector %.2f ms / %d launches" % (d["value"], d["ms_per_step"], k["eg_pass"]["avg_ms"], k["build"]["avg_ms"],
          {a: round(b, 2) for a, b in d["time_split_ms_per_step"].items()}, t.get("vector", 0.0), n.get("vector", 0)))
PY
