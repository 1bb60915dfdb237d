#!/bin/bash
# round 6, session m: the PMC traffic passes again (the calibration copy is now a kernel: torch.mul, three launches), default workload and band 2
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/r06m; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
S="--steps 2 --warmup 1 --cpu-sample 0 --band2-steps 0 --no-kernel-timing"
for W in default band2; do
  X=""; [ $W = band2 ] && X="--band 2"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$W -- $B $S $X --pmc-calibrate > $out/bench_pmc_$W.json 2> $out/pmc_fetch_$W.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$W -- $B $S $X --pmc-calibrate > /dev/null 2> $out/pmc_write_$W.log
done
cd $GRAFT_REPO_ROOT
for W in default band2; do
  python tools/pmc_traffic.py $out/pmc_fetch_$W $out/pmc_write_$W $out/bench_pmc_$W.json $out/pmc_traffic_$W.json > /dev/null 2> $out/pmc_traffic_$W.err; cat $out/pmc_traffic_$W.err | cut -c1-600
done
rm -rf $out/pmc_fetch_* $out/pmc_write_*
python - <<PY
import json
for W in ("default", "band2"):
    try:
        t = json.load(open("$out/pmc_traffic_%s.json" % W)); print(W, t["calibration"]); print({k: (round(v["read_bytes_per_launch"] / 1e9, 3), round(v["write_bytes_per_launch"] / 1e9, 3), round((v.get("algorithmic_bytes_per_launch") or 0) / 1e9, 3)) for k, v in t["kernels"].items()})
    except Exception as e: print(W, "traffic MISSING", e)
PY
