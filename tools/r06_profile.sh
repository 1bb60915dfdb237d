#!/bin/bash
# GPU box: the round-6 profile set of the final kernels.  usage: tools/r06_profile.sh <outdir under gpurun_out>   (then: python tools/r06_collect.py gpurun_out/<outdir>)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
$B > $out/bench_default.json 2> $out/bench_default.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B --cpu-sample 0 --band2-steps 0 > $out/bench_profiled.json 2> $out/bench_profiled.log
S="--steps 2 --warmup 1 --cpu-sample 0 --band2-steps 0 --no-kernel-timing"
for W in default band2; do
  X=""; [ $W = band2 ] && X="--band 2"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$W -- $B $S $X --pmc-calibrate > $out/bench_pmc_$W.json 2> $out/pmc_fetch_$W.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$W -- $B $S $X --pmc-calibrate > /dev/null 2> $out/pmc_write_$W.log
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq_$W -- $B $S $X > /dev/null 2> $out/sq_$W.log
done
# MFMA counters of the SH Gram kernel (k_sh_gram: the one GEMM-shaped piece of the path), on the bench's 515-subvolume estimate
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $out/mfma -- $B --steps 1 --warmup 0 --cpu-sample 0 --band2-steps 0 --no-kernel-timing > /dev/null 2> $out/mfma.log
rocprofv3 --kernel-trace --output-format csv -d $out/kt_sh -- $B --steps 1 --warmup 0 --cpu-sample 0 --band2-steps 0 --no-kernel-timing > /dev/null 2> $out/kt_sh.log
I3D_LADDER=1 $B --cpu-sample 0 --band2-steps 0 > $out/bench_serial.json 2> /dev/null
I3D_DETERMINISTIC=0 $B --cpu-sample 0 --band2-steps 0 > $out/bench_atomics.json 2> /dev/null      # (the LDS-atomic operator pass: the default up to round 4)
$B --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_like.json 2> $out/bench_driver_like.log      # the driver's command (CPU baseline and band-2 leg in the same process)
cd $GRAFT_REPO_ROOT
I3D_DETERMINISTIC=0 timeout 200 python tools/flake_hunt.py 3 > $out/run_to_run_atomics.txt 2>&1
timeout 200 python tools/flake_hunt.py 3 > $out/run_to_run_default.txt 2>&1
KT=$(find $out/kt -name '*kernel_trace.csv' | head -1)
python tools/kernel_trace_avg.py $KT 'i3d::' > $out/kernel_avg_work_only.txt
python tools/timeline_idle.py $KT > $out/timeline_idle.txt 2>&1
cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
python tools/kernel_trace_avg.py $(find $out/kt_sh -name '*kernel_trace.csv' | head -1) 'k_sh_' > $out/sh_kernels.txt 2>&1
for W in default band2; do
  python tools/pmc_traffic.py $out/pmc_fetch_$W $out/pmc_write_$W $out/bench_pmc_$W.json $out/pmc_traffic_$W.json > /dev/null 2> $out/pmc_traffic_$W.err
  python tools/pmc_summary.py $out/sq_$W 'k_build|k_eg_tile|k_eg_gradcol|k_observe|k_pcg_step|k_pcg_dir' --json $out/sq_$W.json > $out/sq_$W.txt 2>&1
  python tools/sq_valu.py $out/sq_$W.json $out/bench_pmc_$W.json $out/sq_counters_$W.json > $out/sq_valu_$W.txt 2>&1
done
python tools/pmc_summary.py $out/mfma 'k_sh_gram' --json $out/mfma_sh_gram_raw.json > $out/mfma_sh_gram.txt 2>&1
rm -rf $out/kt $out/kt_sh $out/pmc_fetch_* $out/pmc_write_* $out/sq_default $out/sq_band2 $out/mfma
head -16 $out/kernel_avg_work_only.txt; cat $out/timeline_idle.txt | head -3; cat $out/mfma_sh_gram.txt; cat $out/sh_kernels.txt | head; tail -2 $out/run_to_run_default.txt | cut -c1-300; tail -1 $out/run_to_run_atomics.txt | cut -c1-300
python - <<PY
import json
for f in ("bench_default", "bench_driver_like", "bench_profiled", "bench_serial", "bench_atomics"):
    try: d = json.loads(open("$out/" + f + ".json").read().strip().splitlines()[-1])
    except Exception as e: print(f, "MISSING", e); continue
    k = d["kernels"]
    print(f, "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), {n: (round(v["avg_ms"], 4), v["launches"]) for n, v in k.items()}, "band2:", d.get("value_band2"), (d.get("roofline_band2") or {}).get("frac"), "cpu:", (d.get("cpu_baseline") or {}).get("value"))
for W in ("default", "band2"):
    try: t = json.load(open("$out/pmc_traffic_%s.json" % W)); print(W, {k: (round(v["traffic_bytes_per_launch"] / 1e9, 3), round((v.get("algorithmic_bytes_per_launch") or 0) / 1e9, 3)) for k, v in t["kernels"].items()})
    except Exception as e: print(W, "traffic MISSING", e)
PY
