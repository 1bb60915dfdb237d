#!/bin/bash
# GPU box: the round-2 profile set (kernel trace + stats of the default bench, PMC traffic passes, SQ counter passes).  usage: tools/r02_profile.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
$B > $out/bench_default.json 2> $out/bench_default.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B > $out/bench_profiled.json 2> $out/bench_profiled.log
S="--steps 2 --warmup 1 --cpu-sample 0 --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B $S --pmc-calibrate > $out/bench_pmc.json 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B $S --pmc-calibrate > /dev/null 2> $out/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq1 -- $B $S > /dev/null 2> $out/sq1.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/sq2 -- $B $S > /dev/null 2> $out/sq2.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_avg.py $(find $out/kt -name '*kernel_trace.csv' | head -1) 'i3d::' > $out/kernel_avg_work_only.txt
python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/bench_pmc.json $out/pmc_traffic.json > /dev/null 2> $out/pmc_traffic.err
python tools/pmc_summary.py $out/sq1 'k_build|k_eg_tile|k_observe|k_pcg_step' --json $out/sq1.json > $out/sq1.txt 2>&1
python tools/pmc_summary.py $out/sq2 'k_build|k_eg_tile|k_observe|k_pcg_step' --json $out/sq2.json > $out/sq2.txt 2>&1
head -12 $out/kernel_avg_work_only.txt; cat $out/sq1.txt | head -60; tail -3 $out/sq2.log; python -c "
import json; d=json.load(open('$out/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['roofline_build'])
t=json.load(open('$out/pmc_traffic.json')); print({k:(v['traffic_bytes_per_launch'], v.get('algorithmic_bytes_per_launch')) for k,v in t['kernels'].items()})"
