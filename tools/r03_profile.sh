#!/bin/bash
# GPU box: the round-3 profile set of the default bench (kernel trace + stats, PMC traffic passes, SQ counter pass) and the two extra bench lines the
# round-2 review asked for (--band 2, the 4-voxel stored shell of SURVEY section 8(d); a 2 M-voxel CPU sample for the linearity of cpu_baseline).
# usage: tools/r03_profile.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
$B > $out/bench_default.json 2> $out/bench_default.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B --cpu-sample 0 > $out/bench_profiled.json 2> $out/bench_profiled.log
S="--steps 2 --warmup 1 --cpu-sample 0 --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B $S --pmc-calibrate > $out/bench_pmc.json 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B $S --pmc-calibrate > /dev/null 2> $out/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq1 -- $B $S > /dev/null 2> $out/sq1.log
$B --band 2 --cpu-sample 0 > $out/bench_band2.json 2> $out/bench_band2.log
$B --steps 2 --warmup 1 --cpu-sample 2.0e6 > $out/bench_cpu2m.json 2> $out/bench_cpu2m.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_avg.py $(find $out/kt -name '*kernel_trace.csv' | head -1) 'i3d::' > $out/kernel_avg_work_only.txt
python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/bench_pmc.json $out/pmc_traffic.json > /dev/null 2> $out/pmc_traffic.err
python tools/pmc_summary.py $out/sq1 'k_build|k_eg_tile|k_observe|k_pcg_step|k_pcg_dir' --json $out/sq1.json > $out/sq1.txt 2>&1
python tools/sq_valu.py $out/sq1.json $out/bench_pmc.json $out/sq_counters.json > $out/sq_valu.txt 2>&1
head -14 $out/kernel_avg_work_only.txt; cat $out/sq_valu.txt | head -20; python -c "
import json
for f in ('bench_default','bench_band2','bench_cpu2m'):
    d=json.load(open('$out/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_build']['frac'], d['time_split_ms_per_step'], (d.get('cpu_baseline') or {}).get('seconds_per_iteration_sample'))
t=json.load(open('$out/pmc_traffic.json')); print({k:(v['traffic_bytes_per_launch'], v.get('algorithmic_bytes_per_launch')) for k,v in t['kernels'].items()})"
