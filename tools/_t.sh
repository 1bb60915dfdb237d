cd /root/repo
timeout 600 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_levels.py tests/test_gpu_loader.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/fusion_prof2 -o fusion -- python /root/repo/tools/fusion_bench.py --frames 20 > /root/repo/gpurun_out/fusion_prof2.log 2>&1
grep -v "^W2026\|^E2026" /root/repo/gpurun_out/fusion_prof2.log | tail -2
find /root/repo/gpurun_out/fusion_prof2 -name "*kernel_stats.csv" | head -1 | xargs cut -c1-150 | head -8
