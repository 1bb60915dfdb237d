cd /root/repo
timeout 300 python -m pytest tests/test_gpu_fusion.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/fusion_prof5 -o fusion -- python /root/repo/tools/fusion_bench.py --frames 30 > /root/repo/gpurun_out/fusion_prof5.log 2>&1
grep -v "^W2026\|^E2026" /root/repo/gpurun_out/fusion_prof5.log | tail -2
find /root/repo/gpurun_out/fusion_prof5 -name "*kernel_stats.csv" | head -1 | xargs cut -c1-120 | head -12
