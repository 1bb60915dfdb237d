#!/bin/bash
# round 6, session t: the new origin-straddling tests (fusion: truncating round; optimizer: keys and SH subvolumes of both signs) and the PCG-stop test in the LDS-atomic mode
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_edge_cases.py tests/test_gpu_bench_parity.py -x -q -m gpu -p no:cacheprovider -k "origin or negative or native_pcg" > $O/tests.log 2>&1; echo "rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log | cut -c1-400
