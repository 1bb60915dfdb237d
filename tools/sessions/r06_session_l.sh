#!/bin/bash
# round 6, session l: the profile set of the round's kernels (tools/r06_profile.sh -> tools/r06_collect.py): default + driver-protocol bench lines, kernel trace, PMC traffic with the
# corrected calibration (default + band 2), SQ counters, MFMA counters of k_sh_gram, serial-loop and LDS-atomic comparisons, run-to-run spread
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r06_profile.sh r06l > gpurun_out/r06l_profile.log 2>&1
tail -40 gpurun_out/r06l_profile.log | cut -c1-700
cat gpurun_out/r06l/pmc_traffic_default.err gpurun_out/r06l/pmc_traffic_band2.err 2>/dev/null | tail -5
