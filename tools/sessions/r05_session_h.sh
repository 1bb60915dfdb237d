#!/bin/bash
# round 5, session h: the profile set (tools/r05_profile.sh) and the full-size parity artefact
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/r05_profile.sh r05h
timeout 1500 python tools/c4_full_parity.py --out gpurun_out/r05h/c4_full_parity.json > /dev/null 2> gpurun_out/r05h/c4_full_parity.log
tail -3 gpurun_out/r05h/c4_full_parity.log | cut -c1-1200
