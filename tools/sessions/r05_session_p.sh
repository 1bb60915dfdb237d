#!/bin/bash
# round 5, session p: the profile set of the round's FINAL kernels (tools/r05_profile.sh; k_eg_gradcol, bit-reproducible pass by default)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/r05_profile.sh r05p
