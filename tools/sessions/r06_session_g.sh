#!/bin/bash
# round 6, session g: why the sharded ladder and the sharded serial loop ended 5e-4 apart at a fixed PCG depth (session f) — current kernels and the round-5 multi-system kernel (lib_base has no GHOSTS variant: single-rank only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/experiments/sharded_ladder_diag.py 3 > $O/diag.txt 2>&1; tail -20 $O/diag.txt | cut -c1-700
