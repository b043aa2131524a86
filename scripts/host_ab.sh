#!/bin/bash
# host-side A/B of two builds on one box: the engine's own phase timers over the bench's 200 steps
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do for L in "$@"; do
MIPX_LIB=$R/simple_mip_solver_amd/csrc/$L MIPX_TREE_PROFILE=1 timeout -k 10 150 python3 $R/bench.py --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 2>&1 >/dev/null | grep "200 steps" | sed "s/^/$L /" | cut -c1-200
done; done
