#!/bin/bash
# A/B of two builds of libmipx.so on ONE box (boxes differ by a few per cent): the bench's K1 launch time,
# alternating.  usage (through gpurun): ROUNDS=3 bash scripts/ab.sh libA.so libB.so ...
R=$GRAFT_REPO_ROOT
N=${ROUNDS:-2}
ARGS="--cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0"
for i in $(seq 1 $N); do
  for L in "$@"; do
    MIPX_LIB=$R/simple_mip_solver_amd/csrc/$L timeout -k 10 150 python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', 'launch_ms %.4f' % d['roofline']['launch_ms'], 'ms_per_step %.4f' % d['ms_per_step'], 'value %.0f' % d['value'], flush=True)"
  done
done
