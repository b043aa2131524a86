#!/bin/bash
# A/B of the device finish against the host loop (MIPX_HOST_FINISH=1) on the bench workload, with the engine's
# host-side phase profile, then a kernel trace of the device-finish run.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fin
rm -rf $O && mkdir -p $O
ARGS="--steps ${STEPS:-100} --tto-seconds 0 --others 0 --highs-seconds 0 --cpu-seconds 0 --no-dive-leg 0"
MIPX_TREE_PROFILE=1 python3 $R/bench.py $ARGS > $O/dev.json 2> $O/dev.err
MIPX_TREE_PROFILE=1 MIPX_HOST_FINISH=1 python3 $R/bench.py $ARGS > $O/host.json 2> $O/host.err
python3 - <<PY
import json
for f in ('dev','host'):
    j=json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1])
    print(f,'value %.3e ms/step %.3f launch_ms %.3f'%(j['value'],j['ms_per_step'],j['roofline']['launch_ms']))
PY
grep "call:" $O/dev.err | tail -2
grep "call:" $O/host.err | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 30 --tto-seconds 0 --others 0 --highs-seconds 0 --cpu-seconds 0 --no-dive-leg 0 > $O/trace.log 2>&1
python3 $R/scripts/summarize_profile.py $(ls $O/trace/*/*kernel_trace.csv | tail -1) > $O/by_grid.csv
head -30 $O/by_grid.csv
