#!/bin/bash
# Per-kernel times of a C4 run (cut rounds in the engine, 4096 nodes per step) for one or more builds of
# libmipx.so on ONE box: rocprofv3 kernel trace of scripts/c4_tree.py, grouped by kernel and grid.
# usage (through gpurun): bash scripts/c4_kernels.sh libmipx.so [libmipx_B.so ...]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export MIPX_LIB=$R/simple_mip_solver_amd/csrc/$L
  rm -rf /tmp/tr_$L
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$L -- python3 scripts/c4_tree.py 256 128 4096 10 > /tmp/c4_$L.log 2>&1)
  echo "== $L: $(grep '^C4' /tmp/c4_$L.log)"
  python3 $R/scripts/summarize_profile.py $(ls /tmp/tr_$L/*/*kernel_trace.csv) | awk -F, '$2==1048576 || $2==2097152 {printf "   %-60s calls %s avg %s us\n", substr($1,1,60), $7, $8}'
done
