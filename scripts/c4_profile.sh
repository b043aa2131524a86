R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trc4
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trc4 -- python3 scripts/c4_tree.py 256 128 4096 10 > $R/gpurun_out/c4_run.log 2>&1)
python3 $R/scripts/summarize_profile.py $(ls /tmp/trc4/*/*kernel_trace.csv) > $R/gpurun_out/r02_c4_kernel_by_grid.csv
