cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c5c
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/scripts/c4_tree.py 1024 512 256 3 > $O/log.txt 2>&1
python3 $R/scripts/summarize_profile.py $(ls $O/t/*/*kernel_trace.csv | tail -1) > $O/by_grid.csv
grep -v rocprofv3 $O/log.txt | tail -5
head -25 $O/by_grid.csv
