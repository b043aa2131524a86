R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in libmipx_A.so libmipx_B.so; do
rm -rf /tmp/gt_$L
(cd $R && MIPX_LIB=$R/simple_mip_solver_amd/csrc/$L rocprofv3 --kernel-trace --output-format csv -d /tmp/gt_$L -- python3 bench.py --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 > /dev/null 2>&1)
python3 - <<PY
import csv, glob
f=sorted(glob.glob('/tmp/gt_$L/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
k1=[r for r in rows if 'lp_dual_simplex' in r['Kernel_Name'] and int(r['Grid_Size_X'])==4194304]
gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(k1[5:-1],k1[6:])]
gaps.sort()
print('$L', len(k1), 'K1 launches; gap us: mean %.0f median %.0f p90 %.0f max %.0f; K1 mean %.0f us' % (sum(gaps)/len(gaps), gaps[len(gaps)//2], gaps[int(len(gaps)*0.9)], gaps[-1], sum((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in k1)/len(k1)))
PY
done
