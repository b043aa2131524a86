cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/icache
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/ic -- python3 $R/bench.py --steps 6 --warmup 3 --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 --no-dive-leg 0 > $O/ic.log 2>&1 || echo "ic failed"
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/if -- python3 $R/bench.py --steps 6 --warmup 3 --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 --no-dive-leg 0 > $O/if.log 2>&1 || echo "if failed"
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/icache'
for tag in ('ic','if'):
    acc=collections.defaultdict(list)
    for f in glob.glob(O+f'/{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'lp_dual_simplex<7, 5, 16, 128, true' in r['Kernel_Name'] and int(r['Grid_Size'])==4194304:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(tag,k,sum(v)/len(v),len(v))
PY
