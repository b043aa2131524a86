#!/bin/bash
# SQ issue counters of the C4 kernels (K2 gomory_cuts above all): where the waves' time goes.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/c4pmc
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/a -- python3 scripts/c4_tree.py 256 128 4096 4 > $O/a.log 2>&1 || echo a failed
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/b -- python3 scripts/c4_tree.py 256 128 4096 4 > $O/b.log 2>&1 || echo b failed
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c -- python3 scripts/c4_tree.py 256 128 4096 4 > $O/c.log 2>&1 || echo c failed
python3 - <<PY
import csv, glob, collections
for tag in 'abc':
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$O/%s/*/*counter_collection.csv' % tag):
        for r in csv.DictReader(open(f)):
            if int(r['Grid_Size']) in (1048576, 2097152):
                acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(tag, k, {c: '%.3g' % (sum(v) / len(v)) for c, v in sorted(d.items())})
PY
