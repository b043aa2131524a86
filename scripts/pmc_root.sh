#!/bin/bash
# PMC passes for the single-workgroup root solve (run on the GPU box through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_root/$tag -- python3 $R/scripts/root_only.py > $R/gpurun_out/pmc_root_$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
tot=collections.defaultdict(list)
for f in glob.glob(R+'/gpurun_out/pmc_root/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'lp_dual_simplex' in r['Kernel_Name']:
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(tot.items()): print(k, [int(x) for x in v])
PY
