#!/bin/bash
# Kernel-trace + HBM PMC passes for the bench workload (run on the GPU box through gpurun).
# Outputs land in gpurun_out/prof/; copy the summaries into profiles/ afterwards.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
ARGS="--steps ${STEPS:-10} --warmup 3 --cpu-seconds 0 ${EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $ARGS > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections, json
R = os.environ['GRAFT_REPO_ROOT']; O = R + '/gpurun_out/prof'
kt = glob.glob(O + '/trace/*/*kernel_trace.csv')[0]
os.system(f'python3 {R}/scripts/summarize_profile.py {kt} > {O}/kernel_by_grid.csv')
st = glob.glob(O + '/trace/*/*kernel_stats.csv')
if st: os.system(f'cp {st[0]} {O}/kernel_stats.csv')
def per_launch(tag, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(O + f'/{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'lp_dual_simplex' in r['Kernel_Name']:
                vals[int(r['Grid_Size'])].append(float(r['Counter_Value']))
    return vals
fe, wr = per_launch('fetch', 'FETCH_SIZE'), per_launch('write', 'WRITE_SIZE')
g = max(fe, key=lambda k: (len(fe[k]) > 3, k))      # the steady-state frontier batch
f_kb = sum(fe[g]) / len(fe[g]); w_kb = sum(wr[g]) / len(wr[g])
out = {"command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py " + os.environ.get('ARGS', ''),
       "kernel": f"lp_dual_simplex (K1), grid {g} threads per launch (one frontier batch)",
       "fetch_size_KB_raw": f_kb, "write_size_KB": w_kb,
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)",
       "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024}
json.dump(out, open(O + '/pmc_latest.json', 'w'), indent=1)
print(json.dumps(out))
print(open(O + '/kernel_by_grid.csv').read())
PY
tail -1 $O/trace.log
