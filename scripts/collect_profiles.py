"""Turn the rocprofv3 passes of scripts/profile_bench.sh into the summaries kept under profiles/."""
import collections, csv, glob, json, os, sys
O = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get('PROFILE_TAG', 'r03')


def newest(pattern):
    """gpurun MERGES a call's files into gpurun_out/: passes of earlier calls may still lie beside the
    latest one -- only the newest file of a pass counts."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


def by_grid(trace_dir, out):
    kt = newest(trace_dir + '/*/*kernel_trace.csv')[0]
    os.system(f'python3 {R}/scripts/summarize_profile.py {kt} > {out}')
    st = newest(trace_dir + '/*/*kernel_stats.csv')
    return st[0] if st else None


def counter(tag, name, kernel_substr):
    vals = collections.defaultdict(list)
    for f in newest(O + f'/{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name and kernel_substr in r['Kernel_Name']:
                vals[int(r['Grid_Size'])].append(float(r['Counter_Value']))
    return vals


def steady(vals):
    g = max(vals, key=lambda k: (len(vals[k]) > 3, k))   # the steady-state frontier batch
    return g, sum(vals[g]) / len(vals[g])


args = open(O + '/args.txt').read().strip()
st = by_grid(O + '/trace', O + f'/{TAG}_kernel_by_grid.csv')
if st:
    os.system(f'cp {st} {O}/{TAG}_kernel_stats.csv')
fe, wr = counter('fetch', 'FETCH_SIZE', 'lp_dual_simplex<'), counter('write', 'WRITE_SIZE', 'lp_dual_simplex<')
g, f_kb = steady(fe)
_, w_kb = steady(wr)
issue = {}
for tag in ('sq1', 'sq2'):
    names = set()
    for f in newest(O + f'/{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            names.add(r['Counter_Name'])
    for nm in sorted(names):
        v = counter(tag, nm, 'lp_dual_simplex<')
        if v and g in v:
            issue[nm] = sum(v[g]) / len(v[g])
if issue.get('SQ_WAVE_CYCLES'):
    wc = issue['SQ_WAVE_CYCLES']
    issue['derived'] = {
        'valu_insts_per_wave_cycle': issue.get('SQ_INSTS_VALU', 0) / (4.0 * wc),
        'active_inst_valu_over_wave_cycles': issue.get('SQ_ACTIVE_INST_VALU', 0) / wc,
        'active_inst_any_over_wave_cycles': issue.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        'wait_any_over_wave_cycles': issue.get('SQ_WAIT_ANY', 0) / wc,
        'note': 'per launch of the steady-state node-LP kernel; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count '
                'quad-cycles summed over waves (MI355X_MICROARCH.md), SQ_INSTS_* count wave-instructions'}
sys.path.insert(0, R)
from simple_mip_solver_amd._ffi import source_hash
out = {"csrc_sha256": source_hash(),
       "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ groups (separate passes) -- python3 bench.py " + args,
       "kernel": f"lp_dual_simplex (K1), grid {g} threads per launch (one frontier batch)",
       "fetch_size_KB_raw": f_kb, "write_size_KB": w_kb,
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)",
       "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024, "issue": issue}
json.dump(out, open(O + '/pmc_latest.json', 'w'), indent=1)
print(json.dumps(out)[:1500])
with open(O + f'/{TAG}_pmc_hbm.csv', 'w') as fh:   # the per-dispatch values behind the two averages
    fh.write('pass,kernel,grid,counter,value_KB\n')
    for tag, name in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        rows = []
        for f in newest(O + f'/{tag}/*/*counter_collection.csv'):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == name and 'lp_dual_simplex<' in r['Kernel_Name'] and int(r['Grid_Size']) >= g // 4:
                    rows.append((int(r.get('Dispatch_Id', 0)), r['Kernel_Name'][:60], r['Grid_Size'], float(r['Counter_Value'])))
        for _, kn, gs, v in sorted(rows)[:16]:
            fh.write(f'{tag},"{kn}",{gs},{name},{v:.6f}\n')
# K1b
st = by_grid(O + '/k1b_trace', O + f'/{TAG}_k1b_kernel_by_grid.csv')
if st:
    os.system(f'cp {st} {O}/{TAG}_k1b_kernel_stats.csv')
fe, wr = counter('k1b_fetch', 'FETCH_SIZE', 'lp_dual_simplex_big'), counter('k1b_write', 'WRITE_SIZE', 'lp_dual_simplex_big')
if fe and wr:
    g, f_kb = steady(fe)
    _, w_kb = steady(wr)
    dur = None
    for r in csv.DictReader(open(O + f'/{TAG}_k1b_kernel_by_grid.csv')):
        if 'lp_dual_simplex_big' in r['kernel'] and int(r['grid_x']) == g:
            dur = float(r['avg_us'])
    k1b = {"command": "rocprofv3 (trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes) -- python3 scripts/c5_tree.py 1024 8 1",
           "kernel": f"lp_dual_simplex_big (K1b) at 1024 x 512, grid {g} threads (1024 nodes + their plunge of depth 8 per launch)",
           "fetch_size_KB_raw": f_kb, "write_size_KB": w_kb, "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024,
           "avg_launch_us": dur,
           "hbm_GBps": None if not dur else (2 * f_kb + w_kb) * 1024 / (dur * 1e-6) / 1e9}
    json.dump(k1b, open(O + f'/{TAG}_k1b_pmc.json', 'w'), indent=1)
    print(json.dumps(k1b))
if glob.glob(O + '/c4_trace/*/*kernel_trace.csv'):
    by_grid(O + '/c4_trace', O + f'/{TAG}_c4_kernel_by_grid.csv')
if glob.glob(O + '/k1c_trace/*/*kernel_trace.csv'):
    by_grid(O + '/k1c_trace', O + f'/{TAG}_k1c_kernel_by_grid.csv')
print(open(O + f'/{TAG}_kernel_by_grid.csv').read()[:3000])
