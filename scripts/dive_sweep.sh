R=$GRAFT_REPO_ROOT
for d in 4 6 8 4 8; do
  timeout -k 10 150 python3 $R/bench.py --dive $d --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dive $d', 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'launch_ms %.3f' % d['roofline']['launch_ms'], 'pivots/LP %.2f' % d['config']['mean_pivots_per_lp'], 'children/step', d['config']['dive_children_per_step'], flush=True)"
done
