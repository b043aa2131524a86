R=$GRAFT_REPO_ROOT
for st in 50 200 600; do
  timeout -k 10 250 python3 $R/bench.py --steps $st --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $st', 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'launch_ms %.3f' % d['roofline']['launch_ms'], 'pivots/LP %.2f' % d['config']['mean_pivots_per_lp'], flush=True)"
done
