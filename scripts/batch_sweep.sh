R=$GRAFT_REPO_ROOT
for b in 4096 8192 16384 8192; do
  timeout -k 10 250 python3 $R/bench.py --batch $b --steps 100 --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $b', 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'launch_ms %.3f' % d['roofline']['launch_ms'], 'pivots/LP %.2f' % d['config']['mean_pivots_per_lp'], flush=True)"
done
