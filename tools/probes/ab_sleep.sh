for s in 2 0 1 3 2 0; do
  JH_PERSIST_SLEEP=$s timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-rainbow 2>/dev/null > /tmp/ab.json
  python -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('sleep $s', round(d['value']), round(d['ms_per_step'],3), round(d['collector_host_us_per_timestep']['act_us_per_step'],2))"
done
