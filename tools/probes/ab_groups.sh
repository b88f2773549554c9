for g in 2 1 2 1; do
  JH_PERSIST_GROUPS=$g timeout 120 python bench.py --no-cpu-baseline --no-rainbow --no-roofline 2>/dev/null > /tmp/ab.json
  python -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('groups $g', round(d['value']), round(d['ms_per_step'],3), d['collector_host_us_per_timestep'])"
done
