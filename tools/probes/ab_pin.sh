for rep in 1 2 3; do
  for pin in 0 1; do
    JH_NO_PIN=$((1-pin)) timeout 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null > /tmp/ab.json
    python -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('pin $pin', round(d['value']), round(d['ms_per_step'],3), d['collector_host_us_per_timestep']['act_us_per_step'], round(d['rainbow']['value']), d['config'].get('host_cores_per_rank'))"
  done
done
