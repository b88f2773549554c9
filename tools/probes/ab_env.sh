# usage: ab_env.sh VAR  -- A/B of bench.py (PPO leg) with VAR=0 / VAR=1, two rounds
for s in 0 1 0 1; do
  env $1=$s timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-rainbow 2>/dev/null > /tmp/ab.json
  python -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$1=$s', round(d['value']), round(d['ms_per_step'],3), round(d['last_result']['critic_loss'],4))"
done
