#!/bin/bash
# A/B an environment switch on bench.py's PPO leg:  tools/probes/ab_env.sh VAR A B [reps]
var=$1; a=$2; b=$3; reps=${4:-3}
for rep in $(seq $reps); do
  for v in $a $b; do
    env $var=$v python bench.py --steps 100 --warmup 10 --no-rainbow --no-cpu-baseline --no-roofline --no-hopper --no-apex 2>/dev/null | grep "^{" > /tmp/l.json
    python - $var $v <<'PY'
import json, sys
d = json.loads(open("/tmp/l.json").readline())
act = d["collector_host_us_per_timestep"]["act_us_per_step"]
print(f"{sys.argv[1]}={sys.argv[2]}", "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "act_us", round(act, 3), "learner_part_ms", round(d["ms_per_step"] - act * 0.128, 4))
PY
  done
done
