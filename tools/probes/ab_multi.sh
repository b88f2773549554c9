#!/bin/bash
# A/B several environment settings on bench.py's PPO leg, alternating on ONE box:
#   tools/probes/ab_multi.sh reps "NAME1:VAR=a VAR2=b" "NAME2:VAR=c" ...
reps=$1; shift
for rep in $(seq $reps); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py --steps 100 --warmup 10 --no-rainbow --no-cpu-baseline --no-roofline --no-hopper --no-apex 2>/dev/null | grep "^{" > /tmp/l.json
    python - "$name" <<'PY'
import json, sys
d = json.loads(open("/tmp/l.json").readline())
act = d["collector_host_us_per_timestep"]["act_us_per_step"]
c = d["collector_host_us_per_timestep"]
print(sys.argv[1].ljust(14), "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "act_us", round(act, 3), "learner_part_ms", round(d["ms_per_step"] - act * 0.128, 4),
      "steady", round(c.get("act_steady_us_per_step", 0), 3), "first", round(c.get("first_step_us_per_run", 0), 1), "query", round(c.get("value_query_us_per_run", 0), 1), "commit", round(c.get("commit_us_per_run", 0), 1))
PY
  done
done
