#!/bin/bash
# A/B two builds of libjorldy_hip.so (ab/lib_old.so, ab/lib_new.so) on bench.py's PPO leg: tools/probes/ab_lib.sh [reps]
reps=${1:-3}
for rep in $(seq $reps); do
  for v in old new; do
    cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
    python bench.py --steps 100 --warmup 10 --no-rainbow --no-cpu-baseline --no-hopper --no-apex 2>/dev/null | grep "^{" > /tmp/l.json
    python - $v <<'PY'
import json, sys
d = json.loads(open("/tmp/l.json").readline())
act = d["collector_host_us_per_timestep"]["act_us_per_step"]
e = d["event_timing"]["other_kernels_avg_us_incl_event_pair"]
print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "act_us", round(act, 3), "learner_part_ms", round(d["ms_per_step"] - act * 0.128, 4),
      {k: v for k, v in e.items() if "fused" in k or "adam" in k})
PY
  done
done
