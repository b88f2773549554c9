#!/bin/bash
# A/B an environment variable on the Rainbow learner (tools/bench_rainbow.py): tools/probes/ab_rb_env.sh VAR reps v1 v2 ...
var=$1; reps=$2; shift 2
for rep in $(seq $reps); do for v in "$@"; do
env $var=$v python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$var=$v', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4))
"; done; done
