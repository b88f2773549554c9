#!/bin/bash
# A/B an environment variable on the Ape-X learner at B = 512 (tools/bench_apex.py): tools/probes/ab_apex_env.sh VAR reps v1 v2 ...
var=$1; reps=$2; shift 2
for rep in $(seq $reps); do for v in "$@"; do
env $var=$v python tools/bench_apex.py --updates 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$var=$v', 'learn_ms', round(d['ms_per_learn_only'],4), {n.replace('jh_tgemm_',''):v['avg_us'] for n,v in k.items() if 'fwd' in n})
"; done; done
