#!/bin/bash
# A/B builds of libjorldy_hip.so (ab/lib_<name>.so) on the Rainbow learner (tools/bench_rainbow.py): tools/probes/ab_rb_lib.sh reps name1 name2 ...
reps=$1; shift
for rep in $(seq $reps); do for v in "$@"; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4), {k.replace('jh_tgemm_',''):v for k,v in d['lib_kernel_avg_us'].items() if 'tgemm' in k})
"; done; done
