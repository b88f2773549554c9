#!/bin/bash
# A/B builds of libjorldy_hip.so (ab/lib_<name>.so) on the Ape-X learner at B = 512 (tools/bench_apex.py): tools/probes/ab_apex_lib.sh reps name1 name2 ...
reps=$1; shift
for rep in $(seq $reps); do for v in "$@"; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
python tools/bench_apex.py --updates 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v', 'learn_ms', round(d['ms_per_learn_only'],4), {n.replace('jh_tgemm_',''):v['avg_us'] for n,v in k.items() if 'tgemm' in n})
"; done; done
