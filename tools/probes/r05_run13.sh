#!/bin/bash
# round 5, GPU call 13: Rainbow learner, every library kernel's average, libraries r8 / tg2 / tg4
mkdir -p gpurun_out
for rep in 1 2; do for v in r8 tg2 tg4; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4), {k.replace('jh_',''):v for k,v in d['lib_kernel_avg_us'].items() if 'tgemm' not in k})
"; done; done 2>&1 | tee gpurun_out/r05_run13_rb_kernels.txt
cp ab/lib_tg4.so jorldy_amd/csrc/libjorldy_hip.so
