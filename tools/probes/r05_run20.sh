#!/bin/bash
# round 5, GPU call 20: grid of the value-net optimizer launch (JH_RB_OPTIM_GRID; default 512 workgroups = 5.7 passes per thread at Rainbow's 3 M parameters)
mkdir -p gpurun_out
for rep in 1 2; do for g in 512 384 256 192; do
JH_RB_OPTIM_GRID=$g python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('grid $g', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4), {k.replace('jh_',''):v for k,v in d['lib_kernel_avg_us'].items() if 'optim' in k})"
done; done 2>&1 | tee gpurun_out/r05_run20_optim_grid.txt
for g in 512 256; do
JH_RB_OPTIM_GRID=$g python tools/bench_apex.py --updates 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('apex grid $g', 'learn_ms', round(d['ms_per_learn_only'],4), {n.replace('jh_',''):v['avg_us'] for n,v in k.items() if 'optim' in n})"
done 2>&1 | tee -a gpurun_out/r05_run20_optim_grid.txt
