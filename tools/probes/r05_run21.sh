#!/bin/bash
# round 5, GPU call 21: two-level ticket in the value-net optimizer (lib tk) against pw3, over grids; tests of the optimizer paths first
mkdir -p gpurun_out
cp ab/lib_tk.so jorldy_amd/csrc/libjorldy_hip.so
timeout 900 python -m pytest tests/test_rbnet_gpu.py tests/test_capture_gpu.py -x -q > gpurun_out/r05_run21_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run21_tests.txt
tail -3 gpurun_out/r05_run21_tests.txt
for rep in 1 2; do for v in pw3 tk; do for g in 512 1024 2048; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
JH_RB_OPTIM_GRID=$g python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v grid $g', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4), {k.replace('jh_',''):v for k,v in d['lib_kernel_avg_us'].items() if 'optim' in k})"
done; done; done 2>&1 | tee gpurun_out/r05_run21_optim.txt
for v in pw3 tk; do for g in 512 1024; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
JH_RB_OPTIM_GRID=$g python tools/bench_apex.py --updates 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('apex $v grid $g', 'learn_ms', round(d['ms_per_learn_only'],4), {n.replace('jh_',''):v['avg_us'] for n,v in k.items() if 'optim' in n})"
done; done 2>&1 | tee -a gpurun_out/r05_run21_optim.txt
cp ab/lib_tk.so jorldy_amd/csrc/libjorldy_hip.so
