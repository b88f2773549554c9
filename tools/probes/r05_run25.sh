#!/bin/bash
# round 5, GPU call 25: conv1 forward with the waves' tile lists rotated by workgroup (lib rot) against tk2: value-net tests, Ape-X and Rainbow learners
mkdir -p gpurun_out
cp ab/lib_rot.so jorldy_amd/csrc/libjorldy_hip.so
timeout 600 python -m pytest tests/test_rbnet_gpu.py -x -q > gpurun_out/r05_run25_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run25_tests.txt
tail -3 gpurun_out/r05_run25_tests.txt
for rep in 1 2; do for v in tk2 rot; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
python tools/bench_apex.py --updates 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('apex $v', 'learn_ms', round(d['ms_per_learn_only'],4), {n.replace('jh_',''):v['avg_us'] for n,v in k.items() if 'conv1' in n})"
python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('rainbow $v', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4), {k.replace('jh_',''):v for k,v in d['lib_kernel_avg_us'].items() if 'conv1' in k})"
done; done 2>&1 | tee gpurun_out/r05_run25_ab.txt
cp ab/lib_rot.so jorldy_amd/csrc/libjorldy_hip.so
