#!/bin/bash
# round 5, GPU call 7: the small Hopper-path kernels with their fetches in flight -- tests, then old vs new on the Hopper learner
mkdir -p gpurun_out
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
timeout 900 python -m pytest tests/test_agents_gpu.py tests/test_baseline_width_gpu.py tests/test_kernels_gpu.py tests/test_dp_two_ranks_gpu.py -x -q > gpurun_out/r05_run7_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run7_tests.txt
tail -5 gpurun_out/r05_run7_tests.txt
for rep in 1 2; do for v in old new; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items()})
"; done; done 2>&1 | tee gpurun_out/r05_run7_ab_hopper.txt
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
