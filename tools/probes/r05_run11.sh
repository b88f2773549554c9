#!/bin/bash
# round 5, GPU call 11: lib tg3 (plain / epilogue kernel variants chosen per launch, four LDS buffers for the PPO 2048-row launches, pipelined value-net
# optimizer, commit kernel one piece per thread) against r8 and tg2: tests, then Rainbow, Ape-X, Hopper, and the headline bench
mkdir -p gpurun_out
cp ab/lib_tg3.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1500 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_agents_gpu.py tests/test_kernels_gpu.py tests/test_capture_gpu.py tests/test_baseline_width_gpu.py tests/test_actors_gpu.py -x -q > gpurun_out/r05_run11_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run11_tests.txt
tail -5 gpurun_out/r05_run11_tests.txt
{
bash tools/probes/ab_rb_lib.sh 2 tg2 tg3
bash tools/probes/ab_apex_lib.sh 2 r8 tg2 tg3
for rep in 1 2; do for v in tg2 tg3; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items()})
"; done; done
for rep in 1 2; do for v in r8 tg3; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v ppo', d['ms_per_step'], round(d['value']), d.get('roofline',{}).get('frac'))
"; done; done
} 2>&1 | tee gpurun_out/r05_run11_ab.txt
cp ab/lib_tg3.so jorldy_amd/csrc/libjorldy_hip.so
