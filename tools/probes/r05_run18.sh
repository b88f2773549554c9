#!/bin/bash
# round 5, GPU call 18: the persistent acting kernel with up to 512 observation granules per exchange (lib pw2 = pw + block-order host collection; config.ppo.mujoco's 32 workers x 11) against tg6:
# collector tests, Hopper end to end at 32 workers, and the PPO headline (whose 96-granule exchange must not move)
mkdir -p gpurun_out
cp ab/lib_pw2.so jorldy_amd/csrc/libjorldy_hip.so
timeout 900 python -m pytest tests/test_agents_gpu.py -x -q -k "control_env or capture_equals or collector or lookahead or function_table" > gpurun_out/r05_run18_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run18_tests.txt
tail -6 gpurun_out/r05_run18_tests.txt
{
for rep in 1 2; do for v in tg6 pw pw2; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 200 python tools/bench_hopper.py --iters 2 --e2e-full 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v hopper e2e', round(d['env_transitions_per_s_end_to_end']), round(d['ms_per_iteration'],1), d['collector'])
"; done; done
for rep in 1 2 3; do for v in tg6 pw pw2; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v ppo', d['ms_per_step'], round(d['value']), d.get('roofline',{}).get('frac'))
"; done; done
} 2>&1 | tee gpurun_out/r05_run18_ab.txt
cp ab/lib_pw2.so jorldy_amd/csrc/libjorldy_hip.so
