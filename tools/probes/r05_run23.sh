#!/bin/bash
# round 5, GPU call 23: the persistent acting kernel's relay mailbox replicated per XCD (lib tk3) against tk2: collector tests, PPO headline (with the kernel's
# own wait / compute stamps), Hopper end to end
mkdir -p gpurun_out
cp ab/lib_tk3.so jorldy_amd/csrc/libjorldy_hip.so
timeout 900 python -m pytest tests/test_agents_gpu.py -x -q -k "control_env or capture_equals or collector or lookahead or function_table" > gpurun_out/r05_run23_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run23_tests.txt
tail -3 gpurun_out/r05_run23_tests.txt
{
for rep in 1 2 3; do for v in tk2 tk3; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v ppo', d['ms_per_step'], round(d['value']), d['roofline']['kernel'], round(d['roofline']['frac'],4), d.get('collector_host_us_per_timestep'))
"; done; done
for v in tk2 tk3; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
JH_PERSIST_DEBUG=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep jh_persist | tail -2 | sed "s/^/$v /"
timeout 200 python tools/bench_hopper.py --iters 2 --e2e-full 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v hopper e2e', round(d['env_transitions_per_s_end_to_end']), round(d['ms_per_iteration'],1), round(d['collector']['act_us_per_step'],2))
"; done
} 2>&1 | tee gpurun_out/r05_run23_ab.txt
cp ab/lib_tk3.so jorldy_amd/csrc/libjorldy_hip.so
