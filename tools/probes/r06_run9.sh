#!/bin/bash
# round 6, GPU call 9: partial heads summed on the device (persistent acting kernel): collector tests, then Hopper end to end and the PPO headline with / without
timeout 1500 python -m pytest tests/test_agents_gpu.py -x -q -k "collector or lookahead or capture or wide_action or native_act" > gpurun_out/r06_run9_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_run9_tests.txt
tail -4 gpurun_out/r06_run9_tests.txt
{
for rep in 1 2; do for r in 0 1; do
echo "JH_PERSIST_REDUCE=$r hopper e2e:"
JH_PERSIST_REDUCE=$r timeout 300 python tools/bench_hopper.py --iters 2 --e2e-full 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['collector']
print('  ms_per_iter', round(d['ms_per_iteration'],2), 'e2e transitions/s', round(d['env_transitions_per_s_end_to_end']), {k:round(v,2) for k,v in c.items() if isinstance(v,float)})
"
echo "JH_PERSIST_REDUCE=$r ppo cartpole:"
JH_PERSIST_REDUCE=$r timeout 300 python bench.py --steps 40 --warmup 10 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), d.get('acting',{}).get('host_us_per_timestep'))
"
done; done
} 2>&1 | tee gpurun_out/r06_run9_ab.txt
