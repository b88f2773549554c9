#!/bin/bash
# round 5, GPU call 26: host prefetch of the successors' granules during the root-row read of a two-timestep exchange (lib pf) against tk2: PPO headline
mkdir -p gpurun_out
cp ab/lib_pf.so jorldy_amd/csrc/libjorldy_hip.so
timeout 600 python -m pytest tests/test_agents_gpu.py -x -q -k "lookahead or collector or function_table" > gpurun_out/r05_run26_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run26_tests.txt
tail -3 gpurun_out/r05_run26_tests.txt
for rep in 1 2 3 4; do for v in tk2 pf; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d.get('collector_host_us_per_timestep') or {}
print('$v ppo', round(d['ms_per_step'],4), round(d['value']), round(c.get('act_us_per_step',0),3), round(c.get('act_steady_us_per_step',0),3))
"; done; done 2>&1 | tee gpurun_out/r05_run26_ab.txt
cp ab/lib_pf.so jorldy_amd/csrc/libjorldy_hip.so
