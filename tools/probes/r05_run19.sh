#!/bin/bash
# round 5, GPU call 19: lib pw3 (persistent acting kernel up to 512 granules; block-order host collection for > 16 rows only): collector + bench tests,
# Hopper end to end, PPO headline against tg6
mkdir -p gpurun_out
cp ab/lib_pw3.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1200 python -m pytest tests/test_agents_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/r05_run19_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run19_tests.txt
tail -4 gpurun_out/r05_run19_tests.txt
{
for v in tg6 pw3; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 200 python tools/bench_hopper.py --iters 2 --e2e-full 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v hopper e2e', round(d['env_transitions_per_s_end_to_end']), round(d['ms_per_iteration'],1), d['collector'])
"; done
for rep in 1 2 3; do for v in tg6 pw3; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v ppo', d['ms_per_step'], round(d['value']), d['roofline']['kernel'], d['roofline']['frac'])
"; done; done
} 2>&1 | tee gpurun_out/r05_run19_ab.txt
cp ab/lib_pw3.so jorldy_amd/csrc/libjorldy_hip.so
