#!/bin/bash
# round 5, GPU call 6: DQN learning curve; where the host's time goes in an acting exchange
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_learning_curve_gpu.py -x -q -s -k dqn > gpurun_out/r05_run6_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run6_tests.txt
tail -6 gpurun_out/r05_run6_tests.txt
B="--steps 60 --warmup 10 --no-rainbow --no-cpu-baseline --no-hopper --no-apex --no-dqn --no-variants --no-roofline"
JH_COLLECT_DEBUG=1 JH_PERSIST_DEBUG=1 python bench.py $B 2> gpurun_out/r05_run6_dbg.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], d['collector_host_us_per_timestep'])"
grep 'jh_collect\|jh_persist' gpurun_out/r05_run6_dbg.err | tail -4
