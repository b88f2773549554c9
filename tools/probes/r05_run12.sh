#!/bin/bash
# round 5, GPU call 12: lib tg4 (commit kernel: four pieces in flight at the old grid; optimizer prefetch predicated; no deep-buffer variant)
# against tg2 (Rainbow) and r8 (PPO headline): tests of the touched paths, then the A/B
mkdir -p gpurun_out
cp ab/lib_tg4.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1500 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_agents_gpu.py tests/test_kernels_gpu.py tests/test_actors_gpu.py -x -q > gpurun_out/r05_run12_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run12_tests.txt
tail -3 gpurun_out/r05_run12_tests.txt
{
bash tools/probes/ab_rb_lib.sh 3 tg2 tg4
for rep in 1 2 3; do for v in r8 tg4; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v ppo', d['ms_per_step'], round(d['value']), d.get('roofline',{}).get('frac'))
"; done; done
} 2>&1 | tee gpurun_out/r05_run12_ab.txt
cp ab/lib_tg4.so jorldy_amd/csrc/libjorldy_hip.so
