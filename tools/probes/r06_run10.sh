#!/bin/bash
# round 6, GPU call 10: the DP step at ONE rank (JH_FORCE_DIST=1): no DP | RCCL | peer pointers (critic exchange inside the select launch); two-rank peer tests
timeout 900 python -m pytest tests/test_dp_two_ranks_gpu.py -x -q -k "peer or plumbing" 2>&1 | tail -3
B="--steps 40 --warmup 10 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for mode in none rccl peer; do
  if [ $mode = none ]; then E=""; elif [ $mode = rccl ]; then E="JH_FORCE_DIST=1"; else E="JH_FORCE_DIST=1 JH_DP_COLLECTIVE=peer"; fi
  env $E timeout 300 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$mode', 'value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4))
"
done; done 2>&1 | tee gpurun_out/r06_run10_dp_one_rank.txt
