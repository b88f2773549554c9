#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_agents_gpu.py -x -q -k "device_side_sum or two_independent_halves" > gpurun_out/r30.log 2>&1; echo "rc=$?" >> gpurun_out/r30.log
tail -5 gpurun_out/r30.log
for rep in a b; do for m in 0 1; do JH_PERSIST_REDUCE=$m timeout 300 python tools/bench_hopper.py --e2e-full --iters 4 > gpurun_out/r30_e2e_$rep$m.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r30_e2e_$rep$m.json')); c=d['collector']; print('split, reduce=$m', round(d['ms_per_iteration'],2), round(d['env_transitions_per_s_end_to_end']), round(c['act_us_per_step'],2), round(c['env_us_per_step'],2))"; done; done
