#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_baseline_width_gpu.py -x -q -k "ppo" > gpurun_out/r22.log 2>&1; echo "rc=$?" >> gpurun_out/r22.log
tail -6 gpurun_out/r22.log
JH_PPO_LOSS_TICKET=1 timeout 600 python -m pytest tests/test_baseline_width_gpu.py -x -q -k "one_launch or folded" > gpurun_out/r22b.log 2>&1; echo "rc=$?" >> gpurun_out/r22b.log
tail -3 gpurun_out/r22b.log
for rep in a b; do for m in 1 0; do JH_PPO_LOSS_TICKET=$m timeout 300 python tools/bench_hopper.py --iters 6 > gpurun_out/r22_hopper_$rep$m.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r22_hopper_$rep$m.json')); print('ticket=$m', d['ms_per_iteration'], d['learner_transitions_per_s'])"; done; done
