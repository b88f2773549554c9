#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_baseline_width_gpu.py -x -q -k "ppo" > gpurun_out/r21.log 2>&1; echo "rc=$?" >> gpurun_out/r21.log
tail -12 gpurun_out/r21.log
for rep in a b; do for m in 0 1; do JH_PPO_NORM_FOLD=$m timeout 300 python tools/bench_hopper.py --iters 6 > gpurun_out/r21_hopper_$rep$m.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r21_hopper_$rep$m.json')); print('fold=$m', d['ms_per_iteration'], d['learner_transitions_per_s'])"; done; done
