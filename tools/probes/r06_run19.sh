#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_baseline_width_gpu.py -x -q -k "ppo" > gpurun_out/r19.log 2>&1; echo "rc=$?" >> gpurun_out/r19.log
tail -30 gpurun_out/r19.log
for m in 0 1; do JH_PPO_ONEPASS=$m timeout 300 python tools/bench_hopper.py --iters 6 > gpurun_out/r19_hopper_$m.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r19_hopper_$m.json')); print('onepass=$m', d['ms_per_iteration'], d['learner_transitions_per_s'])"; done
for m in 0 1; do JH_PPO_ONEPASS=$m timeout 300 python tools/bench_hopper.py --iters 6 > gpurun_out/r19_hopper_b$m.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r19_hopper_b$m.json')); print('onepass=$m', d['ms_per_iteration'], d['learner_transitions_per_s'])"; done
