#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
JH_PMB_GEN16=1 timeout 1200 python -m pytest tests/test_agents_gpu.py tests/test_kernels_gpu.py tests/test_baseline_width_gpu.py -x -q -k "ppo or pponet or collector" > gpurun_out/r34.log 2>&1; echo "rc=$?" >> gpurun_out/r34.log
tail -5 gpurun_out/r34.log
for rep in a b; do for m in 0 1; do JH_PMB_GEN16=$m timeout 300 python tools/bench_hopper.py --e2e --iters 10 > gpurun_out/r34_share_$rep$m.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r34_share_$rep$m.json')); c=d['collector']; print('gen16=$m', round(d['ms_per_iteration'],2), round(d['env_transitions_per_s_end_to_end']), [ (k, v['avg_us']) for k, v in list(d['lib_kernels'].items())[:6]])"; done; done
