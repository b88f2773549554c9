#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_baseline_width_gpu.py tests/test_agents_gpu.py tests/test_compat_gpu.py -x -q > gpurun_out/r20.log 2>&1; echo "rc=$?" >> gpurun_out/r20.log
tail -8 gpurun_out/r20.log
timeout 900 python -m pytest tests/test_dp_two_ranks_gpu.py tests/test_learning_curve_gpu.py -x -q -k "ppo" > gpurun_out/r20b.log 2>&1; echo "rc=$?" >> gpurun_out/r20b.log
tail -4 gpurun_out/r20b.log
