#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ppo_cnn_gpu.py tests/test_compat_gpu.py -x -q > gpurun_out/r18.log 2>&1; echo "rc=$?" >> gpurun_out/r18.log
tail -30 gpurun_out/r18.log
