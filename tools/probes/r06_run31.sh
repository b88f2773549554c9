#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_agents_gpu.py -x -q -k "two_halves_hands_over or two_independent_halves or device_side_sum" > gpurun_out/r31.log 2>&1; echo "rc=$?" >> gpurun_out/r31.log
tail -25 gpurun_out/r31.log
