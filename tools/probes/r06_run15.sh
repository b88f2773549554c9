#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_dp_two_ranks_gpu.py -x -q -k "cnn" > gpurun_out/r15_dp_cnn.log 2>&1; echo "rc=$?" >> gpurun_out/r15_dp_cnn.log
tail -30 gpurun_out/r15_dp_cnn.log
timeout 1200 python -m pytest tests/test_learning_curve_gpu.py -x -q -s -k "cnn_head" > gpurun_out/r15_curve_cnn.log 2>&1; echo "rc=$?" >> gpurun_out/r15_curve_cnn.log
tail -30 gpurun_out/r15_curve_cnn.log
