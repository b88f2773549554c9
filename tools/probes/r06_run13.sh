#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dp_two_ranks_gpu.py -x -q -k "bench" > gpurun_out/r13_dp_bench.log 2>&1; echo "dp bench rc=$?" >> gpurun_out/r13_dp_bench.log
tail -15 gpurun_out/r13_dp_bench.log
