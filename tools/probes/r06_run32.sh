#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_agents_gpu.py tests/test_actors_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/r32.log 2>&1; echo "rc=$?" >> gpurun_out/r32.log
tail -4 gpurun_out/r32.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
