#!/bin/bash
# round 5, GPU call 3: the full GPU suite (wide heads, function-table collector, tightened parity asserts, bench contract)
mkdir -p gpurun_out
export JH_MARGINS_OUT=gpurun_out/r05_margins_run3.json
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_run3_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run3_tests.txt
tail -40 gpurun_out/r05_run3_tests.txt
