#!/bin/bash
# round 5, GPU call 1: the suite on the new library, then old-vs-new on the PPO leg, then a kernel trace of the new one
mkdir -p gpurun_out
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r05_run1_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run1_tests.txt
tail -5 gpurun_out/r05_run1_tests.txt
timeout 400 tools/probes/ab_lib.sh 3 > gpurun_out/r05_run1_ab.txt 2>&1
cat gpurun_out/r05_run1_ab.txt
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
timeout 300 tools/profile_bench.sh r05_run1 --steps 100 --warmup 10 --no-rainbow --no-cpu-baseline --no-hopper --no-apex --no-roofline | grep -i 'pmb\|fused\|adam\|persist'
