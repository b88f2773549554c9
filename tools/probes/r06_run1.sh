#!/bin/bash
# round 6, GPU call 1: the register-blocked tiles of the LDS-DMA kernel: engine tests, then the tile / split / XCD-order sweep on the Ape-X and Hopper learners
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_0_tgemm_gpu.py -x -q > gpurun_out/r06_run1_tgemm_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_run1_tgemm_tests.txt
tail -5 gpurun_out/r06_run1_tgemm_tests.txt
timeout 900 python tools/tgemm_sweep.py --apex --hopper > gpurun_out/r06_run1_sweep.txt 2> gpurun_out/r06_run1_sweep.err; echo "sweep rc $?"
tail -3 gpurun_out/r06_run1_sweep.err
cat gpurun_out/r06_run1_sweep.txt | cut -c1-400
