#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_cmd.sh r06_hopper python tools/bench_hopper.py --iters 3 > gpurun_out/r23_log.txt 2>&1
bash tools/sq_cmd.sh r06_hopper python tools/bench_hopper.py --iters 2 > gpurun_out/r23_sq_hopper.txt 2>&1
rm -rf gpurun_out/prof_* gpurun_out/pmc_*_FETCH_SIZE gpurun_out/pmc_*_WRITE_SIZE gpurun_out/pmc_*_sq
head -14 gpurun_out/r06_hopper_kernel_stats.csv | cut -c1-140
