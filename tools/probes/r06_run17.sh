#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_cmd.sh r06_ppo_atari python tools/bench_ppo_atari.py --iters 4 > gpurun_out/r17_log.txt 2>&1
tail -20 gpurun_out/r17_log.txt
rm -rf gpurun_out/prof_* gpurun_out/pmc_*_FETCH_SIZE gpurun_out/pmc_*_WRITE_SIZE
ls -la gpurun_out | grep ppo_atari
