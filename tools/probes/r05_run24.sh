#!/bin/bash
# round 5, GPU call 24: rocprofv3 kernel statistics of the FINAL library for the Rainbow, Ape-X and Hopper learners and the bench's PPO leg
mkdir -p gpurun_out
timeout 300 tools/profile_stats_cmd.sh r05_rainbow python tools/bench_rainbow.py --updates 300
timeout 300 tools/profile_stats_cmd.sh r05_apex python tools/bench_apex.py --updates 100
timeout 300 tools/profile_stats_cmd.sh r05_hopper python tools/bench_hopper.py --iters 3
timeout 400 tools/profile_stats_cmd.sh r05_bench python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-apex --no-rainbow --no-hopper --no-dqn --no-variants
