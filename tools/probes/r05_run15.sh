#!/bin/bash
# round 5, GPU call 15: SQ wait / issue / MFMA-busy counters and the effective clock of the learner kernels (Hopper, Ape-X, Rainbow, PPO headline)
mkdir -p gpurun_out
timeout 300 tools/sq_cmd.sh r05_hopper python tools/bench_hopper.py --iters 2 2>&1 | cut -c1-420
timeout 300 tools/sq_cmd.sh r05_apex python tools/bench_apex.py --updates 40 2>&1 | cut -c1-420
timeout 300 tools/sq_cmd.sh r05_rainbow python tools/bench_rainbow.py --updates 100 2>&1 | cut -c1-420
timeout 300 tools/sq_cmd.sh r05_ppo python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-variants --no-cpu-baseline --no-roofline 2>&1 | cut -c1-420
