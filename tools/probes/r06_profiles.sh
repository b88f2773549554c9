#!/bin/bash
# round 6: the rocprofv3 evidence of the final library -- kernel statistics (one pass) + HBM traffic counters (two more passes, FETCH_SIZE / WRITE_SIZE apart)
# of the driver's bench command and of the Rainbow / Hopper / Ape-X learners; SQ counters of the learner kernels; the scaled-shape rooflines
mkdir -p gpurun_out
BARGS="--steps 20 --warmup 5 --no-rainbow --no-apex --no-hopper --no-dqn --no-ppo-atari --no-variants --no-cpu-baseline"
bash tools/profile_bench.sh r06_bench $BARGS > gpurun_out/r06_profiles_log.txt 2>&1
bash tools/pmc_bench.sh r06 $BARGS >> gpurun_out/r06_profiles_log.txt 2>&1
bash tools/profile_cmd.sh r06_rainbow python tools/bench_rainbow.py >> gpurun_out/r06_profiles_log.txt 2>&1
bash tools/profile_cmd.sh r06_hopper python tools/bench_hopper.py --iters 3 >> gpurun_out/r06_profiles_log.txt 2>&1
bash tools/profile_cmd.sh r06_apex python tools/bench_apex.py --updates 100 >> gpurun_out/r06_profiles_log.txt 2>&1
bash tools/profile_cmd.sh r06_ppo_atari python tools/bench_ppo_atari.py --iters 4 >> gpurun_out/r06_profiles_log.txt 2>&1
JH_FORCE_DIST=1 bash tools/profile_stats_cmd.sh r06_force_dist python bench.py $BARGS --no-roofline >> gpurun_out/r06_profiles_log.txt 2>&1
JH_FORCE_DIST=1 JH_DP_COLLECTIVE=peer bash tools/profile_stats_cmd.sh r06_force_dist_peer python bench.py $BARGS --no-roofline >> gpurun_out/r06_profiles_log.txt 2>&1
{
bash tools/sq_cmd.sh r06_hopper python tools/bench_hopper.py --iters 2
bash tools/sq_cmd.sh r06_apex python tools/bench_apex.py --updates 60
bash tools/sq_cmd.sh r06_rainbow python tools/bench_rainbow.py
} > gpurun_out/r06_sq_counters_learner_kernels.txt 2>&1
timeout 600 python tools/roofline_scaled.py > gpurun_out/r06_roofline_scaled.json 2> gpurun_out/r06_roofline_scaled.err
rm -rf gpurun_out/prof_* gpurun_out/pmc_*_FETCH_SIZE gpurun_out/pmc_*_WRITE_SIZE gpurun_out/pmc_*_sq
ls -la gpurun_out | grep r06_ | head -40
head -12 gpurun_out/r06_bench_kernel_stats.csv | cut -c1-160
