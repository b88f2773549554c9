#!/bin/bash
# round 5, GPU call 4: the tests that failed in call 3 (regenerated fixture, fixed test), the DP structure at one rank, kernel trace + PMC of the PPO leg
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_baseline_width_gpu.py tests/test_agents_gpu.py tests/test_compat_gpu.py -q -k "halfcheetah or ant_mb256 or more_than_8 or compat or constructs or raises" > gpurun_out/r05_run4_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run4_tests.txt
tail -6 gpurun_out/r05_run4_tests.txt
B="--steps 60 --warmup 10 --no-rainbow --no-cpu-baseline --no-hopper --no-apex --no-dqn --no-variants --no-roofline"
python bench.py $B 2>/dev/null | grep '^{' > gpurun_out/r05_run4_single.json
JH_FORCE_DIST=1 python bench.py $B 2>/dev/null | grep '^{' > gpurun_out/r05_bench_force_dist.json
JH_FORCE_DIST=1 JH_DP_EXACT_CRITIC=0 python bench.py $B 2>/dev/null | grep '^{' > gpurun_out/r05_bench_force_dist_noexact.json
python - <<'PY'
import json
for f in ("r05_run4_single", "r05_bench_force_dist", "r05_bench_force_dist_noexact"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").readline())
        print(f, "ms_per_step", round(d["ms_per_step"], 4), "value", round(d["value"]), d["collector_host_us_per_timestep"]["act_us_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
tools/profile_bench.sh r05_bench --steps 100 --warmup 10 --no-rainbow --no-cpu-baseline --no-hopper --no-apex --no-dqn --no-variants | grep -i 'pmb\|fused\|adam\|persist'
JH_FORCE_DIST=1 tools/profile_bench.sh r05_force_dist --steps 100 --warmup 10 --no-rainbow --no-cpu-baseline --no-hopper --no-apex --no-dqn --no-variants --no-roofline | head -16 | cut -c1-150
tools/pmc_bench.sh r05 --steps 40 --warmup 5 --no-rainbow --no-cpu-baseline --no-hopper --no-apex --no-dqn --no-variants --no-roofline | grep -i 'pmb\|fused\|adam'
