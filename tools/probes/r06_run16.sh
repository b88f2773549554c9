#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_ppo_atari.py --cpu > gpurun_out/r16_ppo_atari.json 2> gpurun_out/r16_ppo_atari.err; echo "rc=$?"
tail -3 gpurun_out/r16_ppo_atari.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r16_ppo_atari.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("ms_per_iteration", "learner_transitions_per_s", "learner_updates_per_s", "us_per_update", "learn_in_hipgraph", "x_cpu_reference")})
print(d["cpu_reference"])
for k, v in d["lib_kernels"].items(): print(k, v)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r16_prof -o ppo_atari -- python $GRAFT_REPO_ROOT/tools/bench_ppo_atari.py --iters 6 > $GRAFT_REPO_ROOT/gpurun_out/r16_prof.log 2>&1; echo "rocprof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/r16_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {}'
