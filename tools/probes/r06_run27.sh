#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/bench_hopper.py --e2e --iters 10 > gpurun_out/r27_hopper_share.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r27_hopper_share.json"))
print({k: d[k] for k in ("ms_per_iteration", "learner_transitions_per_s", "env_transitions_per_s_end_to_end", "minibatch_updates_per_iteration", "learn_in_hipgraph")})
print(d["collector"]); print(d["workload"])
PY
JH_FORCE_DIST=1 timeout 300 python tools/bench_hopper.py --e2e --iters 10 > gpurun_out/r27_hopper_share_dist.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r27_hopper_share_dist.json')); print('force_dist', d['ms_per_iteration'], d['env_transitions_per_s_end_to_end'])"
