#!/bin/bash
# round 6, call 12: the N>1 bench plumbing after the watchdog change + a watchdog that fires
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dp_two_ranks_gpu.py -x -q -k "bench" > gpurun_out/r12_dp_bench.log 2>&1; echo "dp bench rc=$?" >> gpurun_out/r12_dp_bench.log
tail -5 gpurun_out/r12_dp_bench.log
# the watchdog itself: two ranks on the one GPU, Ape-X leg timeout of 1 s -> line printed with the error entry
JH_APEX_DP_TIMEOUT=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-rainbow --no-hopper --no-roofline > gpurun_out/r12_watchdog.log 2>&1; echo "rc=$?" >> gpurun_out/r12_watchdog.log
python - <<'PY'
import json
for l in open("gpurun_out/r12_watchdog.log"):
    if l.startswith("{"):
        d = json.loads(l); print("watchdog line:", d["value"], d["n_gpus"], d.get("apex"), d["legs"]["ppo_env_transitions_s"])
PY
tail -3 gpurun_out/r12_watchdog.log
timeout 900 python bench.py --steps 100 --warmup 20 > gpurun_out/r12_bench1.json 2> gpurun_out/r12_bench1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r12_bench1.json").read().strip().splitlines()[-1]); print(json.dumps(d["legs"]))
PY
