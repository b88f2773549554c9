#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for fr in 256 512 1024 2048; do
python - <<PY
import sys, time, json
sys.path.insert(0, "tools")
import numpy as np, torch
import bench_ppo_atari as b
from jorldy_amd.core.agent import Agent
S, A, H, W, T, B, E = b.S, b.A, b.H, b.W, b.T, b.B, b.E
M = W * T
torch.manual_seed(0); np.random.seed(0)
agent = Agent("ppo", state_size=list(S), action_size=A, hidden_size=H, network="discrete_policy_value", head="cnn", optim_config={"name": "adam", "lr": 2.5e-4}, batch_size=B, n_step=T, n_epoch=E,
              run_step=10_000_000, num_workers=W, device="cuda", forward_rows=$fr)
agent.memory.first_store = False
cols = b.rollout(np.random.RandomState(0), M)
step = 0
def it():
    global step
    step += T
    return agent.process(cols, step)
for _ in range(3): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6): it()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
print("forward_rows", $fr, "ms/iter", round(dt * 1e3, 3), "transitions/s", round(M / dt))
PY
done
