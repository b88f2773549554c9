#!/usr/bin/env python3
"""Where does an acting exchange of config.ppo.mujoco's 32 workers go?  20 rollouts of 256 steps with JH_PERSIST_DEBUG / JH_COLLECT_DEBUG set by the caller."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jorldy_amd import ops
from jorldy_amd.core.agent import Agent
from jorldy_amd.manager import NativeCollector
W, S, A, T = int(os.environ.get("W", 32)), 11, 3, 256
agent = Agent("ppo", state_size=S, action_size=A, hidden_size=512, network="continuous_policy_value", optim_config={"name": "adam", "lr": 3e-4}, batch_size=2048, n_step=T, n_epoch=1,
              num_workers=W, device="cuda", run_step=10**9)
agent.memory.first_store = False
col = NativeCollector(ops.ControlVec(W, S, A, seed=1), agent, W)
step = 0
for it in range(33):
    col.run(T); step += T
    agent.process(None, step)
torch.cuda.synchronize()
print(col.stats())
