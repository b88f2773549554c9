#!/usr/bin/env python3
"""Dump the outputs of the multi-workgroup PPO loss (B > 1024) and of one tiled PPO update (B = 2048, Hopper widths) for the library that is
in place, so that two libraries can be compared bit for bit:   python tools/probes/r05_loss_bits.py out.npz"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from jorldy_amd import ops

out = {}
dev = "cuda"
for cont in (False, True):
    for B, A in ((1025, 3), (2048, 3), (5000, 6), (2048, 1), (40000, 4)):
        rng = np.random.RandomState(B + A + 7 * cont)
        M = B + 50
        idx = torch.as_tensor(rng.permutation(M)[:B]).to(dev)
        f = lambda a: torch.as_tensor(np.asarray(a, np.float32)).to(dev)
        adv, ret, vold = f(rng.randn(M, 1)), f(rng.randn(M, 1)), f(rng.randn(M, 1))
        vp = f(rng.randn(B, 1))
        if cont:
            act = f(np.tanh(rng.randn(M, A)))
            mu, ls = f(rng.randn(B, A)), f(rng.randn(B, A))
            lpo = f(-1.0 + 0.3 * rng.randn(M, A))
            r = ops.ppo_loss_continuous(mu, ls, vp, idx, act, adv, ret, vold, lpo, 0.2, 0.5, 0.01)
        else:
            if A < 2:
                continue
            act = f(rng.randint(0, A, size=(M, 1)))
            z = f(2 * rng.randn(B, A))
            lpo = f(-1.0 + 0.3 * rng.randn(M, 1))
            r = ops.ppo_loss_discrete(z, vp, idx, act, adv, ret, vold, lpo, 0.2, 0.5, 0.01)
        torch.cuda.synchronize()
        for j, t in enumerate(r):
            out[f"loss_c{int(cont)}_B{B}_A{A}_{j}"] = t.cpu().numpy()

# one learn() at Hopper widths (tiled path, 2048-row minibatches): final weights
from jorldy_amd.core.agent import Agent

torch.manual_seed(0)
np.random.seed(0)
S, A, T, W, B = 11, 3, 512, 8, 2048
agent = Agent("ppo", state_size=S, action_size=A, hidden_size=512, network="continuous_policy_value", optim_config={"name": "adam", "lr": 3e-4}, gamma=0.99,
              batch_size=B, n_step=T, n_epoch=2, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, use_standardization=True,
              lr_decay=False, run_step=10**9, num_workers=W, device=dev)
agent.memory.first_store = False
rng = np.random.RandomState(3)
M = W * T
cols = {"state": rng.randn(M, S).astype(np.float32), "action": np.tanh(rng.randn(M, A)).astype(np.float32), "reward": rng.randn(M, 1).astype(np.float32),
        "next_state": rng.randn(M, S).astype(np.float32), "done": (rng.rand(M, 1) < 1e-3)}
res = agent.process(cols, T)
torch.cuda.synchronize()
for k, v in agent.network.state_dict().items():
    out["w_" + k] = v.detach().cpu().numpy()
for k, v in (res or {}).items():
    out["res_" + k] = np.asarray(v)
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
