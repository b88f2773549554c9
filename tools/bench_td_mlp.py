#!/usr/bin/env python3
"""learn() latency of the TD-family agents at CartPole shapes (BASELINE.json configs[0]: S=4, A=2, hidden 512, B=32;
config/dqn/cartpole.py): q-network / dueling / noisy categorical net, backward and optimizer on libjorldy_hip (jh_rbnet_*), learn()
replayed as one hipGraph.
Prints microseconds per learn().    python tools/bench_td_mlp.py"""
import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jorldy_amd.core.agent import Agent
out = {}
for name, extra in (("dqn", {}), ("per", dict(learn_period=1)), ("ape_x", dict(num_workers=8, n_step=3)), ("rainbow", dict(n_step=3))):
    for backend in ("native",):
        torch.manual_seed(0); np.random.seed(0)
        kw = dict(state_size=4, action_size=2, hidden_size=512, optim_config={"name": "adam", "lr": 1e-4}, buffer_size=50000, batch_size=32, start_train_step=0,
                  run_step=10**7, device="cuda")
        kw.update(extra)
        a = Agent(name, **kw)
        a.memory.first_store = False
        rng = np.random.RandomState(0)
        n = 4096
        nst = extra.get("n_step")
        rshape = (n, nst, 1) if nst else (n, 1)
        cols = {"state": rng.randn(n, 4).astype(np.float32), "action": rng.randint(0, 2, size=(n, 1)), "reward": rng.randn(*rshape).astype(np.float32),
                "next_state": rng.randn(n, 4).astype(np.float32), "done": rng.rand(*rshape) < 0.05}
        a.memory.store_soa(cols)
        for _ in range(30):
            a.learn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            a.learn()
        torch.cuda.synchronize()
        out[f"{name}/{backend}"] = round((time.perf_counter() - t0) / 400 * 1e6, 1)
print(json.dumps(out))
