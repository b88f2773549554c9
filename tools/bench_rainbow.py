#!/usr/bin/env python3
"""Secondary measurement (BASELINE.json configs[2], config.rainbow.atari --env.name breakout, single
mode): learner-side hot path at Atari shapes with synthetic transitions:
uint8 (4,84,84) frames, A=4, n_step=3, K=51, B=32, PER alpha .5, buffer N slots.

Per env step: PERBuffer.store of one transition; every `learn_period`=4 steps one Rainbow.learn()
(PER sample -> gather -> 3 CNN forwards + backward (jh_rbnet_*, or torch/MIOpen with --backend torch) -> jh_c51_loss ->
priority write-back -> Adam).  Reports learner updates/s and the implied env-steps/s ceiling
(learn_period x updates/s), next to the same loop on the CPU reference port when --cpu is given.

    python tools/bench_rainbow.py [--buffer 100000] [--updates 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffer", type=int, default=100000)
    ap.add_argument("--updates", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--backend", default="native", choices=["native", "torch"])
    args = ap.parse_args()
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(0)
    np.random.seed(0)
    N, B, n = args.buffer, 32, 3
    agent = Agent("rainbow", state_size=[4, 84, 84], action_size=4, hidden_size=512, head="cnn", optim_config={"name": "adam", "lr": 6.25e-5},
                  gamma=0.99, buffer_size=N, batch_size=B, start_train_step=0, target_update_period=10000, run_step=30_000_000, n_step=n,
                  alpha=0.5, beta=0.4, learn_period=4, uniform_sample_prob=1e-3, v_min=-1, v_max=10, num_support=51, device="cuda", backend=args.backend)
    agent.memory.first_store = False
    rng = np.random.RandomState(0)
    # fill the buffer with synthetic n-step transitions in chunks (SoA fast path)
    chunk = 2048
    t0 = time.perf_counter()
    filled = 0
    while filled < min(N, 20000):
        m = min(chunk, N - filled)
        cols = {
            "state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8),
            "action": rng.randint(0, 4, size=(m, 1)),
            "reward": rng.choice([-1.0, 0.0, 1.0], p=[0.02, 0.9, 0.08], size=(m, n, 1)).astype(np.float32),
            "next_state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8),
            "done": (rng.rand(m, n, 1) < 1e-3),
        }
        agent.memory.store_soa(cols)
        filled += m
    torch.cuda.synchronize()
    fill_s = time.perf_counter() - t0
    # non-trivial priorities
    idx = torch.arange(agent.memory.first_leaf_index, agent.memory.first_leaf_index + filled, device="cuda")
    for o in range(0, filled, 2048):
        agent.memory.update_priorities(idx[o : o + 2048], torch.rand(min(2048, filled - o), device="cuda") ** 0.5)
    one = {k: v[:1] for k, v in cols.items()}
    step = 0

    def env_steps_and_learn():
        nonlocal step
        for _ in range(4):  # learn_period env steps: one store each (rainbow.py:255-262)
            step += 1
            agent.memory.store_soa(one)
        return agent.learn()

    for _ in range(args.warmup):
        env_steps_and_learn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.updates):
        r = env_steps_and_learn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    graphed = agent._graph is not None
    # per-kernel HIP-event timing needs eager launches: separate short pass, not part of `dt`
    ops.lib_profile(True)
    for _ in range(20):
        env_steps_and_learn()
    torch.cuda.synchronize()
    prof = ops.lib_profile_report()
    ops.lib_profile(False)
    t0 = time.perf_counter()
    for _ in range(40):
        agent.learn()
    torch.cuda.synchronize()
    dt_learn = (time.perf_counter() - t0) / 40
    out = {
        "workload": f"config.rainbow.atari breakout-shaped, synthetic uint8 (4,84,84), B=32, n=3, K=51, PER N={N} ({filled} filled), network backend {args.backend}",
        "backend": args.backend,
        "learner_updates_per_s": args.updates / dt,
        "env_steps_per_s_ceiling": 4 * args.updates / dt,
        "ms_per_update_incl_4_stores": dt / args.updates * 1e3,
        "ms_per_learn_only": dt_learn * 1e3,
        "learn_in_hipgraph": graphed,
        "fill_MB_per_s": filled * 2 * 28224 / fill_s / 1e6,
        "last_result": {k: float(v) for k, v in r.items()},
        "lib_kernel_avg_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
        "reference_cpu_updates_per_s_survey": 34.0,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
