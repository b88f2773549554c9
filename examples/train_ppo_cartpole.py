#!/usr/bin/env python3
"""Train PPO on the synthetic CartPole-v1 with the drop-in agent, the way JORLDY's sync mode does
(run_mode.py:163-207: collect update_period steps from every worker, agent.process, repeat) -- but with the rollout
collected by the native collector (persistent acting kernel) and learn() replayed as one hipGraph.

    python examples/train_ppo_cartpole.py [--iterations 150] [--workers 8]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=150)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    W, T = args.workers, 128
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    agent = Agent("ppo", state_size=4, action_size=2, hidden_size=512, network="discrete_policy_value", optim_config={"name": "adam", "lr": 2.5e-4},
                  gamma=0.99, batch_size=256, n_step=T, n_epoch=3, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0,
                  use_standardization=True, lr_decay=True, run_step=W * T * args.iterations * 3, num_workers=W, device="cuda", seed=args.seed)
    agent.memory.first_store = False
    collector = NativeCollector(ops.CartPoleVec(W, seed=1000 + args.seed), agent, W)
    step, t0 = 0, time.perf_counter()
    for it in range(args.iterations):
        collector.run(T)
        done = agent.memory._store.column("done")[: W * T]
        ep_len = min(500.0, W * T / max(1, int(done.sum().item())))
        step += T
        result = agent.process(None, step)
        if it % 10 == 0 or it == args.iterations - 1:
            dt = time.perf_counter() - t0
            print(f"iter {it:4d}  transitions {(it + 1) * W * T:8d}  mean episode length {ep_len:6.1f}  actor_loss {result['actor_loss']:+.4f}  "
                  f"critic_loss {result['critic_loss']:.3f}  {(it + 1) * W * T / dt / 1e3:7.1f} k transitions/s")


if __name__ == "__main__":
    main()
