#!/usr/bin/env python3
"""Train PPO on the Nature-CNN head (config.ppo.atari's agent: `Agent("ppo", head="cnn", ...)`) on a synthetic image task, the way JORLDY's sync mode does
(run_mode.py:163-207) -- with the vectorised sync collector (`VecCollector`: one batched act() per timestep for all workers, frames stay uint8) and learn()
replayed as one hipGraph on the convolutional engine.

The task ("CueFrames": there is no Atari in this image): one-step episodes of (4, 44, 52) uint8 frames -- noise in [0, 60) with one 8 x 8 block of 110 in one of
the four quadrants; reward 1 for the action that names the quadrant.  Random play: 0.25.

    python examples/train_ppo_cnn_cueframes.py [--iterations 24] [--workers 8]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


class CueFramesVec:
    """W independent envs behind the VecCollector protocol: obs(out) / step(action, next_obs, reward, done); a finished row shows its next frame."""

    state_size, action_size, action_type = (4, 44, 52), 4, "discrete"

    def __init__(self, W, seed):
        self.W, self.rng = W, np.random.RandomState(seed)
        self.frames = np.empty((W,) + self.state_size, np.uint8)
        self.quadrant = np.zeros(W, np.int64)
        for w in range(W):
            self._new(w)

    def _new(self, w):
        c, h, wd = self.state_size
        q = int(self.rng.randint(4))
        f = self.rng.randint(0, 60, size=self.state_size).astype(np.uint8)
        x0 = (2 if q % 2 == 0 else wd // 2 + 2) + int(self.rng.randint(0, wd // 2 - 12))
        y0 = (2 if q // 2 == 0 else h // 2 + 2) + int(self.rng.randint(0, h // 2 - 12))
        f[:, y0 : y0 + 8, x0 : x0 + 8] = 110
        self.frames[w], self.quadrant[w] = f, q

    def obs(self, out=None):
        if out is None:
            return self.frames.copy()
        out[...] = self.frames
        return out

    def step(self, action, next_obs, reward, done):
        a = np.asarray(action).reshape(-1)
        for w in range(self.W):
            reward[w] = 1.0 if int(a[w]) == self.quadrant[w] else 0.0
            done[w] = 1
            self._new(w)
            next_obs[w] = self.frames[w]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=24)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import VecCollector

    W, T = args.workers, 32
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    env = CueFramesVec(W, 1000 + args.seed)
    agent = Agent("ppo", state_size=list(env.state_size), action_size=env.action_size, hidden_size=128, network="discrete_policy_value", head="cnn",
                  optim_config={"name": "adam", "lr": 5e-4}, gamma=0.99, batch_size=32, n_step=T, n_epoch=3, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01,
                  clip_grad_norm=1.0, use_standardization=True, lr_decay=True, run_step=W * T * args.iterations * 3, num_workers=W, device="cuda", seed=args.seed)
    agent.memory.first_store = False
    collector = VecCollector(env, agent)
    step, t0 = 0, time.perf_counter()
    for it in range(args.iterations):
        transitions, _ = collector.run(T)
        step += T
        result = agent.process(transitions, step)
        if it % 4 == 0 or it == args.iterations - 1:
            dt = time.perf_counter() - t0
            print(f"iter {it:3d}  transitions {(it + 1) * W * T:6d}  mean reward {float(transitions['reward'].mean()):.3f} (random play 0.25)  "
                  f"actor_loss {result['actor_loss']:+.4f}  critic_loss {result['critic_loss']:.3f}  {(it + 1) * W * T / dt / 1e3:6.1f} k transitions/s")


if __name__ == "__main__":
    main()
