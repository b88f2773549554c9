#!/usr/bin/env python3
"""Where does one batched actor tick go?  N actors at Atari shapes, frame-mode device feed, NO learner running:
wall time per tick, host-side split, and the library's per-kernel times.   python tools/probes/actor_tick_probe.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from jorldy_amd import ops
from jorldy_amd.core.agent import Agent
from jorldy_amd.manager import BatchedValueActors, DeviceActorFeed

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
agent = Agent("ape_x", state_size=[4, 84, 84], action_size=6, hidden_size=512, network="dueling", head="cnn", buffer_size=50000, batch_size=512,
              start_train_step=10**9, n_step=3, num_workers=N, device="cuda")
actors = BatchedValueActors(agent, N)
feed = DeviceActorFeed(actors, agent.memory, 3, 0.99, depth=64, prio_eps=1e-3)
rng = np.random.RandomState(0)
pool = rng.randint(0, 256, size=(257, 84, 84), dtype=np.uint8)
slab = feed.frame_slab
pos = rng.randint(0, 257, size=N)
rew, done = np.zeros((N, 1), np.float32), np.zeros((N, 1), np.float32)


def tick():
    global pos
    slab[:] = pool[pos % 257]
    t0 = time.perf_counter()
    feed.act_frames(None, None, training=True)
    t1 = time.perf_counter()
    feed.push(rew, done)
    t2 = time.perf_counter()
    agent.memory.drain()
    t3 = time.perf_counter()
    pos = pos + 1
    return t1 - t0, t2 - t1, t3 - t2


for _ in range(50):
    tick()
torch.cuda.synchronize()
K = 400
acc = np.zeros(3)
t0 = time.perf_counter()
for _ in range(K):
    acc += tick()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K
print(f"N={N}: {wall * 1e6:.1f} us per tick = {N / wall:.0f} env steps/s; act {acc[0] / K * 1e6:.1f} us, push {acc[1] / K * 1e6:.1f} us, drain {acc[2] / K * 1e6:.1f} us")
ops.lib_profile(True)
for _ in range(20):
    tick()
torch.cuda.synchronize()
prof = ops.lib_profile_report()
ops.lib_profile(False)
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} {v[0] / 20:5.1f} launches/tick  {v[1] / v[0] * 1e3:7.2f} us each")
    tot += v[1] / 20 * 1e3
print(f"  sum of library kernels per tick (event-pair timed, ~2 us each included): {tot:.1f} us")
