#!/usr/bin/env python3
"""The REAL reference vs the port on the same host cores (build container only: needs /root/reference; no GPU).  Test infrastructure like the rest of oracle/: not imported by the product.

bench.py's `cpu_baseline` times oracle/ppo_port.py (kind "port") because the reference tree does not exist on the GPU box.
This script shows what that stands for: PPO.learn of the unmodified reference (config.ppo.cartpole: 1024 transitions, 3 epochs x 4
minibatches of 256, hidden 512) and PPOPort.process on the same transitions, same torch thread count, alternating; and
Rainbow.learn (config.rainbow.atari shapes) vs RainbowPort.learn.  and DQN.learn (config.dqn.cartpole) vs DQNPort.learn.  -> JSON (committed as profiles/r05_cpu_reference_vs_port.json; round 3's: r03_...).

    python oracle/time_reference_vs_port.py [--threads 8] [--iters 10]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    import torch

    from oracle import ppo_port as P
    from oracle import synth
    from oracle.rainbow_port import RainbowPort

    torch.set_num_threads(args.threads)
    scratch = tempfile.mkdtemp(prefix="jref_")
    subprocess.check_call(f"cd {args.ref} && tar --exclude='jorldy/core/env/mlagents' -cf - jorldy | (cd {scratch} && tar xf -)", shell=True)
    cwd = os.getcwd()
    os.chdir(os.path.join(scratch, "jorldy"))
    sys.path.insert(0, os.getcwd())
    sys.dont_write_bytecode = True
    try:
        from core.agent.ppo import PPO
        from core.agent.rainbow import Rainbow

        W, T, S, A = 8, 128, 4, 2
        trs = synth.ppo_rollout(np.random.RandomState(5), W * T, S, A, False, clamp_every=0)
        ref = PPO(state_size=S, action_size=A, hidden_size=512, network="discrete_policy_value", optim_config={"name": "adam", "lr": 2.5e-4}, batch_size=256,
                  n_step=T, n_epoch=3, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, gamma=0.99, run_step=100000, num_workers=W, device="cpu")
        ref.memory.first_store = False
        port = P.PPOPort(S, A, 512, False, 2.5e-4, 0.99, 256, T, 3, 0.95, 0.1, 1.0, 0.01, 1.0, run_step=100000)
        t_ref, t_port = [], []
        step = 0
        for it in range(args.iters + 2):
            step += T
            t0 = time.perf_counter()
            ref.process([dict(t) for t in trs], step)
            t1 = time.perf_counter()
            port.process([dict(t) for t in trs], step)
            t2 = time.perf_counter()
            if it >= 2:
                t_ref.append(t1 - t0)
                t_port.append(t2 - t1)
        out = {"ppo_cartpole_process_1024_transitions": {"reference_ms": float(np.median(t_ref)) * 1e3, "port_ms": float(np.median(t_port)) * 1e3,
                                                         "port_over_reference": float(np.median(t_port) / np.median(t_ref))}}
        # Rainbow learner at Atari shapes
        kw = dict(state_size=(4, 84, 84), action_size=4, hidden_size=512, head="cnn", optim_config={"name": "adam", "lr": 6.25e-5}, gamma=0.99, buffer_size=4096,
                  batch_size=32, start_train_step=0, target_update_period=10000, run_step=100000, n_step=3, alpha=0.5, beta=0.4, learn_period=1,
                  uniform_sample_prob=1e-3, v_min=-1, v_max=10, num_support=51, device="cpu")
        rb = Rainbow(**kw)
        rb.memory.first_store = False
        pt = RainbowPort((4, 84, 84), 4, 512, buffer_size=4096, batch_size=32, n_step=3)
        rng = np.random.RandomState(0)
        rows = [{"state": rng.randint(0, 256, size=(1, 4, 84, 84), dtype=np.uint8), "action": rng.randint(0, 4, size=(1, 1)),
                 "reward": rng.choice([-1.0, 0.0, 1.0], size=(1, 3, 1)), "next_state": rng.randint(0, 256, size=(1, 4, 84, 84), dtype=np.uint8),
                 "done": rng.rand(1, 3, 1) < 1e-3} for _ in range(256)]
        rb.memory.store([dict(r) for r in rows])
        pt.memory.store([dict(r) for r in rows])
        t_ref, t_port = [], []
        for it in range(args.iters + 2):
            t0 = time.perf_counter()
            rb.learn()
            t1 = time.perf_counter()
            pt.learn()
            t2 = time.perf_counter()
            if it >= 2:
                t_ref.append(t1 - t0)
                t_port.append(t2 - t1)
        out["rainbow_atari_learn_B32"] = {"reference_ms": float(np.median(t_ref)) * 1e3, "port_ms": float(np.median(t_port)) * 1e3,
                                          "port_over_reference": float(np.median(t_port) / np.median(t_ref))}
        # DQN at config.dqn.cartpole (BASELINE configs[0]): one learn() of the reference vs oracle/dqn_port.py on the same 256 stored transitions
        from core.agent.dqn import DQN

        from oracle.dqn_port import DQNPort

        dq = DQN(state_size=4, action_size=2, hidden_size=512, optim_config={"name": "adam", "lr": 1e-4}, gamma=0.99, buffer_size=50000, batch_size=32,
                 start_train_step=0, target_update_period=500, run_step=100000, device="cpu")
        dq.memory.first_store = False
        dp = DQNPort(4, 2, 512, lr=1e-4, batch_size=32, start_train_step=0)
        rows = [synth.raw_transition(rng, 4, 2) for _ in range(256)]
        dq.memory.store([dict(r) for r in rows])
        dp.memory.store([dict(r) for r in rows])
        t_ref, t_port = [], []
        for it in range(20 * args.iters + 20):
            t0 = time.perf_counter()
            dq.learn()
            t1 = time.perf_counter()
            dp.learn()
            t2 = time.perf_counter()
            if it >= 20:
                t_ref.append(t1 - t0)
                t_port.append(t2 - t1)
        out["dqn_cartpole_learn_B32"] = {"reference_ms": float(np.median(t_ref)) * 1e3, "port_ms": float(np.median(t_port)) * 1e3,
                                         "port_over_reference": float(np.median(t_port) / np.median(t_ref))}
        out["threads"], out["host"] = args.threads, f"{os.cpu_count()} logical cores (build container)"
        out["note"] = "same process, same torch thread count, alternating calls, medians; the port is what bench.py's cpu_baseline times on the GPU box"
        print(json.dumps(out, indent=1))
    finally:
        os.chdir(cwd)
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
