#!/usr/bin/env python3
"""Secondary measurement (BASELINE.json configs[2], config.rainbow.atari --env.name breakout, single
mode): learner-side hot path at Atari shapes with synthetic transitions:
uint8 (4,84,84) frames, A=4, n_step=3, K=51, B=32, PER alpha .5, buffer N slots.

Per env step: PERBuffer.store of one transition; every `learn_period`=4 steps one Rainbow.learn()
(PER sample -> gather -> 3 CNN forwards + backward (jh_rbnet_*) -> jh_c51_loss ->
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
    ap.add_argument("--stream", action="store_true", help="frame-stack stream (consecutive transitions share frames, as the Atari wrapper produces) instead of i.i.d. stacks")
    ap.add_argument("--frame-dedup", action="store_true", help="store single frames + slot numbers (implies --stream)")
    args = ap.parse_args()
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(0)
    np.random.seed(0)
    N, B, n = args.buffer, 32, 3
    agent = Agent("rainbow", state_size=[4, 84, 84], action_size=4, hidden_size=512, head="cnn", optim_config={"name": "adam", "lr": 6.25e-5},
                  gamma=0.99, buffer_size=N, batch_size=B, start_train_step=0, target_update_period=10000, run_step=30_000_000, n_step=n,
                  alpha=0.5, beta=0.4, learn_period=4, uniform_sample_prob=1e-3, v_min=-1, v_max=10, num_support=51, device="cuda",
                  frame_dedup=args.frame_dedup)
    agent.memory.first_store = False
    rng = np.random.RandomState(0)
    # fill the buffer with synthetic n-step transitions in chunks (SoA fast path)
    chunk = 2048
    t0 = time.perf_counter()
    filled = 0
    stream = args.stream or args.frame_dedup
    t_next = 0
    seq = rng.randint(0, 256, size=(20480 + 4096 + 8, 84, 84), dtype=np.uint8) if stream else None  # one long episode of frames

    def stacks(t0, m):
        """state_t = frames t..t+3, next_state_t = frames t+n..t+n+3 (4-frame stack, n-step assembler)"""
        win = np.lib.stride_tricks.sliding_window_view(seq, 4, axis=0)  # [T-3, 84, 84, 4]
        st = np.ascontiguousarray(np.moveaxis(win[t0 : t0 + m], -1, 1))
        ns = np.ascontiguousarray(np.moveaxis(win[t0 + n : t0 + n + m], -1, 1))
        return st, ns

    while filled < min(N, 20000):
        m = min(chunk, N - filled)
        if stream:
            st, ns = stacks(t_next, m)
            t_next += m
        else:
            st, ns = rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8), rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8)
        cols = {
            "state": st,
            "action": rng.randint(0, 4, size=(m, 1)),
            "reward": rng.choice([-1.0, 0.0, 1.0], p=[0.02, 0.9, 0.08], size=(m, n, 1)).astype(np.float32),
            "next_state": ns,
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
        nonlocal step, t_next
        for _ in range(4):  # learn_period env steps: one store each (rainbow.py:255-262)
            step += 1
            if stream and t_next + n + 5 < len(seq):
                one["state"], one["next_state"] = stacks(t_next, 1)
                t_next += 1
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
        "workload": f"config.rainbow.atari breakout-shaped, synthetic uint8 (4,84,84), B=32, n=3, K=51, PER N={N} ({filled} filled)",
        "backend": "native",
        "stream": stream,
        "frame_dedup": agent.memory._frames.stats() if agent.memory._frames is not None else None,
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
