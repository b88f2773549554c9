#!/usr/bin/env python3
"""Learner-side measurement at config.ppo.atari's shapes (config/ppo/atari.py:16-54: (4, 84, 84) uint8 frames, discrete policy-value net on the Nature-CNN head,
hidden 512, 8 workers x n_step 128 = 1024 transitions per iteration, minibatch 32, 3 epochs = 96 updates; A = 6, Pong): one PPO iteration = the rollout's
frames uploaded from pinned memory -> no-grad passes over the 2048 frames -> log pi_old, GAE -> 96 minibatch updates (uint8 row gather, convolutional forward,
packed PPO loss, backward, clip + Adam), replayed as one hipGraph (core/agent/ppo_cnn.py).  Atari itself is not installable: frames are synthetic.

--cpu: the same iteration on the host through the CPU port of the reference's learn() (oracle/ppo_port.py with the CNN head, pinned to the reference by the
fixtures ppo_disc_cnn_small / ppo_disc_atari): the `cpu_reference` of the leg.

    python tools/bench_ppo_atari.py [--iters 6] [--cpu]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

S, A, H, W, T, B, E = (4, 84, 84), 6, 512, 8, 128, 32, 3


def rollout(rng, M):
    return {"state": rng.randint(0, 256, size=(M,) + S).astype(np.uint8), "next_state": rng.randint(0, 256, size=(M,) + S).astype(np.uint8),
            "action": rng.randint(0, A, size=(M, 1)), "reward": rng.choice([-1.0, 0.0, 1.0], p=[0.05, 0.9, 0.05], size=(M, 1)), "done": (rng.rand(M, 1) < 1e-3)}


def ppo_atari_leg(iters=6, warmup=3, device="cuda", profile=True):
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent

    M = W * T
    torch.manual_seed(0)
    np.random.seed(0)
    agent = Agent("ppo", state_size=list(S), action_size=A, hidden_size=H, network="discrete_policy_value", head="cnn", optim_config={"name": "adam", "lr": 2.5e-4}, gamma=0.99,
                  batch_size=B, n_step=T, n_epoch=E, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, use_standardization=True, lr_decay=True,
                  run_step=10_000_000, num_workers=W, device=device)
    agent.memory.first_store = False
    cols = rollout(np.random.RandomState(0), M)
    pinned = {}
    for k, v in cols.items():  # the rollout is uploaded every iteration: from pinned host memory
        t = torch.empty(v.shape, dtype=torch.from_numpy(v).dtype, pin_memory=True)
        t.numpy()[...] = v
        pinned[k] = t
        cols[k] = t.numpy()
    step = 0

    def iteration():
        nonlocal step
        step += T
        return agent.process(cols, step)

    for _ in range(warmup):
        r = iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        r = iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    n_upd = E * (M // B)
    kern = {}
    if profile:
        ops.lib_profile(True)
        iteration()
        torch.cuda.synchronize()
        prof = ops.lib_profile_report()
        ops.lib_profile(False)
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]:
            kern[k] = {"launches": v[0], "avg_us": round(v[1] / v[0] * 1e3, 2), "total_ms": round(v[1], 3)}
            if v[2] > 0:
                tf = v[2] / (v[1] * 1e-3) / 1e12
                kern[k].update({"TFLOP/s": round(tf, 1), "frac_of_157.3_f32_mfma_peak": round(tf / 157.3, 3)})
    return {"workload": f"config.ppo.atari shapes, synthetic frames: (4, 84, 84) uint8, A = {A}, hidden {H}, W = {W} x T = {T} = {M} transitions/iteration, minibatch {B}, {E} epochs "
                        f"= {n_upd} updates; learner side only (rollout rows uploaded from pinned memory)",
            "backend": agent.backend, "agent": type(agent).__name__, "learn_in_hipgraph": bool(agent._graph is not None), "ms_per_iteration": dt * 1e3,
            "learner_transitions_per_s": M / dt, "learner_updates_per_s": n_upd / dt, "us_per_update": dt / n_upd * 1e6, "minibatch_updates_per_iteration": n_upd,
            "host_to_device_MB_per_iteration": sum(v.nbytes for v in cols.values()) / 1e6, "last_result": {k: float(v) for k, v in r.items()}, "lib_kernels": kern}


def cpu_reference(threads=None, iters=2):
    """One PPO iteration of the same shapes through the CPU port of the reference (test infrastructure: only bench legs and tests import oracle/)."""
    from oracle.ppo_port import PPOPort

    M = W * T
    best = None
    for th in ([threads] if threads else [8, 16, 32]):
        if th > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(th)
        torch.manual_seed(0)
        np.random.seed(0)
        ag = PPOPort(S, A, H, False, 2.5e-4, 0.99, B, T, E, 0.95, 0.1, 1.0, 0.01, 1.0, run_step=10_000_000)
        cols = rollout(np.random.RandomState(0), M)
        trs = [{k: cols[k][i : i + 1] for k in cols} for i in range(M)]
        t0 = time.perf_counter()
        for i in range(iters):
            ag.process(trs, (i + 1) * T)
        dt = (time.perf_counter() - t0) / iters
        if best is None or dt < best[0]:
            best = (dt, th)
    dt, th = best
    return {"value": M / dt, "unit": "transitions/s", "kind": "port", "threads": th, "s_per_iteration": dt, "learner_updates_per_s": E * (M // B) / dt,
            "sample": f"{iters} PPO iteration(s) of the same shapes through oracle/ppo_port.py (CNN head; pinned to the reference's learn() by fixtures ppo_disc_cnn_small / ppo_disc_atari)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    out = ppo_atari_leg(args.iters, args.warmup)
    if args.cpu:
        out["cpu_reference"] = cpu_reference()
        out["x_cpu_reference"] = out["learner_transitions_per_s"] / out["cpu_reference"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
