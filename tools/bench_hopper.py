#!/usr/bin/env python3
"""Learner-side measurement at BASELINE.json configs[4] shapes (config.ppo.mujoco, Hopper-v3: S=11, A=3 continuous,
n_step=2048, 32 workers, distributed_batch_size=2048, 10 epochs): one PPO iteration = 65 536 synthetic transitions
(states N(0,1), actions tanh(N(0,1))) -> GAE + standardise -> 10 x 32 minibatch updates of 2048 rows
on the native continuous policy-value net, replayed as one hipGraph.

--e2e: the per-GPU share of the 8-GPU data-parallel layout end to end -- 32 / 8 = 4 workers and 2048 / 8 = 256 minibatch
rows per GPU: the native collector (persistent acting kernel, continuous policy, host sampling) drives the synthetic
control env that stands in for MuJoCo (jh_control_*, S = 11, A = 3), 4 x 2048 transitions per iteration, then 10 epochs x
32 minibatches of 256 rows (five launches each).

    python tools/bench_hopper.py [--iters 10] [--workers 32] [--e2e]
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
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--e2e", action="store_true", help="per-GPU share of configs[4] with the native collector on the synthetic control env (4 workers, minibatch 256)")
    args = ap.parse_args()
    if args.e2e:
        args.workers, args.batch = 4, 256
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent

    S, A, T, W, B, E = 11, 3, 2048, args.workers, args.batch, 10
    M = W * T
    torch.manual_seed(0)
    np.random.seed(0)
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=512, network="continuous_policy_value", optim_config={"name": "adam", "lr": 3e-4},
                  gamma=0.99, batch_size=B, n_step=T, n_epoch=E, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0,
                  use_standardization=True, lr_decay=True, run_step=1_000_000_000, num_workers=W, device="cuda")
    agent.memory.first_store = False
    rng = np.random.RandomState(0)
    cols = {"state": rng.randn(M, S).astype(np.float32), "action": np.tanh(rng.randn(M, A)).astype(np.float32), "reward": rng.randn(M, 1).astype(np.float32),
            "next_state": rng.randn(M, S).astype(np.float32), "done": (rng.rand(M, 1) < 1e-3)}
    step = 0
    collector = None
    if args.e2e:
        from jorldy_amd.manager import NativeCollector

        collector = NativeCollector(ops.ControlVec(W, S, A, seed=1), agent, W)

    def iteration():
        nonlocal step
        step += T
        if collector is not None:
            collector.run(T)
            return agent.process(None, step)
        return agent.process(cols, step)

    for _ in range(args.warmup):
        r = iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        r = iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    n_upd = E * ((M + B - 1) // B)
    ops.lib_profile(True)
    iteration()
    torch.cuda.synchronize()
    prof = ops.lib_profile_report()
    ops.lib_profile(False)
    H = 512
    fwd = 2.0 * (n_upd * B + 2 * M) * H * H
    flops = {"jh_gemm16_fwd_h2": fwd, "jh_tgemm_ppo_fwd_h2": fwd, "jh_gemm16_bwd_dW2": 2.0 * n_upd * B * H * H, "jh_gemm16_bwd_dh1": 2.0 * n_upd * B * H * H,
             "jh_tgemm_ppo_bwd": 2.0 * n_upd * B * H * (2 * H + 2 * A + 1), "jh_tgemm_ppo_bwd_dW1": 2.0 * n_upd * B * H * S}
    kern = {}
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]:
        kern[k] = {"launches": v[0], "avg_us": round(v[1] / v[0] * 1e3, 2)}
        if k in flops:
            tf = flops[k] / (v[1] * 1e-3) / 1e12
            kern[k].update({"TFLOP/s": round(tf, 1), "frac_of_157.3_f32_mfma_peak": round(tf / 157.3, 3)})
    print(json.dumps({
        "workload": f"config.ppo.mujoco Hopper-shaped (BASELINE.json configs[4]), synthetic: S=11, A=3 continuous, W={W} x T=2048 = {M} transitions/iteration, batch {B}, 10 epochs",
        "backend": agent.backend, "learn_in_hipgraph": bool(agent._graph is not None),
        "ms_per_iteration": dt * 1e3, "learner_transitions_per_s": M / dt, "learner_updates_per_s": n_upd / dt, "minibatch_updates_per_iteration": n_upd,
        "host_to_device_MB_per_iteration": sum(v.nbytes for v in cols.values()) / 1e6,
        "collector": (dict(kind="NativeCollector on jh_control (synthetic stand-in for MuJoCo Hopper)", **collector.stats()) if collector is not None else None),
        "env_transitions_per_s_end_to_end": (M / dt if collector is not None else None),
        "last_result": {k: float(v) for k, v in r.items()}, "lib_kernels": kern}))


if __name__ == "__main__":
    main()
