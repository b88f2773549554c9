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


def hopper_leg(iters=10, warmup=4, workers=32, batch=2048, e2e=False, dist=None, device="cuda"):
    """One measurement at config.ppo.mujoco shapes -> dict (bench.py embeds it as its `hopper` object).  dist: torch.distributed with
    an initialised process group -> data-parallel learners (one flat all-reduce of the gradient bucket per minibatch)."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent

    S, A, T, W, B, E = 11, 3, 2048, workers, batch, 10
    M = W * T
    rank = dist.get_rank() if dist is not None else 0
    torch.manual_seed(0)
    np.random.seed(rank)
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=512, network="continuous_policy_value", optim_config={"name": "adam", "lr": 3e-4},
                  gamma=0.99, batch_size=B, n_step=T, n_epoch=E, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0,
                  use_standardization=True, lr_decay=True, run_step=1_000_000_000, num_workers=W, device=device)
    agent.memory.first_store = False
    if dist is not None:
        from jorldy_amd.parallel import attach_data_parallel

        attach_data_parallel(agent, dist)
    rng = np.random.RandomState(rank)
    cols = {"state": rng.randn(M, S).astype(np.float32), "action": np.tanh(rng.randn(M, A)).astype(np.float32), "reward": rng.randn(M, 1).astype(np.float32),
            "next_state": rng.randn(M, S).astype(np.float32), "done": (rng.rand(M, 1) < 1e-3)}
    # the learner-side leg uploads this rollout every iteration: from PINNED host memory (round 6; pageable, the 7 MB copy was 0.4-5 ms by box -- VERDICT r5 weak #11)
    if not e2e:
        pinned = {}
        for k, v in cols.items():
            t = torch.empty(v.shape, dtype=torch.from_numpy(v).dtype, pin_memory=True)
            t.numpy()[...] = v
            pinned[k] = t  # keeps the allocation alive
            cols[k] = t.numpy()
    step = 0
    collector = None
    if e2e:
        from jorldy_amd.manager import NativeCollector

        collector = NativeCollector(ops.ControlVec(W, S, A, seed=1 + rank), agent, W)

    def iteration():
        nonlocal step
        step += T
        if collector is not None:
            collector.run(T)
            return agent.process(None, step)
        return agent.process(cols, step)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        r = iteration()
    fence()
    if collector is not None:
        collector.stats()
    t0 = time.perf_counter()
    for _ in range(iters):
        r = iteration()
    fence()
    dt = (time.perf_counter() - t0) / iters
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    world = dist.get_world_size() if dist is not None else 1
    n_upd = E * ((M + B - 1) // B)
    cstats = collector.stats() if collector is not None else None
    kern = {}
    # one more iteration with the library's per-kernel event timers: EVERY rank runs it (data-parallel learners meet in the all-reduce),
    # rank 0 reports
    ops.lib_profile(True)
    iteration()
    torch.cuda.synchronize()
    prof = ops.lib_profile_report()
    ops.lib_profile(False)
    if rank == 0:
        H = 512
        fwd = 2.0 * (n_upd * B + (0 if collector is not None else 2 * M)) * H * H
        nh = 2 * A + 1
        flops = {"jh_gemm16_fwd_h2": fwd, "jh_tgemm_ppo_fwd_h2": fwd, "jh_gemm16_bwd_dW2": 2.0 * n_upd * B * H * H, "jh_gemm16_bwd_dh1": 2.0 * n_upd * B * H * H,
                 "jh_tgemm_ppo_bwd": 2.0 * n_upd * B * H * (2 * H + 2 * A + 1), "jh_tgemm_ppo_bwd_dW1": 2.0 * n_upd * B * H * S,
                 "jh_pmb_fwd": 2.0 * n_upd * B * H * (S + H + nh), "jh_pmb_bwd": 2.0 * n_upd * B * H * (2 * H + 3 * nh + S)}
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]:
            kern[k] = {"launches": v[0], "avg_us": round(v[1] / v[0] * 1e3, 2)}
            if v[2] > 0 or k in flops:  # the grouped GEMM engine declares its own flops per launch; the minibatch kernels are counted here
                tf = (v[2] if v[2] > 0 else flops[k]) / (v[1] * 1e-3) / 1e12
                kern[k].update({"TFLOP/s": round(tf, 1), "frac_of_157.3_f32_mfma_peak": round(tf / 157.3, 3)})
    return {
        "workload": f"config.ppo.mujoco Hopper-shaped (BASELINE.json configs[4]), synthetic: S=11, A=3 continuous, W={W} x T=2048 = {M} transitions/iteration/GPU, "
                    f"minibatch {B}/GPU, 10 epochs" + (", native collector on the synthetic control env" if collector is not None else ", learner side only (rollout rows uploaded)"),
        "n_gpus": world, "backend": agent.backend, "learn_in_hipgraph": bool(agent._graph is not None),
        "ms_per_iteration": dt * 1e3, "learner_transitions_per_s": world * M / dt, "learner_updates_per_s": n_upd / dt, "minibatch_updates_per_iteration": n_upd,
        "host_to_device_MB_per_iteration": sum(v.nbytes for v in cols.values()) / 1e6, "rollout_upload_from": "pinned host memory" if not e2e else None,
        "collector": (dict(kind="NativeCollector on jh_control (synthetic stand-in for MuJoCo Hopper)", **cstats) if collector is not None else None),
        "env_transitions_per_s_end_to_end": (world * M / dt if collector is not None else None),
        "last_result": {k: float(v) for k, v in r.items()}, "lib_kernels": kern}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4, help="untimed iterations: 1 eager, 2 captures the split graphs, 3 eager with pre-drawn lists, 4 captures the whole-learn graph")
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--e2e", action="store_true", help="per-GPU share of configs[4] with the native collector on the synthetic control env (4 workers, minibatch 256)")
    ap.add_argument("--e2e-full", action="store_true", help="configs[4] end to end on ONE GPU: all 32 workers on the native collector, minibatch 2048")
    args = ap.parse_args()
    if args.e2e:
        args.workers, args.batch = 4, 256
    if args.e2e_full:
        args.e2e = True
    print(json.dumps(hopper_leg(args.iters, args.warmup, args.workers, args.batch, args.e2e)))


if __name__ == "__main__":
    main()
