#!/usr/bin/env python3
"""Benchmark of the PPO sync hot path (BASELINE.json configs[1]: config.ppo.cartpole --sync
--train.num_workers 8 on 1 x MI355X).

One "step" = one loop body of sync_distributed_train (run_mode.py:180-186):
    collect W x T transitions (batched GPU acting + native host CartPole)  -> GPU rollout store
    agent.process: log pi_old, GAE, n_epoch x minibatch clipped-loss updates (HIP kernels)
`value` = env transitions per second over the whole job (W*T*K*N / wall), measured between
barrier + torch.cuda.synchronize() brackets, max over ranks.

    python bench.py --gpus N --steps K --warmup W
(for N>1 the driver launches it with torch.distributed.run, one rank per GPU; ranks are
data-parallel learners: own envs, own minibatches, one RCCL all-reduce of the flat gradient per
minibatch -> "weak" scaling, global batch = N x 256.)
"""
import argparse
import json
import os
import sys
import time

# kernel arguments (and with them the per-step observations of the acting kernel) are written by the
# host straight into VRAM instead of being fetched over PCIe at kernel start; must be set before the
# HIP runtime initialises
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workers", type=int, default=8, help="sync workers per GPU (config: train.num_workers 8)")
    ap.add_argument("--cpu-baseline-iters", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--python-collector", action="store_true", help="per-timestep Python loop instead of jh_collector_run")
    ap.add_argument("--no-rainbow", action="store_true", help="skip the Rainbow (configs[2]) learner leg")
    ap.add_argument("--rainbow-updates", type=int, default=300)
    return ap.parse_args()


def cpu_baseline(iters, W, T):
    """The reference's CPU path (port: oracle/ppo_port.py, pinned bit-for-bit against the reference)
    timed on this box's host cores: W in-process workers doing B=1 acting on the synthetic CartPole,
    then PPO.learn with torch CPU using all cores."""
    from oracle import ppo_port as P

    # torch CPU with one thread per core is pathological on a 256-core host for these tiny GEMMs
    # (61 s per learn() measured); 8 threads is what the reference's own box used
    cores = min(8, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    np.random.seed(0)
    torch.manual_seed(0)
    agent = P.PPOPort(4, 2, 512, False, 2.5e-4, 0.99, 256, T, 3, 0.95, 0.1, 1.0, 0.01, 1.0, run_step=100000)
    envs = [P._OneEnv(seed=w) for w in range(W)]
    states = [e.reset_obs() for e in envs]
    step = 0
    for _ in range(2):  # warm-up
        trs = P.sync_iteration(agent, envs, states, T)
        step += T
        agent.process(trs, step)
    t0 = time.perf_counter()
    n_tr = 0
    t_collect = 0.0
    for _ in range(iters):
        c0 = time.perf_counter()
        trs = P.sync_iteration(agent, envs, states, T)
        t_collect += time.perf_counter() - c0
        step += T
        agent.process(trs, step)
        n_tr += len(trs)
    dt = time.perf_counter() - t0
    return {
        "value": n_tr / dt,
        "unit": "env_transitions/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{iters} sync iterations of config.ppo.cartpole (W={W}, T={T}, 3 epochs x 4 minibatches of 256), "
                  f"workers run sequentially in-process (no Ray), learner on {cores} torch threads; "
                  f"collect {t_collect / iters * 1e3:.1f} ms + learn {(dt - t_collect) / iters * 1e3:.1f} ms per iteration",
        "learner_updates_per_s": iters * 12 / (dt - t_collect),
    }


# rocprofv3 kernel name of each library-profiler label (template arguments of jh_gemm16_kernel)
_PMC_NAME = {
    "jh_gemm16_bwd_dh1": "jh_gemm16_kernel<0, false, 1,", "jh_gemm16_bwd_dW2": "jh_gemm16_kernel<1, false, 2,",
    "jh_gemm16_bwd_dW1": "jh_gemm16_kernel<1, false, 2,", "jh_gemm16_fwd_h2": "jh_gemm16_kernel<0, true, 0,",
    "jh_gemm16_bwd_dWheads": "jh_gemm16_kernel<1, false, 3,", "jh_adam_kernel": "jh_adam_kernel", "jh_gae_kernel": "jh_gae_kernel",
    "jh_ppo_fused_kernel<CONT>": "jh_ppo_fused_kernel<false>", "jh_gather_kernel": "jh_gather_kernel",
}


def rainbow_leg(rank, world, local_rank, dist, updates, warmup):
    """Second half of BASELINE.json's metric: learner updates/s of Rainbow at config.rainbow.atari shapes
    (configs[2]; uint8 (4,84,84) frames, A=4, B=32 per GPU, n=3, K=51, PER), synthetic transitions.  One env
    step = one PERBuffer.store, one learn() per 4 env steps (learn_period); every rank is a learner with its
    own replay shard, gradients averaged with one RCCL all-reduce per learn() (weak scaling)."""
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.parallel import attach_data_parallel

    N, B, n, filled = 100_000, 32, 3, 8192
    torch.manual_seed(4321)
    agent = Agent("rainbow", state_size=[4, 84, 84], action_size=4, hidden_size=512, head="cnn", optim_config={"name": "adam", "lr": 6.25e-5},
                  gamma=0.99, buffer_size=N, batch_size=B, start_train_step=0, target_update_period=10000, run_step=30_000_000, n_step=n,
                  alpha=0.5, beta=0.4, learn_period=4, uniform_sample_prob=1e-3, v_min=-1, v_max=10, num_support=51, device=f"cuda:{local_rank}")
    agent.memory.first_store = False
    rng = np.random.RandomState(100 + rank)
    for o in range(0, filled, 2048):
        m = 2048
        cols = {"state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8), "action": rng.randint(0, 4, size=(m, 1)),
                "reward": rng.choice([-1.0, 0.0, 1.0], p=[0.02, 0.9, 0.08], size=(m, n, 1)).astype(np.float32),
                "next_state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8), "done": (rng.rand(m, n, 1) < 1e-3)}
        agent.memory.store_soa(cols)
    idx = torch.arange(agent.memory.first_leaf_index, agent.memory.first_leaf_index + filled, device="cuda")
    for o in range(0, filled, 2048):
        agent.memory.update_priorities(idx[o : o + 2048], torch.rand(2048, device="cuda") ** 0.5)
    if dist is not None:
        attach_data_parallel(agent, dist)
    one = {k: v[:1] for k, v in cols.items()}
    np.random.seed(99 + rank)

    def update():
        for _ in range(4):
            agent.memory.store_soa(one)
        return agent.learn()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        update()
    fence()
    t0 = time.perf_counter()
    for _ in range(updates):
        r = update()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"metric": "learner_updates_per_s (Rainbow, config.rainbow.atari shapes, B=32 per GPU)", "value": world * updates / dt, "unit": "updates/s",
            "env_steps_per_s": 4 * world * updates / dt, "ms_per_update_incl_4_stores": dt / updates * 1e3, "updates": updates, "n_gpus": world,
            "scaling": "weak", "backend": agent.backend, "hipgraph": bool(agent._graph is not None), "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config.rainbow.atari breakout-shaped (BASELINE.json configs[2]): uint8 (4,84,84) frames, A=4, B=32, n=3, K=51, PER "
                                   f"N={N} ({filled} filled), one store per env step, one learn() per 4", "parallelism": f"dp{world}"},
            "loss": float(r["loss"]), "cpu_reference_updates_per_s": 34.0, "cpu_reference_note": "BASELINE.md §2: reference Rainbow.learn on 8 host cores (survey box)"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_bench_ppo_cartpole.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs).
    gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-byte requests at 64 bytes ->
    doubled.  Counter unit: KiB.  None when the summary is not available."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_bench_ppo_cartpole.json")
    key = _PMC_NAME.get(kernel)
    if key is None or not os.path.exists(path):
        return None
    for name, v in json.load(open(path)).items():
        if key in name and "FETCH_SIZE_KB_mean" in v and "WRITE_SIZE_KB_mean" in v:
            return (2.0 * v["FETCH_SIZE_KB_mean"] + v["WRITE_SIZE_KB_mean"]) * 1024.0
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("JH_FORCE_DIST") == "1"  # exercise the DP code path on a single rank (testing)
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector, VecCollector
    from jorldy_amd.parallel import make_grad_sync, pin_to_gpu_node

    # "actors pinned to host cores": every rank's collector thread on the cores next to ITS GPU (two PCIe crossings per
    # timestep; the far socket costs +40 % per step); ranks sharing a NUMA node (4 GPUs per socket) split its cores
    cores = pin_to_gpu_node(local_rank, local_rank=local_rank % 4, ranks_on_node=min(4, world))

    W, T = args.workers, 128
    np.random.seed(1234 + rank)
    torch.manual_seed(1234)  # identical initial weights on every rank
    agent = Agent("ppo", state_size=4, action_size=2, hidden_size=512, network="discrete_policy_value",
                  optim_config={"name": "adam", "lr": 2.5e-4}, gamma=0.99, batch_size=256, n_step=T, n_epoch=3, _lambda=0.95,
                  epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, use_standardization=True, lr_decay=True,
                  run_step=10_000_000, num_workers=W, device=f"cuda:{local_rank}")
    agent.memory.first_store = False
    if dist is not None:
        agent.grad_sync = make_grad_sync(agent.network, dist)
    env = ops.CartPoleVec(W, seed=100 + rank)
    collector = (VecCollector if args.python_collector or agent.backend != "native" else NativeCollector)(env, agent, W)

    step = 0

    def one_iteration():
        nonlocal step
        transitions, _ = collector.run(T)
        step += T
        result = agent.process(transitions, step)
        collector.sync(None)
        return result

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_iteration()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = one_iteration()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_mb = (W * T + 255) // 256
    n_updates = 3 * n_mb
    out = {
        "metric": "env_steps_per_s (PPO CartPole sync, W=8 workers/GPU, T=128)",
        "value": world * W * T * args.steps / dt,
        "unit": "env_transitions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "config.ppo.cartpole --sync --train.num_workers 8 (BASELINE.json configs[1]): synthetic CartPole-v1, "
                               "W=8 x T=128 = 1024 transitions/iteration/GPU, MLP 4-512-512-{2,1}, 3 epochs x 4 minibatches of 256",
                   "workers_per_gpu": W, "n_step": T, "batch_size": 256, "n_epoch": 3, "parallelism": f"dp{world}",
                   "backend": agent.backend, "hipgraph": bool(agent._graph is not None), "collector": type(collector).__name__,
                   "host_cores_per_rank": len(cores) if cores else None},
        "learner_updates_per_s": world * n_updates * args.steps / dt,
        "last_result": {k: float(v) for k, v in result.items()},
    }
    if hasattr(collector, "stats"):
        out["collector_host_us_per_timestep"] = collector.stats()

    # ---- roofline of the dominant hand-written kernel -------------------------------------------------
    # Separate pass after the timed region (the timed region replays one hipGraph per learn(), which
    # cannot be bracketed per kernel): the SAME learn() work is enqueued eagerly with a HIP event pair
    # around every kernel launch, recorded inside libjorldy_hip on the launch stream, behind a few
    # graph replays so the queue is full and an event pair measures the kernel, not host launch gaps.
    if rank == 0 and agent.backend == "native" and not args.no_roofline:
        H, S, A, Bm, M = 512, 4, 2, 256, W * T
        prof = {}
        for _ in range(3):
            transitions, _ = collector.run(T)
            step += T
            if agent._graph is not None:
                for _ in range(6):
                    agent._graph.replay()
            ops.lib_profile(True)
            ops.lib_profile_calibrate(64)
            agent.process(transitions, step)
            prof_part = ops.lib_profile_report()
            ops.lib_profile(False)
            for k, v in prof_part.items():
                prof[k] = (prof.get(k, (0, 0.0))[0] + v[0], prof.get(k, (0, 0.0))[1] + v[1])
        rows = n_updates * Bm + 2 * M  # rows through the forward GEMM per learn()
        # algorithmic work per learn() (DESIGN.md "kernels"): flops for the MFMA GEMMs, bytes otherwise
        work = {
            "jh_gemm16_fwd_h2": ("mfma", 2.0 * rows * H * H),
            "jh_gemm16_bwd_dW2": ("mfma", 2.0 * n_updates * Bm * H * H),
            "jh_gemm16_bwd_dh1": ("mfma", 2.0 * n_updates * Bm * H * H),
            "jh_gae_kernel": ("hbm", 24.0 * M),
            "jh_ppo_fused_kernel<CONT>": ("hbm", n_updates * (4.0 * Bm * (2 * A + 7) + 8 * Bm)),
            "jh_adam_kernel": ("hbm", n_updates * 4.0 * 266755 * 7),
            "jh_gradnorm_kernel": ("hbm", n_updates * 4.0 * 266755),
            "jh_gather_kernel": ("hbm", M * (44.0 + 44.0 - 7 - 3)),
        }
        # every event pair carries a fixed recording overhead (~2 us on MI355X): measured with empty pairs in
        # the same passes and subtracted, so the figures agree with rocprofv3's kernel durations
        n_cal, ms_cal = prof.pop("__event_pair_overhead", (1, 0.0))
        ovh_ms = ms_cal / n_cal
        prof = {k: (v[0], max(v[1] - v[0] * ovh_ms, 1e-6)) for k, v in prof.items()}
        name, (n_launch, ms_total) = max(prof.items(), key=lambda kv: kv[1][1])
        bound, per_learn = work.get(name, ("hbm", 0.0))
        n_learn = 3
        avg_s = ms_total / n_launch * 1e-3
        per_launch = per_learn * n_learn / n_launch
        if bound == "mfma":
            achieved, peak, unit = per_launch / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
        out["roofline"] = {"kernel": name, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                           "traffic": pmc_traffic(name), "launches": n_launch, "avg_us": avg_s * 1e6, "algorithmic_work_per_launch": per_launch, "event_pair_overhead_us_subtracted": ovh_ms * 1e3,
                           "note": "latency-bound BASELINE shape (minibatch 256 x hidden 512); see DESIGN.md for scaled shapes"}
        out["kernel_avg_us"] = {k: round(v[1] / v[0] * 1e3, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        out["kernel_total_us_per_learn"] = {k: round(v[1] / n_learn * 1e3, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    if not args.no_rainbow:
        del collector, env
        out["rainbow"] = rainbow_leg(rank, world, local_rank, dist, args.rainbow_updates, 30)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_iters, W, T)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
