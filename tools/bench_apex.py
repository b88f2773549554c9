#!/usr/bin/env python3
"""Learner-side measurement at BASELINE.json configs[3] shapes (config.ape_x.atari --env.name pong,
distributed): dueling Nature-CNN, uint8 (4,84,84) frames, A=6, n_step=3, distributed_batch_size=512,
centered RMSprop, clip_grad_norm 40, PER alpha .6 with ACTOR-SIDE initial priorities, synthetic transitions.

One learner iteration = ingest one actor chunk (update_period=100 n-step transitions + their priorities, one
coalesced ring append + one sum-tree push) and one ApeX.learn() (PER sample -> gather -> 3 CNN forwards +
backward + clip + RMSprop on jh_rbnet_* -> jh_td_loss -> priority write-back), replayed as one hipGraph.

    python tools/bench_apex.py [--updates 100]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffer", type=int, default=2_000_000, help="PER slots (config.ape_x.atari: buffer_size 2e6; frame mode: ~21 GB of de-duplicated 84x84 planes "
                                                                    "in HBM, plain rows would be 113 GB of uint8 stacks -- both fit the 288 GB)")
    ap.add_argument("--prefill", type=int, default=50_000, help="transitions in the buffer before the first learn() (config.ape_x.atari: start_train_step 50000)")
    ap.add_argument("--updates", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--actors", type=int, default=0, help="N > 0: N host actor threads publish chunks into the staging ring while the learner runs")
    ap.add_argument("--actor-hz", type=float, default=0.0, help="env steps/s per actor (0: as fast as the ring takes them = ingestion capacity)")
    ap.add_argument("--e2e", type=int, default=0, help="N > 0: END-TO-END async Ape-X with N actors that really ACT: one batched forward per tick on the GPU "
                                                       "(BatchedValueActors), synthetic Atari-shaped envs, vectorised n-step assembly with actor-side priorities, staging ring, learner")
    ap.add_argument("--sync-period", type=int, default=100, help="--e2e: actor ticks between weight syncs (config.ape_x.atari update_period)")
    ap.add_argument("--frames", action="store_true", help="--device-feed: the envs hand over only their newest 84x84 frame (frame mode: the stack for the "
                                                           "forward is rebuilt in HBM from the plane pool; 7 KB instead of 28 KB per env step over PCIe)")
    ap.add_argument("--device-feed", action="store_true", help="--e2e: the actors' stacks stay in HBM (DeviceActorFeed: plane pool + n-step assembly + "
                                                                "actor-side priorities on the acting stream) instead of VecNStepApeX + the pinned staging ring")
    return ap.parse_args(argv)


def main():
    print(json.dumps(run(parse())))


def run(args, dist=None):
    """One measurement -> dict.  dist: an initialised torch.distributed with world size > 1 -> ONE LEARNER PER GPU (north_star: "Ape-X-style many-actor /
    one-learner re-expressed as one learner per GPU with RCCL all-reduce"): every rank runs this function on its own device with its own actors, its own
    replay shard and sum tree; the learners' gradient buckets are averaged per learn() (attach_data_parallel), the IS weights are those of the one
    logical buffer over all shards.  Every rank runs the same number of learner iterations (the all-reduce is a rendezvous)."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(0)  # (identical initial weights on every rank; attach_data_parallel broadcasts rank 0's anyway)
    np.random.seed(dist.get_rank() if dist is not None else 0)
    N, B, n, chunk_rows, filled = args.buffer, args.batch, 3, 100, min(args.prefill, 16384)  # (host-synthesised rows: 115 MB of randint per 2048)
    agent = Agent("ape_x", state_size=[4, 84, 84], action_size=6, hidden_size=512, network="dueling", head="cnn",
                  optim_config={"name": "rmsprop", "eps": 1.5e-7, "lr": 2.5e-4 / 4, "centered": True}, gamma=0.99, buffer_size=N, batch_size=B,
                  clip_grad_norm=40.0, start_train_step=0, target_update_period=2500, run_step=30_000_000, n_step=n, alpha=0.6, beta=0.4,
                  uniform_sample_prob=1e-3, num_workers=64, device="cuda")
    agent.memory.first_store = False
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    if dist is not None and world > 1:
        from jorldy_amd.parallel import attach_data_parallel

        attach_data_parallel(agent, dist)
    rng = np.random.RandomState(rank)
    if args.device_feed:
        assert args.e2e > 0, "--device-feed is a mode of --e2e"
        filled = 0  # the feed owns the (empty) buffer's row format: the actors fill it

    def synth(m):
        return {"state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8), "action": rng.randint(0, 6, size=(m, 1)),
                "reward": rng.choice([-1.0, 0.0, 1.0], p=[0.02, 0.9, 0.08], size=(m, n, 1)).astype(np.float32),
                "next_state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8), "done": (rng.rand(m, n, 1) < 1e-3)}

    for o in range(0, filled, 2048):
        agent.memory.store_soa(synth(2048), rng.rand(2048) ** 0.5 + 1e-3)
    chunk, chunk_prio = (synth(chunk_rows), rng.rand(chunk_rows) + 1e-3) if not args.device_feed else (None, None)

    def iteration():
        agent.memory.store_soa(chunk, chunk_prio)  # one actor's update_period transitions + actor-side priorities
        return agent.learn()

    async_stats = None
    if args.actors > 0:
        # async mode (run_mode.py:212-363 re-expressed): actor threads publish update_period-sized chunks (with their
        # priorities) into the pinned staging ring as fast as they can; the learner loop = drain + learn()
        import threading

        ring = agent.memory.make_ring(64 * chunk_rows, with_priority=True)
        flat = agent.memory.ring_columns(chunk)
        stop = threading.Event()

        def actor():
            period = chunk_rows / args.actor_hz if args.actor_hz > 0 else 0.0
            nxt = time.perf_counter()
            while not stop.is_set():
                try:
                    ring.produce(flat, chunk_prio, timeout_ms=200)
                except Exception:
                    pass  # ring full for 200 ms: the learner is the bottleneck; try again
                if period:
                    nxt += period
                    time.sleep(max(0.0, nxt - time.perf_counter()))

        threads = [threading.Thread(target=actor, daemon=True) for _ in range(args.actors)]
        for t in threads:
            t.start()
        step = 0

        def iteration():
            nonlocal step
            step += 1
            agent.learn_period_stamp = agent.learn_period  # one learn() per learner iteration, like the sync loop below
            return agent.process(None, step)

    e2e_stats = None
    if args.e2e > 0:
        # configs[3] end to end (run_mode.py:212-363 + process.py:7-31,82-97 + distributed_manager.py:33-51 re-expressed):
        #   actor thread  N synthetic Atari-shaped envs stepped in lockstep; per tick ONE batched forward on the GPU for all
        #                 of them (own stream, acting copy of the network, synced every --sync-period ticks), n-step
        #                 windows + actor-side priorities assembled for all N at once, N transitions into the staging ring
        #   learner       (this thread) drain the ring into the device store + sum tree, one ApeX.learn() per iteration
        import threading

        from jorldy_amd.manager import BatchedValueActors, DeviceActorFeed, VecNStepApeX

        NA = args.e2e
        actors = BatchedValueActors(agent, NA)
        feed = ring = nstep = None
        if args.device_feed:
            feed = DeviceActorFeed(actors, agent.memory, n, 0.99, depth=64, prio_eps=1e-3)
        else:
            ring = agent.memory.make_ring(max(16, 64 * 100 // NA) * NA, with_priority=True)
            nstep = VecNStepApeX(NA, n, 0.99, (4, 84, 84), np.uint8)
        frames = rng.randint(0, 256, size=(257, 84, 84), dtype=np.uint8)  # frame pool of the synthetic envs
        stop = threading.Event()
        counters = {"ticks": 0, "t_act": 0.0, "t_host": 0.0}

        def actor_loop():
            torch.cuda.set_device(agent.device)
            arng = np.random.RandomState(7)
            obs = actors.obs_slab
            pos = arng.randint(0, 257, size=NA)
            for c in range(4):
                obs[:, c] = frames[(pos + c) % 257]
            frame_mode = feed is not None and args.frames
            if frame_mode:
                newest = feed.frame_slab
                newest[:] = frames[(pos + 3) % 257]
            while not stop.is_set():
                t0 = time.perf_counter()
                out = feed.act_frames(None, None, training=True) if frame_mode else (feed or actors).act(None, training=True)
                t1 = time.perf_counter()
                # env.step for all actors: reward / done draws of SURVEY.md §8d C4, next frame stack = shift in one new frame
                reward = arng.choice([-1.0, 0.0, 1.0], p=[0.02, 0.9, 0.08], size=(NA, 1)).astype(np.float32)
                done = (arng.rand(NA, 1) < 1e-3).astype(np.float32)
                emitted = None
                if feed is not None:
                    if feed.push(reward, done) < 0:
                        return
                else:
                    emitted = nstep.push(obs, out["action"], reward, done, out["q"])
                pos = (pos + 1) % 257
                if frame_mode:
                    newest[:] = frames[(pos + 3) % 257]  # the env's new frame; the wrapper's stack bookkeeping has no counterpart
                else:
                    obs[:, :3] = obs[:, 1:]
                    obs[:, 3] = frames[(pos + 3) % 257]
                if emitted is not None:
                    cols, prio = emitted
                    try:
                        ring.produce(agent.memory.ring_columns(cols), prio + 1e-3, timeout_ms=500)
                    except Exception:
                        pass  # ring full: the learner is behind; drop this tick's transitions like a full queue would
                counters["ticks"] += 1
                if counters["ticks"] % args.sync_period == 0:
                    actors.sync()
                counters["t_act"] += t1 - t0
                counters["t_host"] += time.perf_counter() - t1

        th = threading.Thread(target=actor_loop, daemon=True)
        th.start()
        step = 0

        def iteration():
            nonlocal step
            step += 1
            agent.learn_period_stamp = agent.learn_period
            return agent.process(None, step)

    if args.device_feed:  # the actors fill the empty buffer up to start_train_step first, the learner only ingests (ape_x.py:150-153)
        t_fill = time.perf_counter()
        while agent.memory.buffer_counter < args.prefill and time.perf_counter() - t_fill < 300:
            agent.num_transitions += agent.memory.drain()
            time.sleep(0.0005)
        filled = int(agent.memory.buffer_counter)
        prefill_s = time.perf_counter() - t_fill

    def fence():  # one learner per GPU: the timed region is bracketed by a barrier on every rank, its length is the slowest rank's
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    for _ in range(args.warmup):
        iteration()
    fence()
    if args.e2e > 0:
        tick0, tact0, thost0 = counters["ticks"], counters["t_act"], counters["t_host"]
    n0 = agent.num_transitions
    t0 = time.perf_counter()
    for _ in range(args.updates):
        r = iteration()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if args.e2e > 0:
        ticks = counters["ticks"] - tick0
        e2e_stats = {"actors": args.e2e, "actor_ticks_per_s": ticks / dt, "env_steps_per_s": ticks * args.e2e / dt,
                     "act_ms_per_tick": (counters["t_act"] - tact0) / max(1, ticks) * 1e3, "host_ms_per_tick": (counters["t_host"] - thost0) / max(1, ticks) * 1e3,
                     "ingested_transitions_per_s": (agent.num_transitions - n0) / dt, "weight_sync_every_ticks": args.sync_period,
                     "path": ("device feed, frame mode (jh_feed_push_frames)" if args.frames else "device feed (jh_feed_tick)") if feed is not None else "host assembler + pinned staging ring",
                     **(ring.stats() if ring is not None else {})}
        stop.set()
        if feed is not None:
            feed.close()
        th.join(timeout=10)
        if feed is not None:
            e2e_stats.update(feed.stats())
        iteration = lambda: agent.learn()
    if args.actors > 0:
        stop.set()
        for t in threads:
            t.join(timeout=5)
        async_stats = {"actors": args.actors, "actor_hz": args.actor_hz, "ingested_transitions_per_s": (agent.num_transitions - n0) / dt, "ingested_GB_per_s": (agent.num_transitions - n0) * 2 * 28224 / dt / 1e9,
                       **ring.stats()}
        iteration = lambda: agent.learn()
    t0 = time.perf_counter()
    for _ in range(30):
        agent.learn()
    torch.cuda.synchronize()
    dt_learn = (time.perf_counter() - t0) / 30
    graphed = agent._graph is not None
    ops.lib_profile(True)
    for _ in range(5):
        iteration()
    torch.cuda.synchronize()
    prof = ops.lib_profile_report()
    ops.lib_profile(False)
    # flops per launch as the library declares them for its MFMA launches (grouped GEMM engine: 2 M N K of every problem of the
    # group; the dedicated conv1 kernels likewise); v = (launches, total ms, total flops)
    kern = {k: {"avg_us": round(v[1] / v[0] * 1e3, 2), **({"TFLOP/s": round(v[2] / (v[1] * 1e-3) / 1e12, 1), "frac_of_157.3_f32_mfma_peak": round(v[2] / (v[1] * 1e-3) / 157.3e12, 3)} if v[2] > 0 else {})}
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    totals = None
    if world > 1:  # whole-job aggregates: every rank's actors' env steps and ingested transitions over the slowest rank's time
        mine = torch.tensor([float((e2e_stats or {}).get("env_steps_per_s", 0.0)) * dt, float(agent.num_transitions - n0)], dtype=torch.float64,
                            device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        totals = {"n_gpus": world, "env_steps_per_s": float(mine[0]) / dt, "ingested_transitions_per_s": float(mine[1]) / dt, "learner_updates_per_s": args.updates / dt,
                  "sampled_transitions_per_s": world * B * args.updates / dt, "parallelism": f"dp{world}: one learner + {args.e2e} actors + one replay shard per GPU, "
                  "gradient bucket averaged per learn(), IS weights of the one logical buffer"}
    out = {
        "whole_job": totals,
        "workload": f"config.ape_x.atari pong-shaped (BASELINE.json configs[3]), synthetic uint8 (4,84,84), A=6, B={B}, n=3, dueling CNN, centered RMSprop, clip 40, "
                    f"PER N={N} ({filled} filled)",
        "timed_s": dt,
        "prefill": {"transitions": filled, "seconds": round(prefill_s, 2)} if args.device_feed else {"transitions": filled},
        "learner_updates_per_s": args.updates / dt,
        "sampled_transitions_per_s": B * args.updates / dt,
        "ingested_transitions_per_s": (async_stats or e2e_stats)["ingested_transitions_per_s"] if (async_stats or e2e_stats) else chunk_rows * args.updates / dt,
        "async": async_stats,
        "end_to_end": e2e_stats,
        "ms_per_iteration_incl_ingest": dt / args.updates * 1e3,
        "ms_per_learn_only": dt_learn * 1e3,
        "learn_in_hipgraph": graphed,
        "last_result": {k: float(v) for k, v in r.items()},
        "lib_kernels": kern,
    }
    return out


if __name__ == "__main__":
    main()
