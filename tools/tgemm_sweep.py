#!/usr/bin/env python3
"""Tile / split / grid-order sweep of the tile engine's call sites on the real learners (round 6): ONE process builds the Ape-X learner
at config.ape_x.atari shapes (B = 512) and / or the PPO learner at config.ppo.mujoco shapes (2048-row minibatches), then for every
override string (jh_tgemm_set_cfg grammar: "<site>:<TM>x<TN>[:s<splits>][:x<0|1>],...") runs eager learn() passes under the library's
per-launch event timers (idempotent MFMA launches repeated R times inside one event pair) and prints avg us + fraction of the fp32
MFMA peak per call site.

    python tools/tgemm_sweep.py --apex --hopper [--cfg "6:4x2:s4" --cfg ...] [--repeat 3]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

APEX_CFGS = ["", "*:2x2:x1", "*:4x2", "*:4x2:x1", "*:2x4", "*:2x4:x1",
             "6:2x2:s2", "6:4x2:s2", "6:4x2:s3", "6:4x2:s4", "6:2x4:s2", "6:2x4:s4", "6:4x2:s4:x1",
             "9:4x2:s1", "9:2x4:s1", "9:4x2:s2", "9:2x2:s2",
             "2:4x2:s1,3:4x2:s1", "2:4x2:s2,3:4x2:s2",
             "12:2x4,13:2x4", "12:4x2,13:4x2"]
HOPPER_CFGS = ["", "*:2x2:x1", "*:4x2", "*:2x4", "*:4x2:x1",
               "15:2x2:s2", "15:4x2:s2", "15:2x4:s2", "15:4x2:s4", "15:2x4:s4", "15:2x2:s4",
               "16:2x2:s2", "16:4x2:s2", "16:2x4:s2", "16:4x2:s4", "16:2x4:s4"]


def report(prof, only="tgemm"):
    out = {}
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if only in k or "conv1" in k or "col2im" in k:
            e = {"us": round(v[1] / v[0] * 1e3, 2)}
            if v[2] > 0:
                e["frac"] = round(v[2] / (v[1] * 1e-3) / 157.3e12, 3)
            out[k.replace("jh_tgemm_", "")] = e
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--apex", action="store_true")
    ap.add_argument("--hopper", action="store_true")
    ap.add_argument("--cfg", action="append", default=None)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--passes", type=int, default=2)
    args = ap.parse_args()
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(0)
    np.random.seed(0)
    rng = np.random.RandomState(0)
    results = {}

    def sweep(label, cfgs, one_pass):
        for cfg in cfgs:
            ops.tgemm_set_cfg(cfg)
            one_pass()  # untimed: first launch of a new kernel
            ops.lib_profile(True, repeat=args.repeat)
            for _ in range(args.passes):
                one_pass()
            torch.cuda.synchronize()
            prof = ops.lib_profile_report()
            ops.lib_profile(False)
            r = report(prof)
            results[f"{label} {cfg!r}"] = r
            print(label, repr(cfg), json.dumps(r), flush=True)
        ops.tgemm_set_cfg("")

    if args.apex:
        N, B, n = 200_000, 512, 3
        agent = Agent("ape_x", state_size=[4, 84, 84], action_size=6, hidden_size=512, network="dueling", head="cnn",
                      optim_config={"name": "rmsprop", "eps": 1.5e-7, "lr": 2.5e-4 / 4, "centered": True}, gamma=0.99, buffer_size=N, batch_size=B,
                      clip_grad_norm=40.0, start_train_step=0, target_update_period=2500, run_step=30_000_000, n_step=n, alpha=0.6, beta=0.4,
                      uniform_sample_prob=1e-3, num_workers=64, device="cuda", use_graph=False)
        agent.memory.first_store = False

        def synth(m):
            return {"state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8), "action": rng.randint(0, 6, size=(m, 1)),
                    "reward": rng.choice([-1.0, 0.0, 1.0], p=[0.02, 0.9, 0.08], size=(m, n, 1)).astype(np.float32),
                    "next_state": rng.randint(0, 256, size=(m, 4, 84, 84), dtype=np.uint8), "done": (rng.rand(m, n, 1) < 1e-3)}

        for _ in range(2):
            agent.memory.store_soa(synth(2048), rng.rand(2048) ** 0.5 + 1e-3)
        for _ in range(3):
            agent.learn()
        sweep("apex", args.cfg if args.cfg is not None else APEX_CFGS, agent.learn)
        del agent
        torch.cuda.empty_cache()

    if args.hopper:
        S, A, T, W, Bm, E = 11, 3, 2048, 4, 2048, 2  # 4 workers x 2048 = 4 minibatches x 2 epochs per pass: the 2048-row launches, few of them
        agent = Agent("ppo", state_size=S, action_size=A, hidden_size=512, network="continuous_policy_value", optim_config={"name": "adam", "lr": 3e-4},
                      gamma=0.99, batch_size=Bm, n_step=T, n_epoch=E, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0,
                      use_standardization=True, lr_decay=True, run_step=1_000_000_000, num_workers=W, device="cuda", use_graph=False)
        agent.memory.first_store = False
        M = W * T
        cols = {"state": rng.randn(M, S).astype(np.float32), "action": np.tanh(rng.randn(M, A)).astype(np.float32), "reward": rng.randn(M, 1).astype(np.float32),
                "next_state": rng.randn(M, S).astype(np.float32), "done": (rng.rand(M, 1) < 1e-3)}
        step = [0]

        def one():
            step[0] += T
            agent.process(cols, step[0])

        one()
        sweep("hopper", args.cfg if args.cfg is not None else HOPPER_CFGS, one)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(results, open(os.path.join(out_dir, "tgemm_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
