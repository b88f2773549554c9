#!/usr/bin/env python3
"""Learning curves of the REAL reference beside the ports' (build container only: needs /root/reference; CPU only; test infrastructure like
the rest of oracle/, never imported by the product).

The GPU suite's learning-curve tests (tests/test_learning_curve_gpu.py) compare the HIP agents with the reference's CPU path as the PORTS
restate it (oracle/*_port.py, pinned to the reference's learn() by the golden fixtures) because the reference tree cannot travel to a GPU
box.  VERDICT r5 (N2, "next" #6) asks what that stands for: this script runs the UNMODIFIED reference agents -- core.agent.ppo.PPO,
core.agent.dqn.DQN, core.agent.rainbow.Rainbow, imported from a scratch copy exactly like oracle/gen_golden.py -- through the SAME loops, envs,
hyper-parameters and seeds as the tests' CPU side, next to the ports.  -> JSON (committed as profiles/r06_learning_curve_reference_vs_port.json).

    python oracle/reference_learning_curves.py [--threads 8] [--seeds 1 2] [--skip rainbow]
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
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--skip", nargs="*", default=[])
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_learning_curve_reference_vs_port.json"))
    ap.add_argument("--ppo-cnn", action="store_true", help="instead: the reference's PPO on the CNN head (config.ppo.atari's agent) on the CueFrames image task -> "
                                                           "tests/golden/curves_reference_r06_ppo_cnn.json")
    ap.add_argument("--fixtures", action="store_true", help="instead: the reference's curves for the two tasks that have NO port (continuous PPO on the control env, Ape-X on "
                                                            "CartPole) -> tests/golden/curves_reference_r06.json, what tests/test_learning_curve_gpu.py holds the HIP agents against")
    args = ap.parse_args()
    import torch

    import test_learning_curve_gpu as LC  # the loops, envs and budgets of the GPU suite's CPU side
    from oracle import ppo_port as P
    from oracle.dqn_port import DQNPort
    from oracle.dqn_port import make_env as port_env
    from oracle.rainbow_port import RainbowPort

    torch.set_num_threads(args.threads)
    scratch = tempfile.mkdtemp(prefix="jref_")
    subprocess.check_call(f"cd {args.ref} && tar --exclude='jorldy/core/env/mlagents' -cf - jorldy | (cd {scratch} && tar xf -)", shell=True)
    cwd = os.getcwd()
    os.chdir(os.path.join(scratch, "jorldy"))
    sys.path.insert(0, os.getcwd())
    sys.dont_write_bytecode = True
    out = {"host": {"cores": os.cpu_count(), "torch_threads": args.threads}, "what": "the unmodified reference agents (core.agent.*) and the ports (oracle/*_port.py) "
           "through the loops / envs / seeds of tests/test_learning_curve_gpu.py's CPU side"}
    try:
        from core.agent.dqn import DQN
        from core.agent.ppo import PPO
        from core.agent.rainbow import Rainbow

        W, T, ITERS, RUN_STEP = LC.W, LC.T, LC.ITERS, LC.RUN_STEP
        if args.ppo_cnn:
            t0 = time.time()
            mk = lambda: PPO(device="cpu", **LC.pcn_agent_kwargs())
            fx = {"generator": "oracle/reference_learning_curves.py --ppo-cnn (the unmodified reference agent, CPU, scratch copy)", "seeds": args.seeds,
                  "ppo_cueframes": {"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in LC.PCN.items()}, "metric": "mean reward per transition, per iteration (random play 0.25)",
                                    "reference": [LC.pcn_curve_host(mk, s) for s in args.seeds]}}
            print("ppo cnn done", round(time.time() - t0, 1), [[round(float(np.mean(c[:3])), 3), round(float(np.mean(c[-5:])), 3)] for c in fx["ppo_cueframes"]["reference"]], flush=True)
            os.chdir(cwd)
            with open(os.path.join(ROOT, "tests", "golden", "curves_reference_r06_ppo_cnn.json"), "w") as f:
                json.dump(fx, f)
            print("wrote tests/golden/curves_reference_r06_ppo_cnn.json")
            return
        if args.fixtures:
            from core.agent.ape_x import ApeX

            fx = {"generator": "oracle/reference_learning_curves.py --fixtures (the unmodified reference agents, CPU, scratch copy)", "seeds": args.seeds}
            t0 = time.time()
            mk = lambda: PPO(state_size=LC.CTL["S"], action_size=LC.CTL["A"], device="cpu", **LC.ctl_agent_kwargs())
            fx["ppo_control"] = {"config": LC.CTL, "metric": "mean reward per transition, per iteration", "reference": [LC.ctl_curve_host(mk, s) for s in args.seeds]}
            print("ppo control done", round(time.time() - t0, 1), [round(float(np.mean(c[-5:])), 3) for c in fx["ppo_control"]["reference"]], flush=True)
            t0 = time.time()
            mk = lambda: ApeX(state_size=4, action_size=2, device="cpu", **LC.apex_agent_kwargs())
            fx["apex_cartpole"] = {"config": LC.APEX, "metric": f"mean episode length per {LC.APEX['chunk']} env steps", "reference": [LC.apex_curve(mk, s, host_env=True) for s in args.seeds]}
            print("apex done", round(time.time() - t0, 1), [round(float(np.mean(c[-3:])), 1) for c in fx["apex_cartpole"]["reference"]], flush=True)
            os.chdir(cwd)
            with open(os.path.join(ROOT, "tests", "golden", "curves_reference_r06.json"), "w") as f:
                json.dump(fx, f)
            print("wrote tests/golden/curves_reference_r06.json")
            return
        if "ppo" not in args.skip:
            def ref_ppo_curve(seed):
                np.random.seed(seed)
                torch.manual_seed(seed)
                agent = PPO(state_size=4, action_size=2, hidden_size=512, network="discrete_policy_value", optim_config={"name": "adam", "lr": 2.5e-4}, batch_size=256,
                            n_step=T, n_epoch=3, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, gamma=0.99, run_step=RUN_STEP,
                            num_workers=W, device="cpu")
                envs = [P._OneEnv(seed=1000 * seed + w) for w in range(W)]
                states = [e.reset_obs() for e in envs]
                curve, step = [], 0
                for _ in range(ITERS):
                    trs = P.sync_iteration(agent, envs, states, T)  # Actor.run of distributed_manager.py:76-92 with agent.act of whoever is passed
                    curve.append(min(500.0, len(trs) / max(1, sum(int(t["done"][0, 0]) for t in trs))))
                    step += T
                    agent.process(trs, step)
                return curve

            t0 = time.time()
            out["ppo_cartpole"] = {"metric": "mean episode length per iteration (max 500)", "iterations": ITERS, "seeds": args.seeds,
                                   "reference": [ref_ppo_curve(s) for s in args.seeds], "port": [LC._cpu_curve(s, ITERS) for s in args.seeds]}
            out["ppo_cartpole"]["seconds"] = round(time.time() - t0, 1)
            print("ppo done", out["ppo_cartpole"]["seconds"], flush=True)

        if "dqn" not in args.skip:
            cfg = dict(gamma=0.99, epsilon_init=1.0, epsilon_min=0.01, explore_ratio=0.2, buffer_size=50000, batch_size=32, start_train_step=2000, target_update_period=500)

            def cpu_step(env, action):
                nxt, rew, done = env.step(np.asarray(action).reshape(-1))
                return nxt.astype(np.float32), rew.reshape(1, 1).astype(np.float64), done.reshape(1, 1), env.obs().astype(np.float32)

            ref_agent = lambda: DQN(state_size=4, action_size=2, hidden_size=512, network="discrete_q_network", optim_config={"name": "adam", "lr": 1e-4}, lr_decay=True,
                                    run_step=LC.DQN_RUN_STEP, device="cpu", **cfg)
            port_agent = lambda: DQNPort(4, 2, 512, lr=1e-4, run_step=LC.DQN_RUN_STEP, **cfg)
            t0 = time.time()
            out["dqn_cartpole"] = {"metric": "mean episode length per 1000 env steps (max 500)", "steps": LC.DQN_STEPS, "seeds": args.seeds,
                                   "reference": [LC._dqn_curve(ref_agent, lambda s: port_env(1000 + s), cpu_step, s) for s in args.seeds],
                                   "port": [LC._dqn_curve(port_agent, lambda s: port_env(1000 + s), cpu_step, s) for s in args.seeds]}
            out["dqn_cartpole"]["seconds"] = round(time.time() - t0, 1)
            print("dqn done", out["dqn_cartpole"]["seconds"], flush=True)

        if "rainbow" not in args.skip:
            S, A = (4, 44, 52), 4
            hp = dict(hidden_size=128, gamma=0.99, buffer_size=4096, batch_size=32, n_step=3, alpha=0.5, beta=0.4, uniform_sample_prob=1e-3, v_min=-1.0, v_max=2.0, num_support=21)

            def ref_rb(seed):
                np.random.seed(seed)
                torch.manual_seed(seed)
                agent = Rainbow(state_size=list(S), action_size=A, head="cnn", optim_config={"name": "adam", "lr": 2.5e-4}, start_train_step=200, learn_period=4,
                                target_update_period=400, lr_decay=False, run_step=30_000_000, device="cpu", **hp)
                return LC._rainbow_curve(agent, seed)

            def port_rb(seed):
                np.random.seed(seed)
                torch.manual_seed(seed)
                agent = RainbowPort(S, A, lr=2.5e-4, **hp)
                agent.start_train_step, agent.learn_period, agent.target_update_period = 200, 4, 400
                return LC._rainbow_curve(agent, seed)

            t0 = time.time()
            out["rainbow_cueframes"] = {"metric": f"mean reward per {LC.RB_CHUNK} env steps (random play 0.25)", "steps": LC.RB_STEPS, "seeds": args.seeds,
                                        "reference": [ref_rb(s) for s in args.seeds], "port": [port_rb(s) for s in args.seeds]}
            out["rainbow_cueframes"]["seconds"] = round(time.time() - t0, 1)
            print("rainbow done", out["rainbow_cueframes"]["seconds"], flush=True)
    finally:
        os.chdir(cwd)
        shutil.rmtree(scratch, ignore_errors=True)
    # summary: start / end of every curve family
    for k, v in out.items():
        if isinstance(v, dict) and "reference" in v:
            first, last = (5, 10) if k == "ppo_cartpole" else (2, 4 if k == "dqn_cartpole" else 2)
            v["summary"] = {who: {"start": float(np.mean([np.mean(c[:first]) for c in v[who]])), "end": float(np.mean([np.mean(c[-last:]) for c in v[who]]))} for who in ("reference", "port")}
            print(k, v["summary"])
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
