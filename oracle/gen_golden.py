#!/usr/bin/env python3
"""Generate golden input/output vectors by running the UNMODIFIED reference.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference
exists); the GPU box never runs this.  Output: small .npz fixtures under
tests/golden/ that pin (a) the CPU oracle in oracle/jorldy_oracle.py and
(b) the HIP path, because the reference's own tests hold no golden vectors
(SURVEY.md §4 / §8c).

How the reference is exercised without editing it:
  * the tree is copied (minus Unity binaries) to a scratch dir and imported
    from there with PYTHONDONTWRITEBYTECODE (importing `core` rewrites
    `_*_dict.txt` files, SURVEY.md §8c "import trap");
  * intermediate values inside `learn()` are captured with a `sys.settrace`
    line tap that snapshots named locals, and with forward hooks on the
    output `nn.Linear`s so d(loss)/d(logits) is retained;
  * numpy / torch global RNGs are seeded explicitly before every call so the
    HIP mirror can re-draw the identical sample indices.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import inspect
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth  # noqa: E402  (recipes shared with the tests: weights / frames regenerated from a seed)


def _to_np(v):
    import torch

    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy().copy()
    if isinstance(v, np.ndarray):
        return v.copy()
    if isinstance(v, (list, tuple)) and v and isinstance(v[0], (int, float, np.integer, np.floating)):
        return np.asarray(v)
    if isinstance(v, (int, float, bool, np.integer, np.floating)):
        return np.asarray(v)
    return None


class LineTap:
    """Snapshot named locals of `func` each time execution reaches a source line
    containing one of the given marker substrings (BEFORE that line executes),
    and once more at return (key '<return>')."""

    def __init__(self, func, markers):
        self.code = func.__code__
        src, first = inspect.getsourcelines(func)
        self.line_to_key = {}
        for key, (needle, names) in markers.items():
            hits = [i for i, s in enumerate(src) if needle in s]
            assert hits, f"marker {needle!r} not found in {func.__qualname__}"
            self.line_to_key[first + hits[0]] = (key, names)
        self.ret_names = markers.get("<return>", (None, ()))[1] if "<return>" in markers else ()
        self.records = {}  # key -> list of dict
        self.on_line = {}  # key -> callback(frame)

    def _local(self, frame, event, arg):
        if event == "line" and frame.f_lineno in self.line_to_key:
            key, names = self.line_to_key[frame.f_lineno]
            if key in self.on_line:
                self.on_line[key](frame)
            snap = {}
            for n in names:
                if n in frame.f_locals:
                    a = _to_np(frame.f_locals[n])
                    if a is not None:
                        snap[n] = a
            self.records.setdefault(key, []).append(snap)
        return self._local

    def _global(self, frame, event, arg):
        if event == "call" and frame.f_code is self.code:
            return self._local
        return None

    def __enter__(self):
        sys.settrace(self._global)
        return self

    def __exit__(self, *a):
        sys.settrace(None)


RECIPE_SEED = 20260925


def flat(prefix, d, out):
    for k, v in d.items():
        out[f"{prefix}{k}"] = v


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


# ----------------------------------------------------------------------------
# buffers
# ----------------------------------------------------------------------------
def gen_buffers(out_dir):
    from core.buffer import ReplayBuffer, RolloutBuffer, PERBuffer

    rng = np.random.RandomState(1234)

    def mk(n, S=4):
        trs = []
        for _ in range(n):
            trs.append(
                {
                    "state": rng.randn(1, S).astype(np.float32),
                    "action": rng.randint(0, 3, size=(1, 1)),
                    "reward": rng.randn(1, 1),
                    "next_state": rng.randn(1, S).astype(np.float32),
                    "done": rng.rand(1, 1) < 0.2,
                }
            )
        return trs

    # --- ReplayBuffer: ring wrap + seeded sample ---------------------------------
    out = {}
    buf = ReplayBuffer(16)
    buf.first_store = False
    trs = mk(23)
    buf.store(trs[:10])
    buf.store(trs[10:])
    out["n_store"] = np.asarray([10, 13])
    for k in trs[0]:
        out[f"in_{k}"] = np.concatenate([t[k] for t in trs], 0)
    np.random.seed(7)
    s = buf.sample(8)
    for k, v in s.items():
        out[f"sample_{k}"] = v
    out["buffer_index"] = np.asarray(buf.buffer_index)
    out["buffer_counter"] = np.asarray(buf.buffer_counter)
    np.savez(os.path.join(out_dir, "replay_buffer.npz"), **out)

    # --- RolloutBuffer ---------------------------------------------------------------
    out = {}
    rb = RolloutBuffer()
    rb.first_store = False
    trs = mk(12)
    rb.store(trs[:5])
    rb.store(trs[5:])
    s = rb.sample()
    for k in trs[0]:
        out[f"in_{k}"] = np.concatenate([t[k] for t in trs], 0)
    for k, v in s.items():
        out[f"sample_{k}"] = v
    out["size_after"] = np.asarray(rb.size)
    np.savez(os.path.join(out_dir, "rollout_buffer.npz"), **out)

    # --- PERBuffer scenario: N=64, wrap, sample, write-back, actor priorities ------
    # (every transition of one buffer carries "priority" or none does: the key is stacked on sample)
    for name, N, n_ops, B, with_prio in (
        ("per_n64", 64, 6, 16, False),
        ("per_n1000", 1000, 8, 32, False),
        ("per_n1000_prio", 1000, 6, 32, True),
    ):
        out = {}
        per = PERBuffer(N, uniform_sample_prob=0.05 if N == 64 else 1e-3)
        per.first_store = False
        prng = np.random.RandomState(99 + N)
        ops = []
        step = 0
        for op in range(n_ops):
            # store a chunk (some ops carry explicit actor-side priorities)
            n_st = int(prng.randint(N // 3, N // 2 + 5))
            trs = mk(n_st)
            if with_prio:
                pr = prng.rand(n_st) * 3.0
                for t, p in zip(trs, pr):
                    t["priority"] = np.asarray([[p]])  # (1,1) like ape_x.py:174-199
                out[f"op{op}_store_prio"] = pr
            for k in ("state", "reward"):
                out[f"op{op}_store_{k}"] = np.concatenate([t[k] for t in trs], 0)
            out[f"op{op}_n_store"] = np.asarray(n_st)
            per.store(trs)
            out[f"op{op}_tree_after_store"] = per.sum_tree.copy()
            out[f"op{op}_maxp_after_store"] = np.asarray(per.max_priority)
            out[f"op{op}_tree_index"] = np.asarray(per.tree_index)
            # sample
            seed = 1000 + op
            np.random.seed(seed)
            beta = 0.4 + 0.1 * op
            tr, w, idx, sp, mp = per.sample(beta, B)
            out[f"op{op}_seed"] = np.asarray(seed)
            out[f"op{op}_beta"] = np.asarray(beta)
            out[f"op{op}_weights"] = w
            out[f"op{op}_indices"] = idx
            out[f"op{op}_sampled_p"] = np.asarray(sp)
            out[f"op{op}_mean_p"] = np.asarray(mp)
            out[f"op{op}_sample_state"] = tr["state"]
            out[f"op{op}_sample_reward"] = tr["reward"]
            # priority write-back the way the agents do it: fp32 tensor -> .item()
            newp = (prng.rand(B).astype(np.float32) ** 2 * 2.0).astype(np.float32)
            if op == 1:  # force duplicate indices in one batch (last write wins)
                idx = idx.copy()
                idx[1] = idx[0]
                idx[5] = idx[0]
            for i, p in zip(idx, newp):
                per.update_priority(float(p), int(i))
            out[f"op{op}_upd_idx"] = idx
            out[f"op{op}_upd_p"] = newp
            out[f"op{op}_tree_after_update"] = per.sum_tree.copy()
            out[f"op{op}_maxp_after_update"] = np.asarray(per.max_priority)
        out["N"] = np.asarray(N)
        out["B"] = np.asarray(B)
        out["usp"] = np.asarray(per.uniform_sample_prob)
        out["n_ops"] = np.asarray(n_ops)
        out["with_prio"] = np.asarray(with_prio)
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **out)


# ----------------------------------------------------------------------------
# PPO
# ----------------------------------------------------------------------------
def gen_ppo(out_dir, only=None):
    import torch
    from core.agent.ppo import PPO

    cases = [
        # name, S, A, hidden, W, T, batch, n_epoch, continuous, recipe
        ("ppo_disc_small", 4, 3, 32, 4, 16, 16, 2, False, False),
        ("ppo_disc_cartpole", 4, 2, 64, 8, 128, 256, 3, False, False),
        ("ppo_cont_small", 5, 3, 32, 4, 16, 32, 2, True, False),
        ("ppo_cont_hopper", 11, 3, 64, 4, 64, 64, 2, True, False),
        # BASELINE widths (recipe = initial weights regenerated from a seed, big arrays stored thinned):
        # config.ppo.cartpole exactly (hidden 512, 8 x 128, minibatch 256, 3 epochs; config/ppo/cartpole.py:9-40)
        ("ppo_disc_cartpole_h512", 4, 2, 512, 8, 128, 256, 3, False, True),
        # config.ppo.mujoco Hopper shapes (S=11, A=3, hidden 512, T=2048, distributed batch 2048;
        # config/ppo/mujoco.py:8-41) with 2 workers x 2 epochs = 4 minibatches of 2048 rows
        ("ppo_cont_hopper_real", 11, 3, 512, 2, 2048, 2048, 2, True, True),
        # config.ppo.mujoco on its OTHER envs (config/ppo/mujoco.py:3-4 takes --env.name half_cheetah | walker | ant ...): more than 8 head
        # outputs (2 A + 1 = 13 / 17).  HalfCheetah-v3 shapes (S=17, A=6) with minibatches of 1024 rows (the tiled engine), Ant shapes
        # (S=27, A=8) with minibatches of 256 rows (the separate forward / backward calls)
        # (11th element = the rollout seed, CHOSEN so that no ReLU pre-activation of any forward pass of this learn() lies within 3e-7
        # of zero -- `relu_margin` in the fixture: two correct fp32 evaluations of a 512-term dot product differ by ~1e-7, and a unit whose
        # pre-activation is that close to zero is on in one and off in the other: with seed 5 exactly ONE of HalfCheetah's 8 M ReLU
        # decisions (row 297, unit 344 of minibatch 0: -3.2e-7 in float64, +6.0e-8 in torch's fp32) moved the trunk gradients by 1e-3 of
        # their largest entry -- a property of the input, not of either implementation.  Among seeds 5..16 HalfCheetah's best margin is 3.7e-7
        # (seed 15), Ant's 3.2e-6 (seed 5).  JH_GEN_ROLLOUT_SEED overrides for the search.)
        ("ppo_cont_halfcheetah", 17, 6, 512, 2, 1024, 1024, 2, True, True, 15),
        ("ppo_cont_ant_mb256", 27, 8, 512, 2, 256, 256, 2, True, True, 5),
        # round 6: PPO on the CNN head (policy_value.py:8-22 over head.py:21-61).  A small image, and config.ppo.atari's shapes exactly
        # ((4, 84, 84) uint8 frames, hidden 512, minibatch 32, lr 2.5e-4; config/ppo/atari.py:16-36) with A = 6 (Pong) on a 2 x 32 rollout
        ("ppo_disc_cnn_small", (4, 44, 52), 4, 64, 2, 16, 16, 2, False, True),
        ("ppo_disc_atari", (4, 84, 84), 6, 512, 2, 32, 32, 2, False, True),
    ]
    for case in cases:
        name, S, A, H, W, T, B, E, cont, recipe = case[:10]
        rollout_seed = int(os.environ.get("JH_GEN_ROLLOUT_SEED", case[10])) if len(case) > 10 else 5
        if only is not None and name not in only:
            continue
        torch.manual_seed(11)
        np.random.seed(11)
        agent = PPO(
            state_size=list(S) if isinstance(S, tuple) else S,
            action_size=A,
            hidden_size=H,
            network="continuous_policy_value" if cont else "discrete_policy_value",
            head="cnn" if isinstance(S, tuple) else "mlp",
            optim_config={"name": "adam", "lr": 2.5e-4},
            batch_size=B,
            n_step=T,
            n_epoch=E,
            _lambda=0.95,
            epsilon_clip=0.1,
            vf_coef=1.0,
            ent_coef=0.01,
            clip_grad_norm=1.0,
            gamma=0.99,
            run_step=100000,
            num_workers=W,
            device="cpu",
        )
        with torch.no_grad():
            if recipe:
                rec = synth.ppo_recipe({k: v.shape for k, v in agent.network.state_dict().items()}, RECIPE_SEED)
                for k, p in agent.network.named_parameters():
                    p.copy_(torch.from_numpy(rec[k]))
            else:
                # perturb the heads so pi is not ~uniform / value not ~0 (policy gain is 0.01)
                for p in agent.network.parameters():
                    p.add_(0.05 * torch.randn_like(p))
        sd0 = sd_to_np(agent.network.state_dict())

        rng = np.random.RandomState(rollout_seed)
        M = W * T
        trs = synth.ppo_image_rollout(rng, M, S, A) if isinstance(S, tuple) else synth.ppo_rollout(rng, M, S, A, cont, clamp_every=0 if recipe else 17)
        agent.memory.first_store = False

        # capture pre-softmax / pre-clamp head outputs with grads
        head_out = {}

        def mk_hook(tag):
            def hook(mod, inp, outp):
                if outp.requires_grad:
                    outp.retain_grad()
                head_out.setdefault(tag, []).append(outp)

            return hook

        hooks = []
        if cont:
            hooks.append(agent.network.mu.register_forward_hook(mk_hook("mu_raw")))
            hooks.append(agent.network.log_std.register_forward_hook(mk_hook("log_std_raw")))
        else:
            hooks.append(agent.network.pi.register_forward_hook(mk_hook("logits")))
        hooks.append(agent.network.v.register_forward_hook(mk_hook("v")))
        relu_margin = [np.inf]
        if len(case) > 10:  # smallest |pre-activation| over every forward pass of this learn() (both hidden layers)
            def margin_hook(mod, inp, outp):
                relu_margin[0] = min(relu_margin[0], float(outp.detach().abs().min()))

            hooks.append(agent.network.head.l.register_forward_hook(margin_hook))
            hooks.append(agent.network.l.register_forward_hook(margin_hook))

        markers = {
            "gae_done": ("mean_ret = ret.mean().item()", ["value", "next_value", "delta", "adv", "ret", "log_prob_old", "reward", "done"]),
            "mb_loss": ("self.optimizer.zero_grad", ["idx", "ratio", "actor_loss", "critic_loss", "critic_loss1", "critic_loss2", "entropy_loss", "loss", "value_pred", "log_prob"]),
            "mb_grad": ("torch.nn.utils.clip_grad_norm_", []),
            "mb_step": ("self.optimizer.step()", []),
        }
        tap = LineTap(PPO.learn, markers)
        # keyed by minibatch index: a multi-line statement fires its first line more than once
        grads_raw, grads_clip, head_grads = {}, {}, {}

        def on_grad(frame):
            mbi = len(tap.records["mb_loss"]) - 1
            grads_raw[mbi] = {k: p.grad.detach().numpy().copy() for k, p in agent.network.named_parameters()}
            # the LAST two/three forward outputs are this minibatch's heads
            hg = {}
            for tag, lst in head_out.items():
                hg[tag] = lst[-1].detach().numpy().copy()
                hg["d_" + tag] = lst[-1].grad.detach().numpy().copy()
            head_grads[mbi] = hg

        def on_step(frame):
            mbi = len(tap.records["mb_loss"]) - 1
            grads_clip[mbi] = {k: p.grad.detach().numpy().copy() for k, p in agent.network.named_parameters()}

        tap.on_line["mb_grad"] = on_grad
        tap.on_line["mb_step"] = on_step

        np.random.seed(21)
        torch.manual_seed(21)
        with tap:
            result = agent.process(trs, T)
        for h in hooks:
            h.remove()
        assert result, "learn did not run"

        out = {}
        out["cfg"] = np.asarray([0 if isinstance(S, tuple) else S, A, H, W, T, B, E, int(cont)])
        if isinstance(S, tuple):
            out["state_shape"] = np.asarray(S)
        out["hyper"] = np.asarray([0.99, 0.95, 0.1, 1.0, 0.01, 1.0, 2.5e-4])  # gamma, lambda, eps, vf, ent, clip, lr
        out["np_seed"] = np.asarray(21)
        th = synth.thin if recipe else (lambda a: a)
        if recipe:
            # inputs: synth.ppo_rollout(RandomState(5), M, S, A, cont); weights: synth.ppo_recipe
            out["recipe_seed"] = np.asarray(RECIPE_SEED)
            out["rollout_seed"] = np.asarray(rollout_seed)
            if len(case) > 10:
                out["relu_margin"] = np.asarray(relu_margin[0])
            for k in ("state", "reward", "action"):
                out[f"in_{k}_check"] = synth.row_checksum(np.concatenate([t[k] for t in trs], 0).astype(np.float32))[:: max(1, M // 64)]
            flat("sd0_thin/", {k: th(v) for k, v in sd0.items()}, out)
            flat("sd1_thin/", {k: th(v) for k, v in sd_to_np(agent.network.state_dict()).items()}, out)
        else:
            for k in ("state", "next_state", "reward", "done", "action"):
                out[f"in_{k}"] = np.concatenate([t[k] for t in trs], 0)
            flat("sd0/", sd0, out)
            flat("sd1/", sd_to_np(agent.network.state_dict()), out)
        flat("gae/", tap.records["gae_done"][0], out)
        nmb = len(tap.records["mb_loss"])
        out["n_minibatch"] = np.asarray(nmb)
        for i in range(nmb):
            flat(f"mb{i}/", tap.records["mb_loss"][i], out)
            flat(f"mb{i}/head/", head_grads[i], out)
            if i in (0, nmb - 1):  # param grads are big; keep first and last only
                flat(f"mb{i}/grad_raw/", {k: th(v) for k, v in grads_raw[i].items()}, out)
                flat(f"mb{i}/grad_clip/", {k: th(v) for k, v in grads_clip[i].items()}, out)
                if recipe:
                    out[f"mb{i}/grad_raw_norm"] = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in grads_raw[i].values()))
                    for k, v in grads_raw[i].items():
                        out[f"mb{i}/grad_raw_absmax/{k}"] = np.abs(v).max()
        for k, v in result.items():
            out[f"result/{k}"] = np.asarray(v)
        out["lr_after"] = np.asarray(agent.optimizer.param_groups[0]["lr"])
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **out)
        print(name, {k: float(v) for k, v in result.items()}, ("relu_margin %.3g (rollout seed %d)" % (relu_margin[0], rollout_seed)) if len(case) > 10 else "")


# ----------------------------------------------------------------------------
# DQN family
# ----------------------------------------------------------------------------
def _fill(agent, n, S, A, rng, n_step=None, with_q=False):
    """Drive interact_callback + memory.store the way Actor.run / run_mode do."""
    for i in range(n):
        t = synth.raw_transition(rng, S, A, with_q)
        t = agent.interact_callback(t)
        if t:
            agent.memory.store([t])


def gen_dqn_family(out_dir, only=None):
    import torch
    from core.agent.dqn import DQN
    from core.agent.double import Double
    from core.agent.per import PER
    from core.agent.multistep import Multistep
    from core.agent.ape_x import ApeX
    from core.agent.c51 import C51
    from core.agent.rainbow import Rainbow

    S, A, H, B = 4, 3, 32, 32
    common = dict(
        state_size=S,
        action_size=A,
        hidden_size=H,
        optim_config={"name": "adam", "lr": 1e-3},
        gamma=0.99,
        buffer_size=256,
        batch_size=B,
        start_train_step=0,
        target_update_period=10000,
        run_step=100000,
        device="cpu",
    )
    specs = [
        ("dqn", DQN, {}, dict(markers=["q", "target_q", "next_q", "loss"])),
        ("double", Double, {}, dict(markers=["q", "target_q", "next_q", "next_target_q", "max_a", "loss"])),
        ("per", PER, dict(alpha=0.6, beta=0.4, learn_period=1, uniform_sample_prob=0.05), dict(markers=["q", "target_q", "next_q", "next_target_q", "max_a", "td_error", "p_j", "loss", "weights", "indices"])),
        ("multistep", Multistep, dict(n_step=3), dict(markers=["q", "target_q", "next_q", "loss", "reward", "done"])),
        ("ape_x", ApeX, dict(n_step=3, alpha=0.6, beta=0.4, learn_period=1, uniform_sample_prob=0.05, num_workers=4, clip_grad_norm=40.0), dict(markers=["q", "target_q", "next_q", "next_target_q", "max_a", "td_error", "p_j", "loss", "weights", "indices", "reward", "done"], with_q=True)),
        ("c51", C51, dict(v_min=-2, v_max=5, num_support=21), dict(markers=["logit", "p_logit", "q_action", "p_action", "target_p_logit", "target_q_action", "target_action", "target_p_action", "Tz", "b", "l", "u", "target_dist", "loss"])),
        ("rainbow", Rainbow, dict(n_step=3, alpha=0.5, beta=0.4, learn_period=1, uniform_sample_prob=0.05, v_min=-1, v_max=10, num_support=51), dict(markers=["logit", "p_logit", "q_action", "p_action", "next_q_action", "target_p_logit", "target_action", "target_p_action", "Tz", "b", "l", "u", "target_dist", "KL", "p_j", "loss", "weights", "indices", "reward", "done"])),
    ]
    # Nature-CNN head on a small non-square image (conv 8/4, 4/2, 3/1 -> 2x3x64 features)
    specs.append(("rainbow_cnn", Rainbow, dict(n_step=3, alpha=0.5, beta=0.4, learn_period=1, uniform_sample_prob=0.05, v_min=-1, v_max=10, num_support=51),
                  dict(markers=specs[-1][3]["markers"], over=dict(state_size=(4, 44, 52), head="cnn", batch_size=8, buffer_size=64), fill=40)))
    # config.rainbow.atari shapes exactly ((4,84,84) uint8 frames, A=4 = Breakout, B=32, hidden 512, n=3, K=51,
    # v in [-1,10], alpha .5, beta .4; config/rainbow/atari.py:16-44), PER of 64 slots filled by 60 env steps.
    # recipe: frames + initial weights regenerated from seeds (oracle/synth.py), big outputs stored thinned.
    specs.append(("rainbow_cnn_atari", Rainbow, dict(n_step=3, alpha=0.5, beta=0.4, learn_period=1, uniform_sample_prob=1e-3, v_min=-1, v_max=10, num_support=51),
                  dict(markers=specs[-2][3]["markers"], over=dict(state_size=(4, 84, 84), action_size=4, hidden_size=512, head="cnn", batch_size=32, buffer_size=64,
                                                                  optim_config={"name": "adam", "lr": 6.25e-5}), fill=60, recipe=True)))
    # config.dqn.cartpole exactly (BASELINE configs[0]: S=4, A=2, hidden 512, B=32, Adam 1e-4; config/dqn/cartpole.py:9-26)
    specs.append(("dqn_h512", DQN, {}, dict(markers=["q", "target_q", "next_q", "loss"],
                                            over=dict(state_size=4, action_size=2, hidden_size=512, batch_size=32, optim_config={"name": "adam", "lr": 1e-4}),
                                            recipe=True, opt_state=("exp_avg", "exp_avg_sq"))))
    # config.ape_x.atari shapes exactly (BASELINE configs[3]: dueling network + cnn head on (4,84,84) uint8 frames, A=6 = Pong,
    # distributed_batch_size 512, n=3, alpha .6, beta .4, usp 1e-3, CENTERED RMSprop eps 1.5e-7 lr 6.25e-5, clip_grad_norm 40;
    # config/ape_x/atari.py:16-55, learner core/agent/ape_x.py:79-133), PER of 640 slots filled by 600 env steps.
    specs.append(("ape_x_cnn_atari", ApeX, dict(n_step=3, alpha=0.6, beta=0.4, learn_period=1, uniform_sample_prob=1e-3, num_workers=4, clip_grad_norm=40.0),
                  dict(markers=["q", "target_q", "next_q", "next_target_q", "max_a", "td_error", "p_j", "loss", "weights", "indices", "reward", "done"], with_q=True,
                       over=dict(state_size=(4, 84, 84), action_size=6, hidden_size=512, network="dueling", head="cnn", batch_size=512, buffer_size=640,
                                 optim_config={"name": "rmsprop", "lr": 2.5e-4 / 4, "eps": 1.5e-7, "centered": True}),
                       fill=600, recipe=True, opt_state=("square_avg", "grad_avg"), grad64=True)))
    # the same learner on a small image with clip_grad_norm BELOW the gradient norm (at config.ape_x.atari's 40 the clip is a no-op
    # for any sane batch: the fixture above records norm 8.9): pins clip -> centered RMSprop against the reference
    specs.append(("ape_x_cnn_clip", ApeX, dict(n_step=3, alpha=0.6, beta=0.4, learn_period=1, uniform_sample_prob=0.05, num_workers=4, clip_grad_norm=0.5),
                  dict(markers=specs[-1][3]["markers"], with_q=True,
                       over=dict(state_size=(4, 44, 52), action_size=6, hidden_size=64, network="dueling", head="cnn", batch_size=8, buffer_size=64,
                                 optim_config={"name": "rmsprop", "lr": 2.5e-4 / 4, "eps": 1.5e-7, "centered": True}),
                       fill=40, recipe=True, opt_state=("square_avg", "grad_avg"))))
    big = ("rainbow_cnn", "rainbow_cnn_atari", "dqn_h512", "ape_x_cnn_atari", "ape_x_cnn_clip")
    for name, cls, extra, opt in specs:
        if (only is None and name in big) or (only is not None and name != only):
            continue
        torch.manual_seed(3)
        np.random.seed(3)
        kw = dict(common)
        kw.update(extra)
        kw.update(opt.get("over", {}))
        S, B, A, H = kw["state_size"], kw["batch_size"], kw["action_size"], kw["hidden_size"]
        recipe = opt.get("recipe", False)
        agent = cls(**kw)
        with torch.no_grad():
            if recipe:
                shapes = {k: v.shape for k, v in agent.network.state_dict().items()}
                for net, seed in ((agent.network, RECIPE_SEED), (agent.target_network, RECIPE_SEED + 1)):
                    rec = synth.recipe_state_dict(shapes, seed)
                    for k, p in net.named_parameters():
                        p.copy_(torch.from_numpy(rec[k]))
            else:
                for p in agent.network.parameters():
                    p.add_(0.1 * torch.randn_like(p))
                # target differs from online so double-Q is not degenerate
                for p in agent.target_network.parameters():
                    p.add_(0.1 * torch.randn_like(p))
        agent.memory.first_store = False
        rng = np.random.RandomState(17)
        _fill(agent, opt.get("fill", 200), S, A, rng, with_q=opt.get("with_q", False))
        is_per = hasattr(agent.memory, "sum_tree")
        if is_per:
            # non-trivial priorities before the sampled learn
            for leaf in range(agent.memory.size):
                agent.memory.update_priority(float(np.float32(rng.rand() ** 2 + 0.01)), leaf + agent.memory.first_leaf_index)
        sd0 = sd_to_np(agent.network.state_dict())
        sdt = sd_to_np(agent.target_network.state_dict())

        out = {}
        # buffer contents in slot order (so the mirror can be loaded identically)
        n = agent.memory.size
        keys = list(agent.memory.buffer[0].keys())
        for k in keys:
            if k == "priority":
                continue
            col = np.concatenate([agent.memory.buffer[i][k] for i in range(n)], 0)
            if recipe and k in ("state", "next_state") and not isinstance(S, (int, np.integer)):
                # frames: raw = [synth.raw_transition(RandomState(17), S, A) ...]; slot i holds raw[i].state and
                # raw[i + n_step - 1].next_state (rainbow.py:294-308); the checksums pin the regenerated frames
                out[f"buf_{k}_check"] = synth.row_checksum(col)
            else:
                out[f"buf_{k}"] = col
        if recipe:
            out["fill"] = np.asarray(opt.get("fill", 200))
            out["fill_seed"] = np.asarray(17)
            out["recipe_seed"] = np.asarray(RECIPE_SEED)
        if is_per:
            out["tree0"] = agent.memory.sum_tree.copy()
            out["maxp0"] = np.asarray(agent.memory.max_priority)
            out["tree_index0"] = np.asarray(agent.memory.tree_index)

        is_dist = name in ("c51", "rainbow")
        loss_line = "self.optimizer.zero_grad"
        markers = {
            "pre_step": (loss_line, opt["markers"] + (["action"] if recipe else ["state", "action", "next_state"])),
            "step": ("self.optimizer.step()", []),
        }
        has_clip = "clip_grad_norm_" in inspect.getsource(cls.learn)
        if has_clip and recipe:
            markers["pre_clip"] = ("torch.nn.utils.clip_grad_norm_", [])
        tap = LineTap(cls.learn, markers)
        graw = {}
        gpre = {}
        head = {}

        def on_step(frame):
            graw.update({k: p.grad.detach().numpy().copy() for k, p in agent.network.named_parameters()})
            for nm in ("q", "logit"):
                if nm in frame.f_locals and frame.f_locals[nm].grad is not None:
                    head["d_" + nm] = frame.f_locals[nm].grad.detach().numpy().copy()

        def on_pre(frame):
            for nm in ("q", "logit"):
                if nm in frame.f_locals and frame.f_locals[nm].requires_grad:
                    frame.f_locals[nm].retain_grad()

        tap.on_line["step"] = on_step
        tap.on_line["pre_step"] = on_pre
        if "pre_clip" in markers:
            tap.on_line["pre_clip"] = lambda frame: gpre.update({k: p.grad.detach().numpy().copy() for k, p in agent.network.named_parameters()})

        net64 = tgt64 = None
        if opt.get("grad64"):  # float64 twins of both networks (before the step) for a ground-truth gradient
            import copy

            net64, tgt64 = copy.deepcopy(agent.network).double(), copy.deepcopy(agent.target_network).double()
        np.random.seed(42)
        torch.manual_seed(42)
        with tap:
            result = agent.learn()
        rec = tap.records["pre_step"][0]
        if net64 is not None:
            # ApeX.learn (ape_x.py:79-122) restated in float64 on the SAME sampled rows and IS weights: how far the reference's own
            # fp32 gradient is from the exact one (a K = B x 400 = 204 800-term reduction for conv1 at B = 512) -- the tests accept
            # |ours - exact| <= max(1e-5 of the tensor's largest entry, 2 x |reference fp32 - exact|)
            N_ = agent.memory.buffer_size
            leaf = rec["indices"].astype(np.int64) - (N_ - 1)
            st64 = torch.from_numpy(np.concatenate([agent.memory.buffer[i]["state"] for i in leaf], 0)).double()
            ns64 = torch.from_numpy(np.concatenate([agent.memory.buffer[i]["next_state"] for i in leaf], 0)).double()
            a64 = torch.from_numpy(rec["action"]).view(-1).long()
            r64, d64 = torch.from_numpy(rec["reward"]).double(), torch.from_numpy(rec["done"]).double()
            w64 = torch.from_numpy(np.asarray(rec["weights"], dtype=np.float32)).double().view(-1, 1)
            q64 = net64(st64).gather(1, a64.view(-1, 1))
            with torch.no_grad():
                ma = net64(ns64).argmax(1, keepdim=True)
                tq = tgt64(ns64).gather(1, ma)
                for i in reversed(range(agent.n_step)):
                    tq = r64[:, i] + (1 - d64[:, i]) * agent.gamma * tq
            loss64 = (w64 * (tq - q64).abs() ** 2).mean()
            loss64.backward()
            out["loss64"] = np.asarray(loss64.item())
            for k, p64 in net64.named_parameters():
                out[f"grad64_thin/{k}"] = synth.thin(p64.grad.numpy())
        flat("learn/", rec, out)
        flat("learn/", head, out)
        if recipe:
            if gpre:  # gradients BEFORE clip_grad_norm_; graw (taken at optimizer.step()) holds the clipped ones
                out["grad_clip_norm"] = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in graw.values()))
                graw = gpre
            flat("grad_thin/", {k: synth.thin(v) for k, v in graw.items()}, out)
            out["grad_norm"] = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in graw.values()))
            for k, v in graw.items():
                out[f"grad_absmax/{k}"] = np.abs(v).max()
            flat("sd0_thin/", {k: synth.thin(v) for k, v in sd0.items()}, out)
            flat("sdt_thin/", {k: synth.thin(v) for k, v in sdt.items()}, out)
            flat("sd1_thin/", {k: synth.thin(v) for k, v in sd_to_np(agent.network.state_dict()).items()}, out)
            # optimizer moments after the step (Adam exp_avg / exp_avg_sq, RMSprop square_avg / grad_avg), per parameter name
            for st_key in opt.get("opt_state", ()):
                for k, p in agent.network.named_parameters():
                    out[f"opt1_thin/{st_key}/{k}"] = synth.thin(agent.optimizer.state[p][st_key].detach().numpy())
        else:
            flat("grad/", graw, out)
            flat("sd0/", sd0, out)
            flat("sdt/", sdt, out)
            flat("sd1/", sd_to_np(agent.network.state_dict()), out)
        for k, v in result.items():
            out[f"result/{k}"] = np.asarray(v)
        if is_per:
            out["tree1"] = agent.memory.sum_tree.copy()
            out["maxp1"] = np.asarray(agent.memory.max_priority)
        hyper = dict(gamma=0.99, lr=kw["optim_config"]["lr"], B=B, S=np.asarray(S), A=A, H=H, np_seed=42, torch_seed=42)
        hyper.update({k: v for k, v in extra.items() if isinstance(v, (int, float))})
        hyper.update({f"optim_{k}": v for k, v in kw["optim_config"].items() if isinstance(v, (int, float, bool))})
        for k, v in hyper.items():
            out[f"hyper/{k}"] = np.asarray(v)
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **out)
        print(name, {k: float(v) for k, v in result.items()})


def gen_nstep(out_dir):
    """n-step assemblers (rainbow.py:294-308, multistep.py:90-104, ape_x.py:174-199)."""
    from core.agent.rainbow import Rainbow
    from core.agent.ape_x import ApeX

    S, A = 3, 2
    out = {}
    rng = np.random.RandomState(8)
    raw = []
    for i in range(12):
        raw.append(
            {
                "state": rng.randn(1, S).astype(np.float32),
                "action": rng.randint(0, A, size=(1, 1)),
                "reward": rng.randn(1, 1),
                "next_state": rng.randn(1, S).astype(np.float32),
                "done": np.asarray([[i in (4, 9)]]),
                "q": rng.randn(1, 1).astype(np.float32),
            }
        )
    for k in raw[0]:
        out[f"in_{k}"] = np.concatenate([t[k] for t in raw], 0)
    rb = Rainbow(state_size=S, action_size=A, hidden_size=8, n_step=3, buffer_size=16, device="cpu")
    emitted = []
    for t in raw:
        t2 = {k: v for k, v in t.items() if k != "q"}
        e = rb.interact_callback(t2)
        emitted.append(bool(e))
        if e:
            for k, v in e.items():
                out.setdefault(f"rainbow_{k}", []).append(v)
    out["rainbow_emitted"] = np.asarray(emitted)
    ax = ApeX(state_size=S, action_size=A, hidden_size=8, n_step=3, buffer_size=16, num_workers=4, device="cpu")
    emitted = []
    for t in raw:
        e = ax.interact_callback(dict(t))
        emitted.append(bool(e))
        if e:
            for k, v in e.items():
                out.setdefault(f"apex_{k}", []).append(np.asarray(v))
    out["apex_emitted"] = np.asarray(emitted)
    for k in list(out):
        if isinstance(out[k], list):
            out[k] = np.concatenate(out[k], 0)
    np.savez(os.path.join(out_dir, "nstep.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)

    scratch = tempfile.mkdtemp(prefix="jref_")
    subprocess.check_call(
        f"cd {args.ref} && tar --exclude='jorldy/core/env/mlagents' -cf - jorldy | (cd {scratch} && tar xf -)",
        shell=True,
    )
    os.chdir(os.path.join(scratch, "jorldy"))
    sys.path.insert(0, os.getcwd())
    sys.dont_write_bytecode = True
    import torch

    torch.set_num_threads(1)  # deterministic reductions in the fixtures
    try:
        todo = args.only.split(",") if args.only else ["buffers", "ppo", "dqn", "nstep"]  # + rainbow_cnn, rainbow_cnn_atari, dqn_h512, ape_x_cnn_atari on request
        if "buffers" in todo:
            gen_buffers(out_dir)
        if "ppo" in todo:
            gen_ppo(out_dir)
        ppo_only = [t for t in todo if t.startswith("ppo_")]
        if ppo_only:
            gen_ppo(out_dir, only=ppo_only)
        if "dqn" in todo:
            gen_dqn_family(out_dir)
        if "nstep" in todo:
            gen_nstep(out_dir)
        for nm in ("rainbow_cnn", "rainbow_cnn_atari", "dqn_h512", "ape_x_cnn_atari", "ape_x_cnn_clip"):
            if nm in todo:
                gen_dqn_family(out_dir, only=nm)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    print("golden fixtures written to", out_dir)


if __name__ == "__main__":
    main()
