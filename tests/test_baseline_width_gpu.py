"""Parity at the BASELINE.json widths, against runs of the unmodified reference (oracle/gen_golden.py):

  ppo_disc_cartpole_h512   config.ppo.cartpole exactly: 4-512-512-{2,1}, 8 x 128 rows, minibatch 256, 3 epochs
  ppo_cont_hopper_real     config.ppo.mujoco Hopper shapes: 11-512-512-{3,3,1}, T = 2048, minibatch 2048
  ppo_cont_halfcheetah     config.ppo.mujoco HalfCheetah shapes: 17-512-512-{6,6,1} (13 head outputs), minibatch 1024
  ppo_cont_ant_mb256       config.ppo.mujoco Ant shapes: 27-512-512-{8,8,1} (17 head outputs), minibatch 256
                           -> the LDS-tiled engine (jh_tgemm_ppo_fwd_h2 / _bwd / _bwd_dW1) that minibatches
                           >= 1024 rows switch to
  rainbow_cnn_atari        config.rainbow.atari exactly: (4,84,84) uint8 frames, A = 4, B = 32, hidden 512

Initial weights and frames are regenerated from seeds (oracle/synth.py) -- the same values the generator
wrote into the reference agent -- and pinned by checksums / strided samples stored in the fixture.
Tolerances: north_star's 1e-5 on every loss of EVERY update (|ours - ref| / (1 + |ref|)) and on the gradients
of the first update (relative to the largest gradient entry of the tensor); the measured per-update drift is
written to gpurun_out/parity_drift_<fixture>.json (committed copies: profiles/r02_parity_drift_*.json).
"""
import json
import os

import numpy as np

import margins
import pytest
import torch

from oracle import synth
from tests.util import load, npy

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(name, payload):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"parity_drift_{name}.json"), "w") as f:
        json.dump(payload, f, indent=1)


def _thin_cmp(ours, z, prefix, scale_of=None, tol=1e-5, what=""):
    """ours: {name: array}; fixture holds synth.thin(ref) under prefix+name.  |diff| <= tol * scale where
    scale = the tensor's largest |reference entry| (stored when thinned, else computed)."""
    worst = {}
    for k, v in ours.items():
        ref = z[prefix + k]
        got = synth.thin(np.asarray(v))
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        scale = float(scale_of(k)) if scale_of else float(np.abs(ref).max())
        err = float(np.abs(got - ref).max()) / (scale + 1e-30)
        worst[k] = err
        margins.leq(err, tol, f"{what} {k}: max |diff| / the tensor's largest entry")
    return worst


def _ppo_agent(z, **kw):
    from jorldy_amd.core.agent import Agent

    S, A, H, W, T, B, E, cont = [int(x) for x in z["cfg"]]
    gamma, lam, eps, vf, ent, clip, lr = z["hyper"]
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=H, network="continuous_policy_value" if cont else "discrete_policy_value",
                  optim_config={"name": "adam", "lr": lr}, batch_size=B, n_step=T, n_epoch=E, _lambda=lam, epsilon_clip=eps, vf_coef=vf,
                  ent_coef=ent, clip_grad_norm=clip, gamma=gamma, run_step=100000, num_workers=W, device="cuda", backend="native", **kw)
    assert agent.backend == "native"
    rec = synth.ppo_recipe({k: v.shape for k, v in agent.network.state_dict().items()}, int(z["recipe_seed"]))
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in rec.items()})
    _thin_cmp({k: npy(v) for k, v in agent.network.state_dict().items()}, z, "sd0_thin/", tol=0.0, what="initial weights")
    M = W * T
    trs = synth.ppo_rollout(np.random.RandomState(int(z["rollout_seed"])), M, S, A, bool(cont), clamp_every=0)
    cols = {k: np.concatenate([t[k] for t in trs], 0) for k in ("state", "next_state", "reward", "done", "action")}
    for k in ("state", "reward", "action"):
        assert np.array_equal(synth.row_checksum(cols[k].astype(np.float32))[:: max(1, M // 64)], z[f"in_{k}_check"]), k
    agent.memory.first_store = False
    return agent, cols, (S, A, H, W, T, B, E, cont), float(lr)


# ppo_cont_halfcheetah / ppo_cont_ant_mb256 (round 5): config.ppo.mujoco on its other envs -- 13 / 17 head outputs (> 8: the separate
# forward / backward calls and the tiled engine; VERDICT r4 missing #6) at minibatches of 1024 (tiled) and 256 (latency kernels) rows
PPO_WIDE = ["ppo_disc_cartpole_h512", "ppo_cont_hopper_real", "ppo_cont_halfcheetah", "ppo_cont_ant_mb256"]


def _ppo_head_grads_float64(z, cont, eps, vf, ent, actions, heads=None):
    """d(loss)/d(raw heads) of minibatch 0 in float64: core/agent/ppo.py:125-165 restated with torch autograd on the CPU, fed the
    reference's own head values (or `heads`: {name: array}) and upstream quantities from the fixture (test infrastructure: the comparator,
    not the product)."""
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    hv = lambda k: t(heads[k] if heads is not None else z[f"mb0/head/{k}"])
    idx = z["mb0/idx"].astype(np.int64)
    adv, ret, v_old, lp_old = (t(z[f"gae/{k}"])[idx] for k in ("adv", "ret", "value", "log_prob_old"))
    action = t(actions)[idx]
    v = hv("v").reshape(-1, 1).requires_grad_(True)
    if cont:
        mu_raw, ls_raw = hv("mu_raw").requires_grad_(True), hv("log_std_raw").requires_grad_(True)
        m = torch.distributions.Normal(torch.clamp(mu_raw, -5.0, 5.0), torch.tanh(ls_raw).exp())  # policy_value.py:52-56
        log_prob = m.log_prob(torch.atanh(torch.clamp(action, -1 + 1e-7, 1 - 1e-7)))
        leaves = {"mu_raw": mu_raw, "log_std_raw": ls_raw, "v": v}
    else:
        logits = hv("logits").requires_grad_(True)
        pi = torch.exp(torch.log_softmax(logits, dim=-1))
        m = torch.distributions.Categorical(pi)
        log_prob = m.log_prob(action.squeeze(-1).long()).unsqueeze(-1)
        leaves = {"logits": logits, "v": v}
    ratio = (log_prob - lp_old).sum(1, keepdim=True).exp()
    actor = -torch.min(ratio * adv, torch.clamp(ratio, 1 - eps, 1 + eps) * adv).mean()
    v_clip = v_old + torch.clamp(v - v_old, -eps, eps)
    critic = torch.max(torch.nn.functional.mse_loss(v, ret), torch.nn.functional.mse_loss(v_clip, ret))
    loss = actor + vf * critic + ent * (-m.entropy().mean())
    loss.backward()
    return {k: x.grad.numpy() for k, x in leaves.items()}


@pytest.mark.parametrize("name", PPO_WIDE)
def test_ppo_first_minibatch_forward_loss_backward(name):
    """Minibatch 0 of the reference's run, step by step through the C ABI: no-grad passes + GAE, the minibatch
    forward, the clipped loss (fwd+bwd), the backward into the flat gradient bucket.  For ppo_cont_hopper_real
    (2048 rows) this is the tiled path: jh_tgemm_ppo_fwd_h2, jh_tgemm_ppo_bwd (grouped dW2 | dh1 | heads), _bwd_dW1."""
    from jorldy_amd import ops

    z = load(name)
    agent, cols, (S, A, H, W, T, B, E, cont), lr = _ppo_agent(z, use_graph=False)
    gamma, lam, eps, vf, ent, clip, _ = [float(v) for v in z["hyper"]]
    net = agent._net
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    state, nstate, action = dev(cols["state"]), dev(cols["next_state"]), dev(cols["action"])
    reward, done = dev(cols["reward"]), dev(cols["done"])
    agent._grow_native(W * T)
    net = agent._net
    outs_n = net.forward(nstate)
    next_value = outs_n[-1].clone()
    outs = [o.clone() if o is not None else None for o in net.forward(state)]
    value = outs[-1]
    np.testing.assert_allclose(npy(value), z["gae/value"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(npy(next_value), z["gae/next_value"], rtol=1e-5, atol=1e-5)
    logp_old = ops.logp_continuous(outs[0], outs[1], action) if cont else ops.logp_discrete(outs[0], action)
    np.testing.assert_allclose(npy(logp_old), z["gae/log_prob_old"], rtol=1e-5, atol=1e-5)
    adv, ret = ops.gae(reward, done, value, next_value, T, gamma, lam, True)
    np.testing.assert_allclose(npy(adv), z["gae/adv"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(npy(ret), z["gae/ret"], rtol=1e-5, atol=1e-5)
    # minibatch 0 with the reference's own values upstream (isolates this step)
    idx = torch.from_numpy(z["mb0/idx"].astype(np.int64)).cuda()
    adv_r, ret_r, v_r, lp_r = dev(z["gae/adv"]), dev(z["gae/ret"]), dev(z["gae/value"]), dev(z["gae/log_prob_old"])
    stats = torch.zeros(8, device="cuda")
    if cont:
        mu, ls, vp = net.forward(state, idx=idx)
        np.testing.assert_allclose(npy(mu), z["mb0/head/mu_raw"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(npy(ls), z["mb0/head/log_std_raw"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(npy(vp), z["mb0/head/v"], rtol=1e-5, atol=1e-5)
        g_mu, g_ls, g_v, _ = ops.ppo_loss_continuous(mu, ls, vp, idx, action, adv_r, ret_r, v_r, lp_r, eps, vf, ent, stats=stats)
        heads = {"mu_raw": g_mu, "log_std_raw": g_ls, "v": g_v}
        net.backward(state, idx, g_mu, g_ls, g_v)
    else:
        zz, vp = net.forward(state, idx=idx)
        np.testing.assert_allclose(npy(zz), z["mb0/head/logits"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(npy(vp), z["mb0/head/v"], rtol=1e-5, atol=1e-5)
        g_z, g_v, _ = ops.ppo_loss_discrete(zz, vp, idx, action, adv_r, ret_r, v_r, lp_r, eps, vf, ent, stats=stats)
        heads = {"logits": g_z, "v": g_v}
        net.backward(state, idx, g_z, None, g_v)
    s = npy(stats)
    for j, k in enumerate(("loss", "actor_loss", "critic_loss", "entropy_loss")):
        np.testing.assert_allclose(s[j], z[f"mb0/{k}"], rtol=1e-5, atol=1e-5, err_msg=k)
    # The loss kernel by ITSELF, fed the reference's own head values (the fixture's) -- d(loss)/d(log_std_raw) cancels (z - mu)^2 / (var std)
    # against 1 / std, so a head value off by 1e-6 (ours vs the reference's forward) moves it by 1e-5 of its largest entry: an input
    # difference, not the kernel's.  Truth: ppo.py:125-165 re-evaluated with torch autograd in float64 on the same head values; the
    # reference's fp32 gradient (fixture) is itself 3e-6 from it.  |ours - exact| <= max(1e-5, 2 x |reference - exact|) of the largest entry.
    exact = _ppo_head_grads_float64(z, cont, eps, vf, ent, cols["action"])
    st2 = torch.zeros(8, device="cuda")
    if cont:
        iso = ops.ppo_loss_continuous(dev(z["mb0/head/mu_raw"]), dev(z["mb0/head/log_std_raw"]), dev(z["mb0/head/v"]), idx, action, adv_r, ret_r, v_r, lp_r, eps, vf, ent, stats=st2)
        iso = {"mu_raw": iso[0], "log_std_raw": iso[1], "v": iso[2]}
    else:
        iso = ops.ppo_loss_discrete(dev(z["mb0/head/logits"]), dev(z["mb0/head/v"]), idx, action, adv_r, ret_r, v_r, lp_r, eps, vf, ent, stats=st2)
        iso = {"logits": iso[0], "v": iso[1]}
    for tag, g in iso.items():
        ref, ex = z[f"mb0/head/d_{tag}"], exact[tag]
        scale = float(np.abs(ex).max())
        e_ref = float(np.abs(ref.reshape(ex.shape) - ex).max()) / scale
        margins.leq(float(np.abs(npy(g).reshape(ex.shape) - ex).max()) / scale, max(1e-5, 2.0 * e_ref), f"loss kernel alone: d(loss)/d({tag}) vs float64 (reference's own fp32: {e_ref:.2e})")
    # ... and inside the pipeline, where its inputs are OUR forward's heads: against the float64 gradient AT THOSE heads
    ours_heads = {"mu_raw": npy(mu), "log_std_raw": npy(ls), "v": npy(vp)} if cont else {"logits": npy(zz), "v": npy(vp)}
    exact_ours = _ppo_head_grads_float64(z, cont, eps, vf, ent, cols["action"], heads=ours_heads)
    for tag, g in heads.items():
        ex = exact_ours[tag]
        margins.leq(float(np.abs(npy(g).reshape(ex.shape) - ex).max()) / float(np.abs(ex).max()), 1e-5, f"pipeline: d(loss)/d({tag}) vs float64 at our own heads")
    grads = {k: npy(p.grad) for k, p in agent.network.named_parameters()}
    worst = _thin_cmp(grads, z, "mb0/grad_raw/", scale_of=lambda k: z[f"mb0/grad_raw_absmax/{k}"], tol=1e-5, what="d(loss)/d")
    norm = float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    np.testing.assert_allclose(norm, float(z["mb0/grad_raw_norm"]), rtol=1e-5)
    _report(name + "_mb0", {"grad_err_rel_to_absmax": worst, "grad_norm": norm, "grad_norm_ref": float(z["mb0/grad_raw_norm"])})


# bound on |ours - ref| / (1 + |ref|) for the four loss scalars of EVERY update: north_star's 1e-5.  Measured
# (profiles/r02_parity_drift_*.json): 1.3e-7 at update 0, <= 3.6e-7 after 4 Hopper / 12 CartPole updates.
def _bound(i):
    return 1e-5


def _check_every_adam_step(agent, lr, name, M):
    """The optimizer step of EVERY minibatch update of a learn() checked as arithmetic, teacher-forced (VERDICT r4 weak #1 ii: the float64
    optimizer check existed for step 1 only): the eager path's update calls are wrapped -- state before, the call, state after -- and
    float64 Adam (torch.optim.Adam's formulas: exp_avg.lerp_, bias corrections from the step count) fed OUR state before and OUR
    clipped gradient must land on our weights / moments to one fp32 rounding (tests/fp64_truth.py's per-element criterion)."""
    agent._grow_native(2 * M if 2 * M <= 8192 else M)  # (learn() would grow -- i.e. replace -- the net it is about to use: wrap the final one)
    net = agent._net
    seen = {"n": 0}
    b1, b2, eps = 0.9, 0.999, 1e-8

    def wrap(fn):
        def call(*a, **kw):
            w0, m0, v0 = net.params.double().cpu(), net.m.double().cpu(), net.v.double().cpu()
            out = fn(*a, **kw)
            torch.cuda.synchronize()
            t = seen["n"] + 1
            g = net.grads.double().cpu()  # what the bucket holds after the step: the CLIPPED gradient (clip_grad_norm_ is in place in the reference too)
            m1 = m0 + (1.0 - b1) * (g - m0)
            v1 = b2 * v0 + (1.0 - b2) * g * g
            dw = (lr / (1.0 - b1 ** t)) * m1 / (v1.sqrt() / (1.0 - b2 ** t) ** 0.5 + eps)
            w1 = w0 - dw
            e_w = (net.params.double().cpu() - w1).abs() - (2.0 ** -22 * w1.abs() + 1e-4 * dw.abs() + 1e-6 * lr)
            margins.leq(float(e_w.max()) + 1.0, 1.0, f"{name} update {t}: stepped weights vs float64 Adam on our own state and gradient (excess over one fp32 rounding, + 1)")
            margins.leq(float((net.m.double().cpu() - m1).abs().max()), 1e-5 * float(m1.abs().max()) + 1e-30, f"{name} update {t}: exp_avg")
            margins.leq(float((net.v.double().cpu() - v1).abs().max()), 1e-5 * float(v1.abs().max()) + 1e-30, f"{name} update {t}: exp_avg_sq")
            seen["n"] = t
            return out

        return call

    # the four-launch update steps inside ppo_update; the separate-call path (minibatches >= 1024 rows, more than 8 head outputs) in adam_step
    if agent.fused_update and net.fused_ok(agent.batch_size):
        net.ppo_update = wrap(net.ppo_update)
    elif os.environ.get("JH_PPO_ONEPASS", "1") == "1":
        net.ppo_update_rows = wrap(net.ppo_update_rows)  # round 6: one call per update (the loss in one launch); Adam steps inside it
    else:
        net.adam_step = wrap(net.adam_step)
    return seen


@pytest.mark.parametrize("name", PPO_WIDE)
@pytest.mark.parametrize("graph", [False, True])
def test_ppo_learn_at_baseline_width(name, graph):
    z = load(name)
    agent, cols, (S, A, H, W, T, B, E, cont), lr = _ppo_agent(z, use_graph=graph)
    n_upd = int(z["n_minibatch"])
    sd0 = {k: v.clone() for k, v in agent.network.state_dict().items()}
    for rep in range(3 if graph else 1):  # graph: eager warm-up, capture + replay, replay
        agent.network.load_state_dict(sd0)
        agent._net.m.zero_()
        agent._net.v.zero_()
        agent._adam_steps = 0
        agent._net.set_hyper(lr, 0.9, 0.999, 1e-8, step=0.0)
        agent.time_t, agent.learn_stamp = 0, 0
        np.random.seed(int(z["np_seed"]))
        checked = _check_every_adam_step(agent, lr, name, W * T) if not graph else None
        result = agent.process(cols, T)
        if checked is not None:
            assert checked["n"] == n_upd, (checked["n"], n_upd)
        if graph and rep >= 1:
            assert agent._graph is not None
        s = npy(agent._stats[:n_upd]).astype(np.float64)
        drift = []
        for i in range(n_upd):
            e = max(abs(s[i, j] - float(z[f"mb{i}/{k}"])) / (1.0 + abs(float(z[f"mb{i}/{k}"])))
                    for j, k in enumerate(("loss", "actor_loss", "critic_loss", "entropy_loss")))
            drift.append(e)
        _report(f"{name}_{'graph' if graph else 'eager'}", {"per_update_max_err_over_1_plus_abs_ref": drift, "bound": [_bound(i) for i in range(n_upd)]})
        for i, e in enumerate(drift):
            margins.leq(e, _bound(i), f"{name} rep {rep} update {i} loss scalars")
        for k in ("actor_loss", "critic_loss", "entropy_loss", "mean_ret"):
            np.testing.assert_allclose(result[k], z[f"result/{k}"], rtol=2e-5, atol=2e-5, err_msg=k)
        np.testing.assert_allclose(agent.optimizer.param_groups[0]["lr"], z["lr_after"], rtol=1e-12)
        # clipped gradients of the LAST minibatch are what the bucket holds after learn()
        last = n_upd - 1
        grads = {k: npy(p.grad) for k, p in agent.network.named_parameters()}
        _thin_cmp(grads, z, f"mb{last}/grad_clip/", scale_of=lambda k: z[f"mb{last}/grad_raw_absmax/{k}"], tol=2e-4, what="last clipped gradient")
        # updated weights: Adam moves a weight by ~lr per step whatever its gradient, so weights with a ~0
        # gradient may land a step apart; 99.5 % within 2e-5, none further than the possible travel
        tot = bad = 0
        worst = 0.0
        for k, v in agent.network.state_dict().items():
            d = np.abs(synth.thin(npy(v)) - z[f"sd1_thin/{k}"])
            tot += d.size
            bad += int((d > 2e-5).sum())
            worst = max(worst, float(d.max()))
        margins.leq(bad / tot, 0.005, "fraction of weights off")
        margins.leq(worst, 2.1 * lr * n_upd, "worst weight difference vs travel")


@pytest.mark.parametrize("name", ["ppo_cont_hopper_real", "ppo_cont_halfcheetah", "ppo_cont_ant_mb256"])
def test_ppo_one_launch_loss_update_is_bit_identical_to_the_separate_calls(name, monkeypatch):
    """jh_pponet_ppo_update_rows (round 6: the loss forward + backward in ONE launch for any minibatch size -- every workgroup keeps both critic branches' value
    gradients, the last one to arrive reduces the partials, the backward's first kernel mixes) against the path it replaces (jh_pponet_forward -> the two-pass
    jh_ppo_fwd / jh_ppo_bwd kernels -> jh_pponet_backward -> jh_pponet_adam_step): the same statistics of every update and the same weights and moments after a
    whole learn(), bit for bit -- 2048-row minibatches (8 workgroups), 1024 rows with 13 head outputs (the value head in the second heads launch), 256 rows with 17."""
    z = load(name)
    out = {}
    monkeypatch.setenv("JH_PPO_NORM_FOLD", "0")  # (the folded norm sums the squares in another order: its own test below)
    for mode in ("0", "1"):
        monkeypatch.setenv("JH_PPO_ONEPASS", mode)
        agent, cols, (S, A, H, W, T, B, E, cont), lr = _ppo_agent(z, use_graph=False)
        np.random.seed(int(z["np_seed"]))
        agent.process(cols, T)
        torch.cuda.synchronize()
        n_upd = int(z["n_minibatch"])
        out[mode] = (npy(agent._stats[:n_upd]).copy(), npy(agent._net.params).copy(), npy(agent._net.m).copy(), npy(agent._net.v).copy(), npy(agent._net.grads).copy())
    for a, b, what in zip(out["0"], out["1"], ("statistics of every update", "weights", "exp_avg", "exp_avg_sq", "last clipped gradient")):
        assert np.array_equal(a, b), f"{name}: {what} differ between the separate calls and the one-launch loss (max |diff| {np.abs(a - b).max():.3e})"


@pytest.mark.parametrize("name", ["ppo_cont_hopper_real", "ppo_cont_halfcheetah"])
def test_ppo_norm_folded_into_the_dw1_launches_equals_the_norm_kernel(name, monkeypatch):
    """clip_grad_norm_'s sum of squares without a launch of its own (round 6): the part of the bucket the grouped GEMM wrote is squared by extra workgroups of
    the dW1 column reduction's first launch, the rest by the combine kernel that forms those elements.  Another order of the same additions: the clip
    coefficient agrees to fp32 rounding, and so does a whole learn() -- statistics of every update to 1e-6, weights to 1e-7 + a millionth of a step."""
    z = load(name)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("JH_PPO_NORM_FOLD", mode)
        agent, cols, (S, A, H, W, T, B, E, cont), lr = _ppo_agent(z, use_graph=False)
        np.random.seed(int(z["np_seed"]))
        agent.process(cols, T)
        torch.cuda.synchronize()
        n_upd = int(z["n_minibatch"])
        out[mode] = (npy(agent._stats[:n_upd]).astype(np.float64), npy(agent._net.params).astype(np.float64), npy(agent._net.grads).astype(np.float64))
    s0, w0, g0 = out["0"]
    s1, w1, g1 = out["1"]
    assert not np.array_equal(g0, g1) or np.array_equal(w0, w1)  # (the clipped gradient carries the coefficient: usually a few ulps apart)
    margins.leq(float(np.abs(s0 - s1).max() / (1.0 + np.abs(s0).max())), 1e-6, f"{name}: statistics, folded norm vs norm kernel")
    margins.leq(float(np.abs(g0 - g1).max() / np.abs(g0).max()), 1e-6, f"{name}: last clipped gradient, folded norm vs norm kernel")
    margins.leq(float(np.abs(w0 - w1).max()), 1e-7 + 1e-2 * lr, f"{name}: weights after a learn(), folded norm vs norm kernel")


def test_rainbow_learn_at_atari_shapes():
    """config.rainbow.atari: one Rainbow.learn() of the reference on (4,84,84) uint8 frames, B = 32, hidden 512,
    A = 4 (conv 8/4, 4/2, 3/1 -> 3136 -> 512 -> noisy 512 x 2 -> {4 x 51, 51}) vs the native network + PER + C51
    kernels.  Sampled indices and IS weights bit-exact; losses, KL, priorities, logits within 1e-5; every
    parameter gradient within 1e-5 of the tensor's largest entry."""
    from jorldy_amd.core.agent import Agent

    z = load("rainbow_cnn_atari")
    h = lambda k: z[f"hyper/{k}"].item()
    H, A, K, B, n = int(h("H")), int(h("A")), int(h("num_support")), int(h("B")), int(h("n_step"))
    S = tuple(int(v) for v in z["hyper/S"])
    agent = Agent("rainbow", state_size=S, action_size=A, hidden_size=H, head="cnn", optim_config={"name": "adam", "lr": h("lr")},
                  gamma=h("gamma"), buffer_size=64, batch_size=B, start_train_step=0, target_update_period=10000, run_step=100000,
                  n_step=n, alpha=h("alpha"), beta=h("beta"), learn_period=1, uniform_sample_prob=h("uniform_sample_prob"),
                  v_min=h("v_min"), v_max=h("v_max"), num_support=K, device="cuda", backend="native", use_graph=False)
    assert agent.backend == "native"
    shapes = {k: v.shape for k, v in agent.network.state_dict().items()}
    seed = int(z["recipe_seed"])
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, seed).items()})
    agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, seed + 1).items()})
    _thin_cmp({k: npy(v) for k, v in agent.network.state_dict().items()}, z, "sd0_thin/", tol=0.0, what="initial weights")
    _thin_cmp({k: npy(v) for k, v in agent.target_network.state_dict().items()}, z, "sdt_thin/", tol=0.0, what="target weights")
    # frames: the env steps the generator drew; slot i = (raw[i].state, raw[i+n-1].next_state) (rainbow.py:294-308)
    rng = np.random.RandomState(int(z["fill_seed"]))
    raw = [synth.raw_transition(rng, S, A) for _ in range(int(z["fill"]))]
    n_rows = len(raw) - n + 1
    cols = {"state": np.concatenate([raw[i]["state"] for i in range(n_rows)], 0),
            "next_state": np.concatenate([raw[i + n - 1]["next_state"] for i in range(n_rows)], 0)}
    for k in ("state", "next_state"):
        assert np.array_equal(synth.row_checksum(cols[k]), z[f"buf_{k}_check"]), k
    for k in ("action", "reward", "done"):
        cols[k] = z[f"buf_{k}"]
    agent.memory.first_store = False
    agent.memory.store_soa(cols)
    agent.memory._tree.load(z["tree0"], float(np.asarray(z["maxp0"]).reshape(-1)[0]), int(z["tree_index0"]), n_rows)
    assert agent.memory._store.column("state").dtype == torch.uint8
    torch.manual_seed(int(h("torch_seed")))
    noise = []
    for _ in range(3):  # network(state), network(next_state), target_network(next_state): utils.py:58-60 draw order
        d = {}
        for tag, (i, o) in (("a1", (H, H)), ("v1", (H, H)), ("a2", (H, K * A)), ("v2", (H, K))):
            d[tag] = (torch.randn(i).cuda(), torch.randn(o).cuda())
        noise.append(d)
    agent._noise = noise
    np.random.seed(int(h("np_seed")))
    result = agent.learn()
    errs = {}
    for k in ("loss", "max_Q", "max_logit", "min_logit"):
        errs[k] = abs(result[k] - float(z[f"result/{k}"])) / (1.0 + abs(float(z[f"result/{k}"])))
        np.testing.assert_allclose(result[k], z[f"result/{k}"], rtol=1e-5, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(result["sampled_p"], z["result/sampled_p"], rtol=1e-12)
    assert result["mean_p"] == z["result/mean_p"].item()
    st = agent._static
    assert np.array_equal(npy(st["idx"]), z["learn/indices"].astype(np.int64)), "sampled tree indices"
    assert np.array_equal(npy(st["w"]), z["learn/weights"].astype(np.float32).reshape(-1)), "IS weights (fp32, as_tensor)"
    np.testing.assert_allclose(npy(st["logits"][0]), z["learn/logit"], rtol=1e-5, atol=1e-5, err_msg="online logits")
    errs["logit_max_abs"] = float(np.abs(npy(st["logits"][0]) - z["learn/logit"]).max())
    # tree after the write-back of OUR fp32 priorities KL^alpha
    np.testing.assert_allclose(agent.memory.sum_tree, z["tree1"], rtol=2e-5, atol=1e-6)
    grads = {k: v.cpu().numpy() for k, v in agent._net.export_state(agent._net.grads).items()}
    worst = _thin_cmp(grads, z, "grad_thin/", scale_of=lambda k: z[f"grad_absmax/{k}"], tol=1e-5, what="d(loss)/d")
    norm = float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    np.testing.assert_allclose(norm, float(z["grad_norm"]), rtol=1e-5)
    _report("rainbow_cnn_atari", {"result_err_over_1_plus_abs_ref": errs, "grad_err_rel_to_absmax": worst, "grad_norm": norm, "grad_norm_ref": float(z["grad_norm"])})
    lr = float(h("lr"))
    tot = bad = 0
    for k, v in agent.network.state_dict().items():
        d = np.abs(synth.thin(npy(v)) - z[f"sd1_thin/{k}"])
        tot += d.size
        bad += int((d > 0.05 * lr).sum())
        margins.leq(float(d.max()), 2.1 * lr, f"worst weight difference {k}")
    margins.leq(bad / tot, 0.005, "fraction of weights off")


# ---------------------------------------------------------------------------------------------------------
# configs[0] and configs[3] at their real shapes (VERDICT r2 "missing" #2): DQN H=512 / B=32 / Adam 1e-4 and the Ape-X
# learner = dueling CNN on (4,84,84) uint8, A=6, B=512, n=3, centered RMSprop eps 1.5e-7, clip_grad_norm 40; plus the
# same learner on a small image with the clip ACTIVE (norm 31.8 -> 0.5).  Fixtures: runs of the unmodified reference.
# ---------------------------------------------------------------------------------------------------------
TD_WIDE = [("dqn_h512", "dqn"), ("ape_x_cnn_atari", "ape_x"), ("ape_x_cnn_clip", "ape_x")]


def _td_wide_agent(z, agent_name, use_graph):
    from jorldy_amd.core.agent import Agent

    h = lambda k: z[f"hyper/{k}"].item()
    cnn = z["hyper/S"].ndim > 0
    S = tuple(int(v) for v in z["hyper/S"]) if cnn else int(h("S"))
    A, H, B = int(h("A")), int(h("H")), int(h("B"))
    oc = {"name": "rmsprop" if "hyper/optim_centered" in z.files else "adam", "lr": h("optim_lr")}
    if oc["name"] == "rmsprop":
        oc.update(eps=h("optim_eps"), centered=bool(h("optim_centered")))
    extra = {k: h(k) for k in ("n_step", "alpha", "beta", "learn_period", "uniform_sample_prob", "num_workers", "clip_grad_norm") if f"hyper/{k}" in z.files}
    for k in ("n_step", "learn_period", "num_workers"):
        if k in extra:
            extra[k] = int(extra[k])
    if cnn:
        extra.update(network="dueling", head="cnn")
    n_rows = int(z["buf_action"].shape[0])
    agent = Agent(agent_name, state_size=S, action_size=A, hidden_size=H, optim_config=oc, gamma=h("gamma"), buffer_size=640 if B == 512 else (64 if cnn else 256),
                  batch_size=B, start_train_step=0, target_update_period=10000, run_step=100000, device="cuda", backend="native", use_graph=use_graph, **extra)
    assert agent.backend == "native"
    shapes = {k: v.shape for k, v in agent.network.state_dict().items()}
    seed = int(z["recipe_seed"])
    w0 = {k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, seed).items()}
    wt = {k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, seed + 1).items()}
    if cnn:  # slot i = (raw[i].state, raw[i + n].state): ape_x.py:174-199 (deque of n + 1, next_state = the newest STATE)
        n = extra["n_step"]
        rng = np.random.RandomState(int(z["fill_seed"]))
        raw = [synth.raw_transition(rng, S, A, True) for _ in range(int(z["fill"]))]
        assert n_rows == len(raw) - n
        cols = {"state": np.concatenate([raw[i]["state"] for i in range(n_rows)], 0),
                "next_state": np.concatenate([raw[i + n]["state"] for i in range(n_rows)], 0)}
        for k in ("state", "next_state"):
            assert np.array_equal(synth.row_checksum(cols[k]), z[f"buf_{k}_check"]), k
    else:
        cols = {k: z[f"buf_{k}"] for k in ("state", "next_state")}
    for k in ("action", "reward", "done"):
        cols[k] = z[f"buf_{k}"]
    agent.memory.first_store = False
    agent.memory.store_soa(cols)
    if cnn:
        assert agent.memory._store.column("state").dtype == torch.uint8
    return agent, w0, wt, n_rows, oc


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("fixture,agent_name", TD_WIDE)
def test_td_learn_at_baseline_shapes(fixture, agent_name, use_graph):
    """One learn() of the reference (DQN.learn dqn.py:117-151; ApeX.learn ape_x.py:79-133) vs the native network + HIP TD /
    PER kernels: sampled indices and IS weights bit-exact, q / loss / max_Q within 1e-5, every parameter gradient within
    1e-5 of the tensor's largest entry, the optimizer's moments after the step (Adam exp_avg / exp_avg_sq; centered RMSprop
    grad_avg / square_avg, after clip_grad_norm_).  use_graph: the third learn() (a hipGraph replay) is the one compared."""
    from jorldy_amd.core.optimizer import Optimizer

    z = load(fixture)
    h = lambda k: z[f"hyper/{k}"].item()
    agent, w0, wt, n_rows, oc = _td_wide_agent(z, agent_name, use_graph)
    per = "tree0" in z.files
    B, A = int(h("B")), int(h("A"))
    defaults = Optimizer(**oc, params=[torch.nn.Parameter(torch.zeros(1))]).defaults
    reps = 3 if use_graph else 1
    for rep in range(reps):
        agent.network.load_state_dict(w0)
        agent.target_network.load_state_dict(wt)
        agent._net.m.zero_()
        agent._net.v.zero_()
        agent._adam_steps = 0
        agent._set_native_hyper(defaults, 0)
        if per:
            agent.memory._tree.load(z["tree0"], float(np.asarray(z["maxp0"]).reshape(-1)[0]), int(z["tree_index0"]), n_rows)
        np.random.seed(int(h("np_seed")))
        result = agent.learn()
    if use_graph:
        assert agent._graph is not None, "learn() was not captured"
    _thin_cmp({k: npy(v) for k, v in w0.items()}, z, "sd0_thin/", tol=0.0, what="initial weights")
    _thin_cmp({k: npy(v) for k, v in wt.items()}, z, "sdt_thin/", tol=0.0, what="target weights")
    errs = {}
    for k in ("loss", "max_Q"):
        errs[k] = abs(result[k] - float(z[f"result/{k}"])) / (1.0 + abs(float(z[f"result/{k}"])))
        margins.leq(errs[k], 1e-5, f"result {k}")
    st = agent._static
    act = z["learn/action"].astype(np.int64).reshape(-1)
    q_all = npy(st["logits"][0]).reshape(B, A)
    q_ref = z["learn/q"].reshape(-1)
    errs["q_max_abs"] = float(np.abs(q_all[np.arange(B), act] - q_ref).max())
    np.testing.assert_allclose(q_all[np.arange(B), act], q_ref, rtol=1e-5, atol=1e-5, err_msg="Q(s, a)")
    if per:
        assert np.array_equal(npy(st["idx"]), z["learn/indices"].astype(np.int64)), "sampled tree indices"
        assert np.array_equal(npy(st["w"]), z["learn/weights"].astype(np.float32).reshape(-1)), "IS weights (fp32, as_tensor)"
        np.testing.assert_allclose(result["sampled_p"], z["result/sampled_p"], rtol=1e-12)
        assert result["mean_p"] == z["result/mean_p"].item()
        # our fp32 |td|^alpha written back: a TD error off by 1e-6 moves a SMALL priority by 0.6 |td|^-0.4 x that (8e-5 relative seen at
        # |td| ~ 1e-3), so leaves are compared as TD errors (north star: 1e-5), inner nodes with the matching absolute slack
        N = (z["tree1"].shape[0] + 1) // 2
        alpha = h("alpha")
        np.testing.assert_allclose(agent.memory.sum_tree[N - 1:] ** (1 / alpha), z["tree1"][N - 1:] ** (1 / alpha), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(agent.memory.sum_tree, z["tree1"], rtol=2e-5, atol=2e-5)
    # after the step the bucket holds what p.grad holds in the reference: the gradient scaled in place by clip_grad_norm_
    # (coefficient min(1, clip / (norm + 1e-6)), torch.nn.utils.clip_grad_norm_); the fixture stores the RAW gradient
    coef = 1.0
    if "grad_clip_norm" in z.files:
        coef = min(1.0, h("clip_grad_norm") / (float(z["grad_norm"]) + 1e-6))
    grads = {k: v.cpu().numpy() / np.float32(coef) for k, v in agent._net.export_state(agent._net.grads).items()}
    vs64 = {}
    if any(k.startswith("grad64_thin/") for k in z.files):
        # the fixture also holds the EXACT gradient (the reference's learn() restated in float64 on the same rows): at B = 512 the
        # reference's own fp32 conv1 gradient -- a 204 800-term reduction on the CPU -- is 2.5e-5 of the largest entry away from it.
        # Accept |ours - exact| <= max(1e-5, 2 x |reference - exact|) per tensor, and report both distances
        worst = {}
        for k, v in grads.items():
            g64, g32, scale = z[f"grad64_thin/{k}"], z[f"grad_thin/{k}"], float(z[f"grad_absmax/{k}"]) + 1e-30
            e_ours, e_ref = float(np.abs(synth.thin(v) - g64).max()) / scale, float(np.abs(g32 - g64).max()) / scale
            worst[k] = float(np.abs(synth.thin(v) - g32).max()) / scale
            vs64[k] = {"ours_vs_exact": e_ours, "reference_vs_exact": e_ref}
            margins.leq(e_ours, max(1e-5, 2.0 * e_ref), f"d(loss)/d{k} vs the exact gradient, of the largest entry (reference: {e_ref:.3e})")
        _report(f"{fixture}_{'graph' if use_graph else 'eager'}_grad_vs_float64", vs64)
    else:
        worst = _thin_cmp(grads, z, "grad_thin/", scale_of=lambda k: z[f"grad_absmax/{k}"], tol=1e-5, what="d(loss)/d")
    norm = float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    np.testing.assert_allclose(norm, float(z["grad_norm"]), rtol=1e-5)
    if coef < 1.0:
        np.testing.assert_allclose(norm * coef, float(z["grad_clip_norm"]), rtol=1e-5)
    # optimizer moments after the step (computed from the CLIPPED gradient): m = exp_avg | grad_avg, v = exp_avg_sq | square_avg
    names = ("exp_avg", "exp_avg_sq") if oc["name"] == "adam" else ("grad_avg", "square_avg")
    mom = {}
    for bucket, nm in ((agent._net.m, names[0]), (agent._net.v, names[1])):
        got = {k: v.cpu().numpy() for k, v in agent._net.export_state(bucket).items()}
        # where the fixture knows the reference's own fp32 distance from the exact gradient (vs64), a moment built from that
        # gradient (m ~ g, v ~ g^2: twice the relative error) cannot be pinned tighter than that
        mom[nm] = {}
        for k, v in got.items():
            if not vs64:
                mom[nm].update(_thin_cmp({k: v}, z, f"opt1_thin/{nm}/", tol=2e-5, what=nm))
                continue
            # the fixture holds the exact gradient: from zero state the first step's moments are c g (exp_avg / grad_avg) and c g^2
            # (exp_avg_sq / square_avg) of the clipped gradient, so the moment has an exact value too and the criterion is the
            # gradient's: |ours - exact| <= max(tol, 2 |reference - exact|), instead of a distance to the reference's own rounding
            first = nm in ("exp_avg", "grad_avg")
            c = (1.0 - defaults["betas"][0 if first else 1]) if oc["name"] == "adam" else (1.0 - defaults["alpha"])
            gc = coef * z[f"grad64_thin/{k}"].astype(np.float64)
            exact, ref = c * (gc if first else gc * gc), z[f"opt1_thin/{nm}/{k}"].astype(np.float64)
            scale = float(np.abs(ref).max()) + 1e-30
            e_ours, e_ref = float(np.abs(synth.thin(v) - exact).max()) / scale, float(np.abs(ref - exact).max()) / scale
            margins.leq(e_ours, max(2e-5, 2.0 * e_ref), f"{nm} {k} vs the moment of the exact gradient, of the largest entry (reference: {e_ref:.3e})")
            mom[nm][k] = e_ours
    if "grad_clip_norm" in z.files:
        cn, clip = float(z["grad_clip_norm"]), h("clip_grad_norm")
        assert (cn < clip * 1.0001) and (float(z["grad_norm"]) <= clip or abs(cn - clip) < 1e-4 * clip)
    _report(f"{fixture}_{'graph' if use_graph else 'eager'}", {"result_err_over_1_plus_abs_ref": errs, "grad_err_rel_to_absmax": worst,
                                                               "moment_err_rel_to_absmax": mom, "grad_norm": norm, "grad_norm_ref": float(z["grad_norm"])})
    # updated weights: both optimizers normalise the step (Adam: lr * sign-ish; centered RMSprop's first step: lr * g / (|g| sqrt(.0099) + eps)
    # ~ 10.05 lr), so a weight with a ~0 gradient may land a step apart: 99.5 % within 5 % of a step, none beyond the possible travel
    lr = float(h("optim_lr"))
    # ... and the step itself as ARITHMETIC (the statistical criterion below would not notice an eps or a bias correction
    # that is a few per cent off): float64 torch.optim from the fixture's w0 and zero state, fed our (clipped) gradient, must land on our weights
    import fp64_truth as T64

    clipped = {k: v for k, v in agent._net.export_state(agent._net.grads).items()}
    T64.check_first_step_from_our_gradient(w0, clipped, agent.network.state_dict(), lambda ps: Optimizer(**oc, params=ps), lr, f"{fixture} first step")
    travel = lr * (1.0 if oc["name"] == "adam" else 10.06)
    tot = bad = 0
    for k, v in agent.network.state_dict().items():
        d = np.abs(synth.thin(npy(v)) - z[f"sd1_thin/{k}"])
        tot += d.size
        bad += int((d > 0.05 * travel).sum())
        margins.leq(float(d.max()), 2.1 * travel, f"worst weight difference {k}")
    margins.leq(bad / tot, 0.005, "fraction of weights off by > 5 % of a step")


@pytest.mark.parametrize("use_graph", [True])
def test_rainbow_learn_at_atari_shapes_graph_replay(use_graph):
    """The hipGraph replay of Rainbow.learn() at config.rainbow.atari shapes against the reference's fixture (the bench replays
    this graph; the eager launch sequence is test_rainbow_learn_at_atari_shapes).  The reference's Gaussian draws are written
    into the graph's static noise buffer before every learn() (agent._noise = "static")."""
    from jorldy_amd.core.agent import Agent

    z = load("rainbow_cnn_atari")
    h = lambda k: z[f"hyper/{k}"].item()
    H, A, K, B, n = int(h("H")), int(h("A")), int(h("num_support")), int(h("B")), int(h("n_step"))
    S = tuple(int(v) for v in z["hyper/S"])
    agent = Agent("rainbow", state_size=S, action_size=A, hidden_size=H, head="cnn", optim_config={"name": "adam", "lr": h("lr")},
                  gamma=h("gamma"), buffer_size=64, batch_size=B, start_train_step=0, target_update_period=10000, run_step=100000,
                  n_step=n, alpha=h("alpha"), beta=h("beta"), learn_period=1, uniform_sample_prob=h("uniform_sample_prob"),
                  v_min=h("v_min"), v_max=h("v_max"), num_support=K, device="cuda", backend="native", use_graph=True)
    shapes = {k: v.shape for k, v in agent.network.state_dict().items()}
    seed = int(z["recipe_seed"])
    w0 = {k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, seed).items()}
    wt = {k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, seed + 1).items()}
    rng = np.random.RandomState(int(z["fill_seed"]))
    raw = [synth.raw_transition(rng, S, A) for _ in range(int(z["fill"]))]
    n_rows = len(raw) - n + 1
    cols = {"state": np.concatenate([raw[i]["state"] for i in range(n_rows)], 0),
            "next_state": np.concatenate([raw[i + n - 1]["next_state"] for i in range(n_rows)], 0)}
    for k in ("action", "reward", "done"):
        cols[k] = z[f"buf_{k}"]
    agent.memory.first_store = False
    agent.memory.store_soa(cols)
    torch.manual_seed(int(h("torch_seed")))
    noise = []
    for _ in range(3):
        d = {}
        for tag, (i, o) in (("a1", (H, H)), ("v1", (H, H)), ("a2", (H, K * A)), ("v2", (H, K))):
            d[tag] = (torch.randn(i).cuda(), torch.randn(o).cuda())
        noise.append(d)
    agent._noise = "static"
    agent._static, agent._graph = agent._alloc_static(), None
    for rep in range(3):
        agent.network.load_state_dict(w0)
        agent.target_network.load_state_dict(wt)
        agent._net.m.zero_()
        agent._net.v.zero_()
        agent._adam_steps = 0
        agent._net.set_hyper(h("lr"), 0.9, 0.999, 1e-8, 0)
        agent.memory._tree.load(z["tree0"], float(np.asarray(z["maxp0"]).reshape(-1)[0]), int(z["tree_index0"]), n_rows)
        for i in range(3):
            agent.network.pack_noise(noise[i], agent._static["noise"][i])
        np.random.seed(int(h("np_seed")))
        result = agent.learn()
    assert agent._graph is not None, "learn() was not captured"
    for k in ("loss", "max_Q", "max_logit", "min_logit"):
        np.testing.assert_allclose(result[k], z[f"result/{k}"], rtol=1e-5, atol=1e-5, err_msg=k)
    st = agent._static
    assert np.array_equal(npy(st["idx"]), z["learn/indices"].astype(np.int64))
    assert np.array_equal(npy(st["w"]), z["learn/weights"].astype(np.float32).reshape(-1))
    np.testing.assert_allclose(npy(st["logits"][0]), z["learn/logit"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(agent.memory.sum_tree, z["tree1"], rtol=2e-5, atol=1e-6)
    grads = {k: v.cpu().numpy() for k, v in agent._net.export_state(agent._net.grads).items()}
    _thin_cmp(grads, z, "grad_thin/", scale_of=lambda k: z[f"grad_absmax/{k}"], tol=1e-5, what="d(loss)/d")
