"""End-to-end parity of the drop-in agents on the GPU against the reference's own learn() runs
(fixtures from oracle/gen_golden.py): same inputs, same initial weights, same numpy RNG seed ->
same sampled indices (bit-exact), losses within 1e-5, updated weights within Adam-step noise."""
import os

import numpy as np
import pytest
import torch

import margins
from tests.util import cu, f32, load, npy

pytestmark = pytest.mark.gpu


def _sd(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def _rows(z, prefix, keys, n):
    return [{k: z[f"{prefix}{k}"][i : i + 1] for k in keys} for i in range(n)]


def _cmp_sd(net, gold, lr, n_updates, atol=2e-5):
    """Adam's first steps move a weight by ~lr*sign(g): weights whose gradient is ~0 can legitimately
    land one step apart, so require (a) 99.5 % of all weights within `atol`, (b) none further than
    the total possible travel."""
    tot, bad, worst = 0, 0, 0.0
    for k, v in net.state_dict().items():
        d = np.abs(npy(v) - gold[k].numpy())
        tot += d.size
        bad += int((d > atol).sum())
        worst = max(worst, float(d.max()))
    margins.leq(bad / tot, 0.005, f"fraction of weights further than {atol} from the reference's")
    margins.leq(worst, 2.1 * lr * n_updates, "worst weight difference vs the possible travel")


PPO_CASES = ["ppo_disc_small", "ppo_disc_cartpole", "ppo_cont_small", "ppo_cont_hopper"]


def _drift(name, s, z, n_upd, later):
    """Per-update error of the four loss scalars, |ours - ref| / (1 + |ref|): north_star's 1e-5 on update 0,
    `later` on the following ones (two fp32 Adam trajectories separate); the measured values go to
    gpurun_out/parity_drift_<name>.json."""
    import json

    drift = []
    for i in range(n_upd):
        drift.append(max(abs(float(s[i, j]) - float(z[f"mb{i}/{k}"])) / (1.0 + abs(float(z[f"mb{i}/{k}"])))
                         for j, k in enumerate(("loss", "actor_loss", "critic_loss", "entropy_loss"))))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"parity_drift_{name}.json"), "w") as f:
        json.dump({"per_update_max_err_over_1_plus_abs_ref": drift}, f, indent=1)
    for i, e in enumerate(drift):
        margins.leq(e, 1e-5 if i == 0 else later, f"{name} update {i} loss scalars (all: {['%.1e' % v for v in drift]})")


@pytest.mark.parametrize("name", PPO_CASES)
@pytest.mark.parametrize("soa", [False, True])
def test_ppo_learn_matches_reference(name, soa):
    from jorldy_amd.core.agent import Agent

    z = load(name)
    S, A, H, W, T, B, E, cont = [int(x) for x in z["cfg"]]
    gamma, lam, eps, vf, ent, clip, lr = z["hyper"]
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=H, network="continuous_policy_value" if cont else "discrete_policy_value",
                  optim_config={"name": "adam", "lr": lr}, batch_size=B, n_step=T, n_epoch=E, _lambda=lam, epsilon_clip=eps, vf_coef=vf,
                  ent_coef=ent, clip_grad_norm=clip, gamma=gamma, run_step=100000, num_workers=W, device="cuda")
    agent.network.load_state_dict(_sd(z, "sd0/"))
    agent.memory.first_store = False
    keys = ["state", "next_state", "reward", "done", "action"]
    M = W * T
    np.random.seed(int(z["np_seed"]))
    if soa:
        result = agent.process({k: z[f"in_{k}"] for k in keys}, T)
    else:
        result = agent.process(_rows(z, "in_", keys, M), T)
    assert agent.memory.size == 0  # cleared after sample (test_rollout_buffer.py:38)
    # per-update losses (the reference's .item() values) and the reported means
    n_upd = int(z["n_minibatch"])
    s = npy(agent._stats[:n_upd])
    _drift(f"{name}_torch_{'soa' if soa else 'rows'}", s, z, n_upd, later=1e-5)
    for k in ("actor_loss", "critic_loss", "entropy_loss", "mean_ret"):
        np.testing.assert_allclose(result[k], z[f"result/{k}"], rtol=1e-5, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(result["max_ratio"], z["result/max_ratio"], rtol=1e-3)
    np.testing.assert_allclose(agent.optimizer.param_groups[0]["lr"], z["lr_after"], rtol=1e-12)
    _cmp_sd(agent.network, _sd(z, "sd1/"), lr, n_upd)


def _fill_from_fixture(agent, z, per):
    keys = [k[4:] for k in z.files if k.startswith("buf_")]
    cols = {k: z[f"buf_{k}"] for k in keys}
    n = cols["state"].shape[0]
    agent.memory.first_store = False
    if per:
        agent.memory.store_soa(cols)
        agent.memory._tree.load(z["tree0"], float(np.asarray(z["maxp0"]).reshape(-1)[0]), int(z["tree_index0"]), n)
    else:
        agent.memory.store_soa(cols)
    return n


def _h(z, k):
    return z[f"hyper/{k}"].item()


@pytest.mark.parametrize("name", ["dqn", "double", "multistep", "per", "ape_x"])
def test_td_agents_learn_matches_reference(name):
    """The q-network / dueling encoder, its backward and the optimizer on libjorldy_hip (jh_rbnet_*) around the HIP loss / PER
    kernels, against the reference's own learn() (fixture)."""
    from jorldy_amd.core.agent import Agent

    z = load(name)
    extra = {}
    for k in ("n_step", "alpha", "beta", "learn_period", "uniform_sample_prob", "num_workers", "clip_grad_norm"):
        if f"hyper/{k}" in z.files:
            extra[k] = _h(z, k)
    agent = Agent(name, state_size=int(_h(z, "S")), action_size=int(_h(z, "A")), hidden_size=int(_h(z, "H")), optim_config={"name": "adam", "lr": _h(z, "lr")},
                  gamma=_h(z, "gamma"), buffer_size=256, batch_size=int(_h(z, "B")), start_train_step=0, target_update_period=10000, run_step=100000, device="cuda", **extra)
    assert agent.backend == "native"
    agent.network.load_state_dict(_sd(z, "sd0/"))
    agent.target_network.load_state_dict(_sd(z, "sdt/"))
    per = name in ("per", "ape_x")
    _fill_from_fixture(agent, z, per)
    np.random.seed(int(_h(z, "np_seed")))
    result = agent.learn()
    np.testing.assert_allclose(result["loss"], z["result/loss"], rtol=1e-5)
    np.testing.assert_allclose(result["max_Q"], z["result/max_Q"], rtol=1e-5)
    if per:
        np.testing.assert_allclose(result["sampled_p"], z["result/sampled_p"], rtol=1e-12)
        assert result["mean_p"] == z["result/mean_p"].item()
        # tree after OUR fp32 priorities: equals the reference's up to the fp32 rounding of |td|^alpha
        np.testing.assert_allclose(agent.memory.sum_tree, z["tree1"], rtol=1e-5, atol=1e-6)
    _cmp_sd(agent.network, _sd(z, "sd1/"), _h(z, "lr"), 1)


def test_c51_agent_learn_matches_reference():
    from jorldy_amd.core.agent import Agent

    z = load("c51")
    agent = Agent("c51", state_size=int(_h(z, "S")), action_size=int(_h(z, "A")), hidden_size=int(_h(z, "H")), optim_config={"name": "adam", "lr": _h(z, "lr")},
                  gamma=_h(z, "gamma"), buffer_size=256, batch_size=int(_h(z, "B")), start_train_step=0, target_update_period=10000, run_step=100000,
                  v_min=_h(z, "v_min"), v_max=_h(z, "v_max"), num_support=int(_h(z, "num_support")), device="cuda")
    agent.network.load_state_dict(_sd(z, "sd0/"))
    agent.target_network.load_state_dict(_sd(z, "sdt/"))
    _fill_from_fixture(agent, z, False)
    np.random.seed(int(_h(z, "np_seed")))
    result = agent.learn()
    for k in ("loss", "max_Q", "max_logit", "min_logit"):
        np.testing.assert_allclose(result[k], z[f"result/{k}"], rtol=1e-5, err_msg=k)
    _cmp_sd(agent.network, _sd(z, "sd1/"), _h(z, "lr"), 1)


@pytest.mark.parametrize("fixture", ["rainbow", "rainbow_cnn"])
def test_rainbow_agent_learn_matches_reference(fixture):
    """The reference draws NoisyNet noise with the CPU generator inside forward (utils.py:58-60); the
    same draws are regenerated here (same seed, same order) and injected so the whole update is
    comparable.  rainbow_cnn = Nature-CNN head on uint8 frames."""
    from jorldy_amd.core.agent import Agent

    z = load(fixture)
    H, A, K = int(_h(z, "H")), int(_h(z, "A")), int(_h(z, "num_support"))
    S = z["hyper/S"]
    cnn = S.ndim > 0
    agent = Agent("rainbow", state_size=tuple(int(v) for v in S) if cnn else int(S), action_size=A, hidden_size=H, head="cnn" if cnn else "mlp",
                  optim_config={"name": "adam", "lr": _h(z, "lr")},
                  gamma=_h(z, "gamma"), buffer_size=64 if cnn else 256, batch_size=int(_h(z, "B")), start_train_step=0, target_update_period=10000, run_step=100000,
                  n_step=int(_h(z, "n_step")), alpha=_h(z, "alpha"), beta=_h(z, "beta"), learn_period=1, uniform_sample_prob=_h(z, "uniform_sample_prob"),
                  v_min=_h(z, "v_min"), v_max=_h(z, "v_max"), num_support=K, device="cuda")
    assert agent.backend == "native"
    agent.network.load_state_dict(_sd(z, "sd0/"))
    agent.target_network.load_state_dict(_sd(z, "sdt/"))
    n = _fill_from_fixture(agent, z, True)
    if cnn:
        assert agent.memory._store.column("state").dtype == torch.uint8  # frames stay uint8 in HBM
    torch.manual_seed(int(_h(z, "torch_seed")))
    noise = []
    for _ in range(3):  # network(state), network(next_state), target_network(next_state)
        d = {}
        for tag, (i, o) in (("a1", (H, H)), ("v1", (H, H)), ("a2", (H, K * A)), ("v2", (H, K))):
            d[tag] = (torch.randn(i).cuda(), torch.randn(o).cuda())
        noise.append(d)
    agent._noise = noise
    np.random.seed(int(_h(z, "np_seed")))
    result = agent.learn()
    for k in ("loss", "max_Q", "max_logit", "min_logit"):
        np.testing.assert_allclose(result[k], z[f"result/{k}"], rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(result["sampled_p"], z["result/sampled_p"], rtol=1e-12)
    np.testing.assert_allclose(agent.memory.sum_tree, z["tree1"], rtol=1e-4 if cnn else 2e-5, atol=1e-6)
    grads = agent._net.export_state(agent._net.grads)  # gradients of every parameter against the reference's autograd
    for k, v in grads.items():
        ref = z[f"grad/{k}"]
        margins.leq(float(np.abs(v.cpu().numpy() - ref).max()), 5e-5 * (float(np.abs(ref).max()) + 1e-12) + 1e-9, f"grad {k}")
    _cmp_sd(agent.network, _sd(z, "sd1/"), _h(z, "lr"), 1)


def test_rainbow_native_graph_replay_equals_eager():
    """Native learn() as one hipGraph == the same launches issued eagerly (same device RNG stream)."""
    from jorldy_amd.core.agent import Agent

    z = load("rainbow_cnn")
    H, A, K = int(_h(z, "H")), int(_h(z, "A")), int(_h(z, "num_support"))
    res = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        agent = Agent("rainbow", state_size=(4, 44, 52), action_size=A, hidden_size=H, head="cnn", optim_config={"name": "adam", "lr": 1e-3},
                      buffer_size=64, batch_size=8, start_train_step=0, target_update_period=3, run_step=1000, n_step=3, alpha=0.5, beta=0.4,
                      learn_period=1, uniform_sample_prob=0.05, v_min=-1, v_max=10, num_support=K, device="cuda", use_graph=use_graph)
        assert agent.backend == "native"
        agent.network.load_state_dict(_sd(z, "sd0/"))
        agent.target_network.load_state_dict(_sd(z, "sdt/"))
        _fill_from_fixture(agent, z, True)
        np.random.seed(7)
        torch.manual_seed(11)
        out = []
        for it in range(6):
            r = agent.learn()
            agent.learning_rate_decay(10 * (it + 1))
            if it == 3:
                agent.update_target()
            out.append(r["loss"])
        if use_graph:
            assert agent._graph is not None, "learn() was not captured"
        res.append((out, torch.cat([p.reshape(-1) for p in agent.network.parameters()]).clone(), agent.memory.sum_tree))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-6)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=0, atol=0)
    np.testing.assert_array_equal(res[0][2], res[1][2])


def test_rainbow_native_checkpoint_interchanges_with_torch_modules(tmp_path):
    """ckpt written by the agent = the reference's format ({"network": state_dict, "optimizer": torch.optim state_dict},
    rainbow.py / dqn.py:184-199): loads into the reference-shaped torch module + torch.optim.Adam (tests/mirror) and back
    with parameters and Adam moments intact."""
    from jorldy_amd.core.agent import Agent
    from mirror.networks import Network

    z = load("rainbow")
    H, A, K = int(_h(z, "H")), int(_h(z, "A")), int(_h(z, "num_support"))
    mk = lambda: Agent("rainbow", state_size=int(z["hyper/S"]), action_size=A, hidden_size=H, optim_config={"name": "adam", "lr": 1e-3}, buffer_size=256,
                       batch_size=int(_h(z, "B")), start_train_step=0, run_step=1000, n_step=3, num_support=K, device="cuda", use_graph=False)
    a = mk()
    a.network.load_state_dict(_sd(z, "sd0/"))
    _fill_from_fixture(a, z, True)
    np.random.seed(1)
    for _ in range(3):
        a.learn()
    a.save(str(tmp_path))
    ck = torch.load(str(tmp_path / "ckpt"), map_location="cpu", weights_only=False)
    mod = Network("rainbow", int(z["hyper/S"]), A, K, "factorized", D_hidden=H, head="mlp")
    mod.load_state_dict(ck["network"])  # strict: same keys, same shapes
    opt = torch.optim.Adam(mod.parameters(), lr=1e-3)
    opt.load_state_dict(ck["optimizer"])
    for (k, v1), (_, v2) in zip(a.network.state_dict().items(), mod.state_dict().items()):
        assert torch.equal(v1.cpu(), v2), k
    st = opt.state_dict()["state"]
    assert len(st) == len(list(mod.parameters())) and all(int(float(s["step"])) == 3 for s in st.values())
    m_nat = a._net.export_state(a._net.m)
    for (k, v), s_ in zip(m_nat.items(), st.values()):
        assert torch.equal(v.cpu(), s_["exp_avg"]), k
    (tmp_path / "b").mkdir()
    torch.save({"network": mod.state_dict(), "optimizer": opt.state_dict()}, str(tmp_path / "b" / "ckpt"))  # what the reference's save() writes
    c = mk()
    c.load(str(tmp_path / "b"))
    assert c._adam_steps == 3
    assert torch.equal(c._net.params, a._net.params) and torch.equal(c._net.m, a._net.m) and torch.equal(c._net.v, a._net.v)


# ---- the reference's own smoke/bookkeeping assertions (jorldy/test/core/agent/*.py) ------------------
class _MockEnv:
    """test/conftest.py:9-45."""

    def __init__(self, state_size, action_size, episode_len=5):
        self.state_size, self.action_size, self.episode_len, self.t = state_size, action_size, episode_len, 0

    def reset(self):
        return np.random.random((1, self.state_size))

    def step(self, action):
        self.t += 1
        done = np.array([[self.t == self.episode_len]])
        if done:
            self.t = 0
        return np.random.random((1, self.state_size)), np.random.random((1, 1)), done


def _interact(env, agent, run_step, start=0):
    """test/core/agent/utils.py:5-24."""
    state = env.reset()
    for step in range(start + 1, start + run_step + 1):
        action_dict = agent.act(state, training=True)
        assert action_dict["action"].shape == (1, 1) or action_dict["action"].shape[0] == 1
        next_state, reward, done = env.step(action_dict["action"])
        transition = {"state": state, "next_state": next_state, "reward": reward, "done": done}
        transition.update(action_dict)
        transition = agent.interact_callback(transition)
        if transition:
            if "priority" in transition:
                transition["priority"] = np.asarray(transition["priority"]).reshape(1, 1)
            agent.process([transition], step)
        state = next_state if not done else env.reset()


@pytest.mark.parametrize("name,extra,check", [
    ("dqn", dict(), lambda a, rs: a.memory.size == rs and a.time_t == rs),
    ("double", dict(), lambda a, rs: a.memory.size == rs),
    ("multistep", dict(n_step=3), lambda a, rs: a.memory.size == rs - 3 + 1),
    ("per", dict(learn_period=2), lambda a, rs: a.memory.size == rs and a.beta <= 1.0),
    ("ape_x", dict(n_step=3, num_workers=2, learn_period=2), lambda a, rs: a.memory.size == rs - 3),  # test_ape_x_agent.py:46-48
    ("c51", dict(num_support=11), lambda a, rs: a.memory.size == rs),
    ("rainbow", dict(n_step=3, num_support=11, learn_period=2), lambda a, rs: a.memory.size == rs - 3 + 1),
    ("ppo", dict(n_step=8, batch_size=4, n_epoch=2), lambda a, rs: a.memory.size == rs % 8),  # test_ppo_agent.py:34
])
def test_reference_bookkeeping_asserts(name, extra, check, tmp_path):
    from jorldy_amd.core.agent import Agent

    S, A, run_step = 6, 3, 20
    kw = dict(state_size=S, action_size=A, hidden_size=16, batch_size=4, start_train_step=5, buffer_size=64, run_step=run_step, device="cuda")
    kw.update(extra)
    agent = Agent(name, **kw)
    agent.memory.first_store = False
    _interact(_MockEnv(S, A), agent, run_step)
    assert check(agent, run_step)
    # check_save_load / check_sync_in_out (test/core/agent/utils.py:27-39)
    agent.save(str(tmp_path))
    agent.load(str(tmp_path))
    item = agent.sync_out()
    assert all(v.device.type == "cpu" for v in item["weights"].values())
    agent.sync_in(**item)


# ---- native backend (hand-written MLP fwd/bwd + clip + Adam) ---------------------------------------
def _unflat(flat, module):
    """flat bucket (state_dict order) -> {name: tensor shaped like the module's parameter}."""
    out, o = {}, 0
    for k, p in module.named_parameters():
        out[k] = flat[o : o + p.numel()].view_as(p)
        o += p.numel()
    return out


def _policy64(agent, name, S, A, H):
    """The agent's CURRENT policy as the reference's module in float64 on the CPU (tests/mirror): what `agent.network(x)` meant
    when the product still carried a torch forward."""
    from mirror.networks import Network

    m = Network(name, S, A, D_hidden=H).double()
    m.load_state_dict({k: v.detach().cpu().double() for k, v in agent.network.state_dict().items()})
    return m


def test_pponet_forward_backward_adam_vs_float64():
    """jh_pponet_* against the reference's module + torch.optim.Adam evaluated in float64 on the CPU (tests/fp64_truth.py), with
    torch-CPU fp32 beside it: heads, every gradient, the clip norm, Adam's moments and the stepped weights, three teacher-forced steps."""
    import fp64_truth as T
    from jorldy_amd import ops
    from mirror.networks import Network

    # (the last three: more than 8 head outputs -- 13 / 13 / 17 -- on the separate forward / backward calls; the tiled engine at 13 outputs: test_baseline_width_gpu, ppo_cont_halfcheetah)
    # the last two: a hidden row NARROWER than the packed head gradient (Humanoid's 35 outputs = 36 columns over hidden 32, 19 outputs = 20 columns
    # over hidden 16): the columns beyond the row were never packed and their head gradients silently zero
    for cont, S, H, A, B in ((False, 4, 512, 2, 256), (True, 11, 64, 3, 100), (False, 7, 32, 5, 8), (True, 17, 64, 6, 100), (False, 5, 32, 12, 40), (True, 27, 128, 8, 200),
                             (True, 45, 32, 17, 64), (True, 9, 16, 9, 33)):
        torch.manual_seed(0)
        ref64 = Network("continuous_policy_value" if cont else "discrete_policy_value", S, A, D_hidden=H).double()
        with torch.no_grad():
            for p in ref64.parameters():
                p.add_(0.05 * torch.randn_like(p))
        T.round_to_fp32_(ref64)
        ref32 = T.as32(ref64)
        net = ops.PPONet(S, H, A, cont, 1024, "cuda:0")
        assert sum(p.numel() for p in ref64.parameters()) == net.n_params
        lr = 1e-3
        truth = T.OptimTruth(ref64, ref32, lambda ps: torch.optim.Adam(ps, lr=lr), lr, ("exp_avg", "exp_avg_sq"))
        g = torch.Generator().manual_seed(1)
        x_all = torch.randn(300, S, generator=g)
        x_dev = x_all.cuda()
        flat = lambda d: torch.cat([d[k].reshape(-1) for k, _ in ref64.named_parameters()]).cuda()
        for it in range(3):
            params, m, v = truth.teacher_force()
            net.params.copy_(flat(params))
            if m is None:
                net.m.zero_()
                net.v.zero_()
            else:
                net.m.copy_(flat(m))
                net.v.copy_(flat(v))
            net.set_hyper(lr, 0.9, 0.999, 1e-8, step=float(it))
            idx = torch.randperm(300, generator=g)[:B]
            outs = net.forward(x_dev, idx=idx.cuda())
            r64, r32 = ref64.raw(x_all[idx].double()), ref32.raw(x_all[idx])
            for nm, a, b, c in zip(("mu", "log_std", "value") if cont else ("logits", "value"), outs, r64, r32):
                T.vs_exact(a, b, c, 1e-5, f"cont={cont} step {it} {nm}")
            gs = [torch.randn(o.shape, generator=g) / B for o in r64]
            truth.opt64.zero_grad(set_to_none=True)
            truth.opt32.zero_grad(set_to_none=True)
            torch.autograd.backward(list(r64), [t.double() for t in gs])
            torch.autograd.backward(list(r32), gs)
            gd = [t.cuda() for t in gs]
            if cont:
                net.backward(x_dev, idx.cuda(), gd[0], gd[1], gd[2])
            else:
                net.backward(x_dev, idx.cuda(), gd[0], None, gd[1])
            ours_g = _unflat(net.grads, ref64)
            p32 = dict(ref32.named_parameters())
            for k, p in ref64.named_parameters():
                T.vs_exact(ours_g[k], p.grad, p32[k].grad, 1e-5, f"cont={cont} step {it} grad {k}")
            ours_raw = {k: v.clone() for k, v in ours_g.items()}  # adam_step clips the bucket in place
            norm64 = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ours_raw.values())))  # float64 norm of OUR gradient
            norm = torch.zeros(1, device="cuda")
            net.adam_step(0.5, norm)
            margins.close(float(norm[0]), norm64, rtol=1e-5, what=f"cont={cont} step {it} grad norm")
            truth.step(0.5, ours_raw, _unflat(net.params, ref64), _unflat(net.m, ref64), _unflat(net.v, ref64), tag=f"cont={cont} adam step {it}")


@pytest.mark.parametrize("cont,S,H,A,B", [(False, 4, 512, 2, 256), (True, 11, 64, 3, 100), (False, 7, 32, 5, 8), (False, 8, 64, 3, 250), (True, 3, 96, 2, 1000)])
def test_ppo_update_five_launches_equal_separate_calls(cont, S, H, A, B):
    """jh_pponet_ppo_update (forward into partial heads, loss on the partials, one backward grid with the dW1 partials,
    combine + norm, Adam) against jh_pponet_forward -> jh_ppo_loss_* -> jh_pponet_backward -> jh_pponet_adam_step on the
    same inputs: same statistics, same gradient bucket (raw with do_adam = 0, clipped with 1), same parameters.
    Shapes cover layer 1 in registers (S = 4 / 8 vector loads, S = 7 / 3 scalar), the l1 kernel (S = 11), ragged B."""
    from jorldy_amd import ops

    torch.manual_seed(S * 1000 + B)
    M = 1500
    x = torch.randn(M, S, device="cuda")
    action = torch.tanh(torch.randn(M, A, device="cuda")) if cont else torch.randint(0, A, (M, 1), device="cuda").float()
    adv, ret, vold = torch.randn(M, 1, device="cuda"), torch.randn(M, 1, device="cuda"), torch.randn(M, 1, device="cuda")
    logp_old = -torch.rand(M, A if cont else 1, device="cuda") - 0.3
    idx = torch.randperm(M, device="cuda")[:B].contiguous()
    nets = [ops.PPONet(S, H, A, cont, 2048, "cuda:0") for _ in range(3)]
    p0 = torch.randn(nets[0].n_params, device="cuda") * (0.7 / np.sqrt(H))
    for net in nets:
        net.params.copy_(p0)
        net.set_hyper(1e-3, 0.9, 0.999, 1e-8, step=0.0)
    eps, vf, ent, clip = 0.1, 1.0, 0.01, 0.5
    # reference sequence on net 0
    n0 = nets[0]
    st0 = torch.zeros(8, device="cuda")
    if cont:
        mu, ls, vp = n0.forward(x, idx=idx)
        g_mu, g_ls, g_v, _ = ops.ppo_loss_continuous(mu, ls, vp, idx, action, adv, ret, vold, logp_old, eps, vf, ent, stats=st0)
        n0.backward(x, idx, g_mu, g_ls, g_v)
    else:
        z, vp = n0.forward(x, idx=idx)
        g_z, g_v, _ = ops.ppo_loss_discrete(z, vp, idx, action, adv, ret, vold, logp_old, eps, vf, ent, stats=st0)
        n0.backward(x, idx, g_z, None, g_v)
    raw = n0.grads.clone()
    n0.adam_step(clip)
    # five launches without Adam (data-parallel form), then the separate optimizer step
    n1, n2 = nets[1], nets[2]
    st1, st2 = torch.zeros(8, device="cuda"), torch.zeros(8, device="cuda")
    n1.ppo_update(x, idx, action, adv, ret, vold, logp_old, eps, vf, ent, clip, st1, do_adam=False)
    scale = float(raw.abs().max())
    margins.leq(float((n1.grads - raw).abs().max()), 1e-5 * scale, "raw gradient bucket")
    torch.testing.assert_close(st1, st0, rtol=1e-5, atol=1e-6)
    n1.adam_step(clip)
    n2.ppo_update(x, idx, action, adv, ret, vold, logp_old, eps, vf, ent, clip, st2, do_adam=True)
    torch.testing.assert_close(st2, st1, rtol=0, atol=0)  # same kernels, same bits
    import fp64_truth as T

    for tag, n in (("separate calls", n0), ("update + adam_step", n1), ("update with Adam", n2)):
        if n is not n0:
            margins.leq(float((n.grads - n0.grads).abs().max()), 1e-5 * float(n0.grads.abs().max()), "clipped gradient bucket")
        # the stepped weights as ARITHMETIC (no more "worst difference 2.1e-3"): a weight after Adam's first step is
        # ill conditioned in its gradient (it moves by lr sign(g) whatever |g|), so two paths whose gradients agree to 1e-5 may land a
        # step apart on weights with g ~ 0 -- but EACH path's weights must be float64 Adam applied to ITS OWN clipped gradient, per element
        T.check_first_step_from_our_gradient({"bucket": p0}, {"bucket": n.grads}, {"bucket": n.params},
                                             lambda ps: torch.optim.Adam(ps, lr=1e-3, betas=(0.9, 0.999), eps=1e-8), 1e-3, f"five launches ({tag}) cont={cont} H={H} B={B}")
    torch.testing.assert_close(n2.params, n1.params, rtol=0, atol=1e-7)


@pytest.mark.parametrize("name", PPO_CASES)
@pytest.mark.parametrize("graph", [False, True])
def test_ppo_native_backend_matches_reference(name, graph):
    """Whole learn() on hand-written kernels (and replayed from a hipGraph) vs the reference run."""
    from jorldy_amd.core.agent import Agent

    z = load(name)
    S, A, H, W, T, B, E, cont = [int(x) for x in z["cfg"]]
    gamma, lam, eps, vf, ent, clip, lr = z["hyper"]
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=H, network="continuous_policy_value" if cont else "discrete_policy_value",
                  optim_config={"name": "adam", "lr": lr}, batch_size=B, n_step=T, n_epoch=E, _lambda=lam, epsilon_clip=eps, vf_coef=vf,
                  ent_coef=ent, clip_grad_norm=clip, gamma=gamma, run_step=100000, num_workers=W, device="cuda", backend="native", use_graph=graph)
    assert agent.backend == "native"
    keys = ["state", "next_state", "reward", "done", "action"]
    cols = {k: z[f"in_{k}"] for k in keys}
    agent.memory.first_store = False
    n_rep = 3 if graph else 1  # graph: eager warm-up, capture+replay, replay -- each from the same start
    for rep in range(n_rep):
        agent.network.load_state_dict(_sd(z, "sd0/"))
        agent._net.m.zero_()
        agent._net.v.zero_()
        agent._adam_steps = 0
        agent._net.set_hyper(lr, 0.9, 0.999, 1e-8, step=0.0)
        agent.time_t, agent.learn_stamp = 0, 0
        np.random.seed(int(z["np_seed"]))
        result = agent.process(cols, T)
        if graph and rep >= 1:
            assert agent._graph is not None
        n_upd = int(z["n_minibatch"])
        s = npy(agent._stats[:n_upd])
        _drift(f"{name}_native_{'graph' if graph else 'eager'}", s, z, n_upd, later=1e-5)
        for k in ("actor_loss", "critic_loss", "entropy_loss", "mean_ret"):
            np.testing.assert_allclose(result[k], z[f"result/{k}"], rtol=1e-5, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(agent.optimizer.param_groups[0]["lr"], z["lr_after"], rtol=1e-12)
        _cmp_sd(agent.network, _sd(z, "sd1/"), lr, n_upd, atol=3e-5)


def test_ppo_native_checkpoint_roundtrip(tmp_path):
    """ckpt keeps the reference's {"network", "optimizer"} format and restores the native Adam state."""
    from jorldy_amd.core.agent import Agent

    z = load("ppo_disc_small")
    S, A, H, W, T, B, E, cont = [int(x) for x in z["cfg"]]
    mk = lambda: Agent("ppo", state_size=S, action_size=A, hidden_size=H, optim_config={"name": "adam", "lr": 1e-3}, batch_size=B, n_step=T, n_epoch=E, run_step=1000, device="cuda", backend="native", use_graph=False)
    cols = {k: z[f"in_{k}"] for k in ["state", "next_state", "reward", "done", "action"]}
    a1 = mk()
    a1.memory.first_store = False
    np.random.seed(1)
    a1.process(cols, T)
    a1.save(str(tmp_path))
    ck = torch.load(str(tmp_path / "ckpt"), weights_only=False)
    assert set(ck) == {"network", "optimizer"} and set(ck["network"]) == {"head.l.weight", "head.l.bias", "l.weight", "l.bias", "pi.weight", "pi.bias", "v.weight", "v.bias"}
    assert int(ck["optimizer"]["state"][0]["step"]) == a1._adam_steps
    a2 = mk()
    a2.memory.first_store = False
    a2.load(str(tmp_path))
    a2.time_t = a1.time_t
    for ag in (a1, a2):
        np.random.seed(2)
        ag.process(cols, 2 * T)
    torch.testing.assert_close(a1._net.params, a2._net.params, rtol=0, atol=0)
    torch.testing.assert_close(a1._net.m, a2._net.m, rtol=0, atol=0)


def test_native_act_discrete_distribution():
    """On-device multinomial acting: empirical action frequencies match softmax(logits)."""
    from jorldy_amd.core.agent import Agent

    agent = Agent("ppo", state_size=4, action_size=3, hidden_size=32, device="cuda", backend="native")
    with torch.no_grad():
        agent.network.pi.bias.copy_(torch.tensor([0.0, 1.0, -0.5], device="cuda"))
    obs = np.zeros((64, 4), np.float32)
    counts = np.zeros(3)
    for _ in range(200):
        a = agent.act(obs, training=True)["action"]
        assert a.shape == (64, 1) and a.dtype == np.int64
        counts += np.bincount(a.reshape(-1), minlength=3)
    with torch.no_grad():
        pi, _ = _policy64(agent, "discrete_policy_value", 4, 3, 32)(torch.zeros(1, 4, dtype=torch.float64))
    p = npy(pi)[0]
    np.testing.assert_allclose(counts / counts.sum(), p, atol=0.015)
    g = agent.act(obs, training=False)["action"]
    assert np.all(g == int(np.argmax(p)))


def test_native_collector_matches_python_collector_layout():
    """jh_collector_run: worker-major transitions in the rollout store == the per-step Python loop's,
    for the same env seed (actions differ run to run only through the sampling RNG, so compare with a
    greedy-free invariant: replay the stored actions through the oracle env)."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector
    from oracle.jorldy_oracle import CartPoleOracle

    W, T = 5, 40
    agent = Agent("ppo", state_size=4, action_size=2, hidden_size=32, n_step=T, batch_size=50, device="cuda", backend="native")
    agent.memory.first_store = False
    env = ops.CartPoleVec(W, seed=9)
    col = NativeCollector(env, agent, W)
    col.run(T)
    torch.cuda.synchronize()
    st = agent.memory._store
    assert st.size == W * T
    cols = {k: npy(st.column(k)[: W * T]) for k in ("state", "action", "reward", "next_state", "done")}
    orc = CartPoleOracle(W, seed=9)
    a = cols["action"].reshape(W, T)
    for t in range(T):
        obs = orc.obs()
        nxt, rew, done = orc.step(a[:, t])
        rows = np.arange(W) * T + t  # worker-major
        np.testing.assert_array_equal(cols["state"][rows], obs)
        np.testing.assert_array_equal(cols["next_state"][rows], nxt)
        np.testing.assert_array_equal(cols["reward"][rows, 0], rew)
        np.testing.assert_array_equal(cols["done"][rows, 0].astype(bool), done)
    assert set(np.unique(a)) <= {0, 1}
    # and the learner consumes it
    res = agent.process(None, T)
    assert set(res) == {"actor_loss", "critic_loss", "entropy_loss", "max_ratio", "min_prob", "mean_ret"} and agent.memory.size == 0


@pytest.mark.parametrize("T,H", [(16, 64), (17, 64), (128, 512)])
def test_collector_lookahead_two_timesteps_per_exchange_is_bit_identical(T, H, monkeypatch):
    """jh_collect.hip run_loop_lookahead: for two-action envs the collector publishes every env's state AND both successor states
    in one exchange with the persistent acting kernel and takes TWO timesteps from it.  Same policy evaluations at the visited
    states, same sampling counters, same per-env RNG streams: stored transitions, the captured heads / values and the envs
    themselves must equal the one-timestep-per-exchange path (JH_COLLECT_LOOKAHEAD=1) BIT FOR BIT -- over several runs (the
    persistent kernel is relaunched per run), even and odd T (an odd run ends on a one-step exchange), with episode ends
    (random policy: ~22-step episodes, so resets happen inside looked-ahead successors), and through learn() updates in between."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    W = 8
    res = {}
    for la in (1, 2):
        monkeypatch.setenv("JH_COLLECT_LOOKAHEAD", str(la))
        torch.manual_seed(5)
        np.random.seed(5)
        agent = Agent("ppo", state_size=4, action_size=2, hidden_size=H, n_step=T, batch_size=64, n_epoch=1, device="cuda", seed=3, lr_decay=False)
        agent.memory.first_store = False
        env = ops.CartPoleVec(W, seed=4)
        col = NativeCollector(env, agent, W)
        out = []
        for it in range(3):
            col.run(T)
            torch.cuda.synchronize()
            st, M = agent._static, W * T
            store = agent.memory._store
            rec = {k: npy(store.column(k)[:M]).copy() for k in ("state", "action", "reward", "next_state", "done")}
            rec.update(h0=npy(st["h0"]).copy(), value=npy(st["value"]).copy(), next_value=npy(st["next_value"]).copy(), obs=env.obs().copy())
            out.append(rec)
            r = agent.process(None, T * (it + 1))  # a learn() between the runs: the next run acts with updated weights
            rec["loss"] = (r["actor_loss"], r["critic_loss"], r["entropy_loss"])
        assert any(o["done"].any() for o in out), "no episode ended: the reset path was not exercised"
        res[la] = out
        col.terminate()
    for a, b in zip(res[1], res[2]):
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


class _PyCartPole:
    """The oracle's CartPole (bit-identical to the library's jh_cartpole) as a PYTHON vector env with the VecCollector protocol, plus --
    when `forkable` -- fork / copy_row / row ranges: what a user's own env has to offer the C collector's function table."""

    state_size, action_size, action_type = 4, 2, "discrete"

    def __init__(self, W, seed=0, forkable=False):
        from oracle.jorldy_oracle import CartPoleOracle

        self.W, self.o = W, CartPoleOracle(W, seed=seed)
        if forkable:
            self.fork = lambda rows: _PyCartPole(rows, 0, True)
            self.copy_row = self._copy_row

    def _copy_row(self, di, src, si):
        self.o.s[di], self.o.t[di], self.o.rng[di] = src.o.s[si], src.o.t[si], src.o.rng[si]

    def obs(self, out, r0=0, r1=None):
        out[:] = self.o.s[r0 : self.W if r1 is None else r1].astype(np.float32)
        return out

    def step(self, action, nxt, rew, done, r0=0, r1=None):
        r1 = self.W if r1 is None else r1
        full = self.o
        if (r0, r1) != (0, self.W):  # a scratch env stepped on a row range: the oracle steps whole envs
            from oracle.jorldy_oracle import CartPoleOracle

            sub = CartPoleOracle.__new__(CartPoleOracle)
            sub.W, sub.s, sub.t, sub.rng = r1 - r0, full.s[r0:r1], full.t[r0:r1], full.rng[r0:r1]  # views: stepped in place
            full = sub
        n, r, d = full.step(np.asarray(action).reshape(-1))
        nxt[:], rew[:], done[:] = n, r, d
        return nxt, rew, done


@pytest.mark.parametrize("forkable", [False, True])
def test_c_collector_on_a_python_env_through_the_function_table(forkable):
    """jh_collector_create_env: the collector takes its env as a table of functions -- here Python callbacks around the
    oracle's CartPole.  Same dynamics and RNG streams as the library's own CartPole, same agent seed: stored transitions, captured heads
    and values must equal the built-in env's BIT FOR BIT, without the fork capability (one timestep per exchange) and with it (two
    timesteps per exchange: the lookahead is a capability of the table, not of the built-in type)."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    W, T, H = 8, 17, 64
    res = {}
    for kind in ("builtin", "python"):
        torch.manual_seed(5)
        np.random.seed(5)
        agent = Agent("ppo", state_size=4, action_size=2, hidden_size=H, n_step=T, batch_size=64, n_epoch=1, device="cuda", seed=3, lr_decay=False)
        agent.memory.first_store = False
        env = ops.CartPoleVec(W, seed=4) if kind == "builtin" else _PyCartPole(W, seed=4, forkable=forkable)
        col = NativeCollector(env, agent, W)
        out = []
        for it in range(2):
            col.run(T)
            torch.cuda.synchronize()
            st, M = agent._static, W * T
            store = agent.memory._store
            rec = {k: npy(store.column(k)[:M]).copy() for k in ("state", "action", "reward", "next_state", "done")}
            rec.update(h0=npy(st["h0"]).copy(), value=npy(st["value"]).copy(), next_value=npy(st["next_value"]).copy())
            out.append(rec)
            agent.process(None, T * (it + 1))
        res[kind] = out
        col.terminate()
    assert any(o["done"].any() for o in res["builtin"]), "no episode ended: the reset path was not exercised"
    for a, b in zip(res["builtin"], res["python"]):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("cont,W,H", [(True, 32, 512), (True, 8, 64), (False, 8, 128)])
def test_persistent_kernel_device_side_sum_of_the_partial_heads_is_bit_identical(cont, W, H, monkeypatch):
    """JH_PERSIST_REDUCE=1 (round 6): the column tiles' partial heads are fetched and added IN TILE ORDER by one workgroup of the acting
    kernel instead of by the host -- the same additions in the same order: stored transitions, captured heads / values bit-identical to the
    direct path, at config.ppo.mujoco's 32 workers x 7 outputs (two row tiles, three granules), at one tile per wave (hidden 64) and for
    a discrete policy's two-timestep exchanges (24 rows)."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    T, S, A = 12, 11 if cont else 4, 3 if cont else 2
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("JH_PERSIST_REDUCE", mode)
        torch.manual_seed(5)
        np.random.seed(5)
        agent = Agent("ppo", state_size=S, action_size=A, hidden_size=H, network="continuous_policy_value" if cont else "discrete_policy_value", n_step=T, batch_size=64,
                      n_epoch=1, device="cuda", seed=3, lr_decay=False, num_workers=W)
        agent.memory.first_store = False
        env = ops.ControlVec(W, S, A, seed=4) if cont else ops.CartPoleVec(W, seed=4)
        col = NativeCollector(env, agent, W)
        out = []
        for it in range(2):
            col.run(T)
            torch.cuda.synchronize()
            st, M = agent._static, W * T
            store = agent.memory._store
            rec = {k: npy(store.column(k)[:M]).copy() for k in ("state", "action", "reward", "next_state", "done")}
            rec.update(h0=npy(st["h0"]).copy(), value=npy(st["value"]).copy(), next_value=npy(st["next_value"]).copy())
            out.append(rec)
            agent.process(None, T * (it + 1))
        res[mode] = out
        col.terminate()
    for a, b in zip(res["0"], res["1"]):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("cont,capture", [(True, True), (True, False), (False, True)])
def test_collector_two_independent_halves_per_timestep_is_bit_identical(cont, capture, monkeypatch):
    """Round 6: 32 workers are exchanged with the acting kernel as two independent halves of 16 (jh_collect.hip: run_loop_split -- half A's round trip runs under half
    B's sampling, env steps and bookkeeping; tags per row tile in the kernel).  The same rows through the same arithmetic under the same sampling keys: the stored
    transitions and the captured heads / values of two consecutive rollouts are bit-identical to the one-exchange-of-32 path (JH_COLLECT_SPLIT=0), for
    config.ppo.mujoco's continuous policy and for a discrete one (CartPole, 32 workers: one timestep per exchange), with and without the acting-time capture."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    W, T, S, A = 32, 24, 11 if cont else 4, 3 if cont else 2
    if not capture:
        monkeypatch.setenv("JH_COLLECT_CAPTURE", "0")
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("JH_COLLECT_SPLIT", mode)
        torch.manual_seed(5)
        np.random.seed(5)
        agent = Agent("ppo", state_size=S, action_size=A, hidden_size=512, network="continuous_policy_value" if cont else "discrete_policy_value", n_step=T, batch_size=256,
                      n_epoch=1, device="cuda", seed=3, lr_decay=False, num_workers=W)
        agent.memory.first_store = False
        env = ops.ControlVec(W, S, A, seed=4) if cont else ops.CartPoleVec(W, seed=4)
        col = NativeCollector(env, agent, W)
        out = []
        for it in range(3):
            col.run(T)
            torch.cuda.synchronize()
            st, M = agent._static, W * T
            store = agent.memory._store
            rec = {k: npy(store.column(k)[:M]).copy() for k in ("state", "action", "reward", "next_state", "done")}
            if capture:
                rec.update(h0=npy(st["h0"]).copy(), value=npy(st["value"]).copy(), next_value=npy(st["next_value"]).copy())
            out.append(rec)
            agent.process(None, T * (it + 1))
        res[mode] = (out, col.stats())
        col.terminate()
    for a, b in zip(res["0"][0], res["1"][0]):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    assert len({tuple(r["action"].reshape(-1)[:64]) for r in res["1"][0]}) == 3  # (three different rollouts, not one repeated)
    print("acting host us per timestep: one exchange", res["0"][1]["act_us_per_step"] + res["0"][1]["env_us_per_step"], " two halves", res["1"][1]["act_us_per_step"] + res["1"][1]["env_us_per_step"])


@pytest.mark.parametrize("stall", ["5,0", "5,1"])
def test_collector_two_halves_hands_over_to_one_launch_per_step_when_the_kernel_gives_up(stall, monkeypatch):
    """A stalled environment (no observations for ~0.2 s) makes the persistent acting kernel give up; jh_collector_run then finishes the rollout with one launch per
    timestep.  With the two halves out of phase the hand-over happens either with both halves at the same timestep (the stall sits in front of half 0's read) or with
    half 1 of a timestep still owed (in front of half 1's: that half is finished from one launch over its 16 rows, sampled under its rows' own stream keys).  The
    test hook JH_COLLECT_TEST_STALL idles the host for 0.35 s at that point.  The rollout completes, every worker's trajectory is continuous, the next run is normal."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    W, T, S, A = 32, 16, 11, 3
    torch.manual_seed(5)
    np.random.seed(5)
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=512, network="continuous_policy_value", n_step=T, batch_size=256, n_epoch=1, device="cuda", seed=3,
                  lr_decay=False, num_workers=W)
    agent.memory.first_store = False
    col = NativeCollector(ops.ControlVec(W, S, A, seed=4), agent, W)
    for it, env_stall in enumerate((None, stall, None)):
        if env_stall is None:
            monkeypatch.delenv("JH_COLLECT_TEST_STALL", raising=False)
        else:
            monkeypatch.setenv("JH_COLLECT_TEST_STALL", env_stall)
        import time

        t_run = time.perf_counter()
        col.run(T)
        torch.cuda.synchronize()
        t_run = time.perf_counter() - t_run
        # the hook ran (0.35 s), the dead kernel was noticed at once (its give-up word: no 40 M-spin wait), the rest took per-step launches; an undisturbed run is ~1 ms
        assert (0.35 <= t_run < 2.0) if env_stall else (it == 0 or t_run < 0.2), t_run  # (the first run of a process may include the kernel's module load)
        M = W * T
        store = agent.memory._store
        rec = {k: npy(store.column(k)[:M]).reshape(W, T, -1) for k in ("state", "action", "reward", "next_state", "done")}
        assert all(np.isfinite(v).all() for v in rec.values()) and np.abs(rec["action"]).max() <= 1.0
        cont = rec["done"][:, :-1, 0] == 0  # where the episode goes on, the next row starts where this one ended
        assert np.array_equal(rec["next_state"][:, :-1][cont], rec["state"][:, 1:][cont])
        assert len(np.unique(rec["action"].reshape(M, -1), axis=0)) > M // 2  # sampled, not a constant
        st = agent._static
        assert np.isfinite(npy(st["value"])).all() and np.isfinite(npy(st["next_value"])).all() and np.isfinite(npy(st["h0"])).all()
        result = agent.process(None, T * (it + 1))
        assert np.isfinite(result["actor_loss"]) and np.isfinite(result["critic_loss"])
    col.terminate()


@pytest.mark.parametrize("forkable,where", [(False, "step"), (True, "step"), (True, "copy_row")])
def test_c_collector_stops_when_the_env_fails(forkable, where, capfd):
    """jh_env_vtbl's contract: obs / step return a negative status and the run STOPS.  A Python env that raises in the middle of a rollout --
    in step (one and two timesteps per exchange: the speculative steps of the lookahead dropped the status until round 6) or in copy_row
    (no status in the table: recorded, returned by the next callback) -- makes run() raise, and the failed run appends NOTHING to the
    rollout store (its staging rows are half written).  The collector is usable again afterwards."""
    from jorldy_amd import _lib as L
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    W, T = 8, 16
    torch.manual_seed(1)
    np.random.seed(1)
    agent = Agent("ppo", state_size=4, action_size=2, hidden_size=64, n_step=T, batch_size=64, n_epoch=1, device="cuda", seed=3, lr_decay=False)
    agent.memory.first_store = False
    env = _PyCartPole(W, seed=4, forkable=forkable)
    calls = {"n": 0, "armed": False}
    inner = env.step if where == "step" else env.copy_row

    def flaky(*a, **k):
        calls["n"] += 1
        if calls["armed"] and calls["n"] > 9:
            raise RuntimeError("simulator fell over")
        return inner(*a, **k)

    if where == "step":
        env.step = flaky
        if forkable:
            env.fork = lambda rows: _flaky_fork(rows, flaky_owner=calls)
    else:
        env.copy_row = flaky
    col = NativeCollector(env, agent, W)
    col.run(T)  # a healthy run first
    agent.process(None, T)
    torch.cuda.synchronize()
    assert agent.memory.size == 0
    calls["armed"], calls["n"] = True, 0
    with pytest.raises(L.JhError):
        col.run(T)
    torch.cuda.synchronize()
    assert agent.memory.size == 0, "a failed run appended rows"
    assert "simulator fell over" in capfd.readouterr().err
    calls["armed"] = False
    col.run(T)  # and the collector still works
    torch.cuda.synchronize()
    assert agent.memory.size == W * T
    col.terminate()


def _flaky_fork(rows, flaky_owner):
    """A forked scratch env whose step shares the failure switch of the env it was forked from."""
    e = _PyCartPole(rows, 0, True)
    inner = e.step

    def step(*a, **k):
        flaky_owner["n"] += 1
        if flaky_owner["armed"] and flaky_owner["n"] > 9:
            raise RuntimeError("simulator fell over")
        return inner(*a, **k)

    e.step = step
    e.fork = lambda r: _flaky_fork(r, flaky_owner)
    return e


@pytest.mark.parametrize("W", [8, 32])
@pytest.mark.parametrize("cont", [False, True])
@pytest.mark.parametrize("persistent,early", [(True, False), (False, False), (True, True)])
def test_collector_capture_equals_the_learners_own_no_grad_passes(cont, persistent, early, W, monkeypatch):
    """Acting-time capture (jh_collector_set_capture): the raw heads / V(s) / V(s') the acting kernel computed while collecting
    replace the two no-grad passes at the start of PPO.learn (ppo.py:83-94).  Same weights, same inputs: the captured numbers
    equal the learner's own pass up to fp32 summation order, next_value == V(next_state) wherever done = 0, and three
    iterations of collect + learn (with the pre-drawn index lists and the pre-launched acting kernel) give the same losses
    (1e-5) as the capture-free path.  early: the commit launch and the learner's launches are enqueued BEFORE the rollout's host
    loop (NativeCollector.begin / agent.process_begin / loop / process_end; the commit is gated by a flag the loop sets last)."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector

    if W == 32 and not (cont and persistent and not early):
        pytest.skip("32 workers x 11 observations (352 granules, three poll instructions, two row tiles): the continuous persistent form only")
    monkeypatch.setenv("JH_COLLECT_PERSISTENT", "1" if persistent else "0")
    T, H = 64, 512 if persistent else 64
    S, A = (11, 3) if cont else (4, 2)
    res = {}
    for capture in (True, False):
        monkeypatch.setenv("JH_COLLECT_CAPTURE", "1" if capture else "0")
        torch.manual_seed(5)
        np.random.seed(5)
        agent = Agent("ppo", state_size=S, action_size=A, hidden_size=H, network="continuous_policy_value" if cont else "discrete_policy_value",
                      n_step=T, batch_size=128, n_epoch=2, device="cuda", backend="native", seed=3, lr_decay=True, run_step=10000)
        agent.memory.first_store = False
        env = (ops.ControlVec(W, S, A, seed=4) if cont else ops.CartPoleVec(W, seed=4))
        col = NativeCollector(env, agent, W)
        assert col.capture == capture
        out = []
        for it in range(4):
            split = early and capture and it > 0
            if split:
                col.begin(T)
                agent.process_begin(T * (it + 1))
                col.loop()
                if it < 3:
                    col.arm_prelaunch(T)
                out.append(agent.process_end())
                assert agent._captured == 0
                continue
            col.run(T)
            if capture and it == 0:
                torch.cuda.synchronize()
                st, M = agent._static, W * T
                cap = [npy(t).copy() if t is not None else None for t in (st["h0"], st["h1"], st["value"], st["next_value"])]
                x = agent.memory._store.column("state")[:M].float()
                nx = agent.memory._store.column("next_state")[:M].float()
                dn = npy(agent.memory._store.column("done")[:M]).reshape(-1).astype(bool)
                o = agent._net.forward(x)
                own = [npy(o[0]).copy(), npy(o[1]).copy() if cont else None, npy(o[-1]).copy()]
                own_nv = npy(agent._net.forward(nx)[-1]).copy()
                np.testing.assert_allclose(cap[0], own[0], rtol=2e-5, atol=2e-6)
                if cont:
                    np.testing.assert_allclose(cap[1], own[1], rtol=2e-5, atol=2e-6)
                np.testing.assert_allclose(cap[2].reshape(-1), own[2].reshape(-1), rtol=2e-5, atol=2e-6)
                np.testing.assert_allclose(cap[3].reshape(-1)[~dn], own_nv.reshape(-1)[~dn], rtol=2e-5, atol=2e-6)
                assert np.all(np.isfinite(cap[3]))
            if it < 3:
                col.arm_prelaunch(T)
            out.append(agent.process(None, T * (it + 1)))
            assert agent._captured == 0
        torch.cuda.synchronize()
        res[capture] = out
        col.terminate()
    for a, b in zip(res[True], res[False]):
        for k in ("actor_loss", "critic_loss", "entropy_loss", "mean_ret"):
            margins.leq(abs(a[k] - b[k]), 1e-5 * (1.0 + abs(b[k])), f"result {k}")


@pytest.mark.parametrize("persistent", [True, False])
@pytest.mark.parametrize("H,W", [(64, 5), (512, 8), (512, 32), (128, 19), (256, 16)])
def test_native_collector_continuous_policy_on_control_env(H, W, persistent, monkeypatch):
    """jh_collector_create_control / jh_collector_run (config.ppo.mujoco shapes: S = 11, A = 3, continuous): the stored
    worker-major transitions replayed through the oracle env reproduce states / rewards / dones bit for bit; the stored
    actions are tanh-squashed samples of the policy (persistent acting kernel with W1 in LDS for S > 8 and 88
    observation granules -- 352 for the config's 32 workers, three poll instructions per poll and two row tiles; 209 for 19; 176 for the
    16 workers per GPU of a two-GPU run: two instructions, one row tile --, or
    one launch per timestep); the learner consumes the rollout."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector
    from oracle.jorldy_oracle import ControlOracle

    monkeypatch.setenv("JH_COLLECT_PERSISTENT", "1" if persistent else "0")
    S, A, T = 11, 3, 40
    torch.manual_seed(3)
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=H, network="continuous_policy_value", n_step=T, batch_size=50, device="cuda", backend="native", seed=5)
    agent.memory.first_store = False
    env = ops.ControlVec(W, S, A, seed=9)
    col = NativeCollector(env, agent, W)
    col.run(T)
    torch.cuda.synchronize()
    st = agent.memory._store
    assert st.size == W * T
    cols = {k: npy(st.column(k)[: W * T]) for k in ("state", "action", "reward", "next_state", "done")}
    assert cols["action"].shape == (W * T, A) and cols["action"].dtype == np.float32 and np.all(np.abs(cols["action"]) <= 1.0)
    orc = ControlOracle(W, S, A, seed=9)
    a = cols["action"].reshape(W, T, A)
    for t in range(T):
        obs = orc.obs()
        nxt, rew, done = orc.step(a[:, t])
        rows = np.arange(W) * T + t  # worker-major
        np.testing.assert_array_equal(cols["state"][rows], obs)
        np.testing.assert_array_equal(cols["next_state"][rows], nxt)
        np.testing.assert_array_equal(cols["reward"][rows, 0], rew)
        np.testing.assert_array_equal(cols["done"][rows, 0].astype(bool), done)
    # the actions are samples of THIS policy at the stored states: z = atanh(a) ~ Normal(mu, std)
    with torch.no_grad():
        mu, std, _ = _policy64(agent, "continuous_policy_value", S, A, H)(torch.from_numpy(cols["state"]).double())
    zz = np.arctanh(np.clip(cols["action"].astype(np.float64), -1 + 1e-7, 1 - 1e-7))
    zs = (zz - npy(mu)) / npy(std)
    assert abs(zs.mean()) < 0.12 and 0.85 < zs.std() < 1.15, (zs.mean(), zs.std())
    res = agent.process(None, T)
    assert set(res) == {"actor_loss", "critic_loss", "entropy_loss", "max_ratio", "min_prob", "mean_ret"} and agent.memory.size == 0


def test_native_act_continuous_distribution():
    """PPO.act for a continuous policy on the native path (jh_pponet_act_continuous): tanh(Normal(mu, std).sample()),
    tanh(mu) when not training (ppo.py:55-63)."""
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(0)
    agent = Agent("ppo", state_size=11, action_size=3, hidden_size=64, network="continuous_policy_value", device="cuda", backend="native")
    with torch.no_grad():
        agent.network.mu.bias.copy_(torch.tensor([0.3, -0.5, 0.0], device="cuda"))
        agent.network.log_std.bias.copy_(torch.tensor([-0.5, 0.0, 0.4], device="cuda"))
    obs = np.zeros((64, 11), np.float32)
    zs = []
    for _ in range(600):  # 38 400 samples per dimension: the mean's sampling error is ~0.0075 against atol 0.04
        a = agent.act(obs, training=True)["action"]
        assert a.shape == (64, 3) and a.dtype == np.float32
        zs.append(np.arctanh(np.clip(a.astype(np.float64), -1 + 1e-7, 1 - 1e-7)))
    zs = np.concatenate(zs, 0)
    with torch.no_grad():
        mu, std, _ = _policy64(agent, "continuous_policy_value", 11, 3, 64)(torch.zeros(1, 11, dtype=torch.float64))
    np.testing.assert_allclose(zs.mean(0), npy(mu)[0], atol=0.04)
    np.testing.assert_allclose(zs.std(0), npy(std)[0], rtol=0.05)
    g = agent.act(obs, training=False)["action"]
    np.testing.assert_allclose(g, np.tile(np.tanh(npy(mu)), (64, 1)), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cont,S,A", [(True, 17, 6), (True, 27, 8), (False, 6, 12)])
def test_native_act_with_more_than_8_head_outputs(cont, S, A):
    """config.ppo.mujoco on HalfCheetah / Ant (2 A + 1 = 13 / 17 head outputs) and a 12-action discrete policy: acting on the separate-call
    forward (round 5; VERDICT r4 missing #6).  Raw heads against the reference's modules in float64, the greedy action exact
    in the heads, samples distributed like the policy."""
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(1)
    net = "continuous_policy_value" if cont else "discrete_policy_value"
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=64, network=net, device="cuda", backend="native", seed=3)
    with torch.no_grad():
        for p in agent.network.parameters():
            p.add_(0.05 * torch.randn_like(p))  # (means well inside the +-5 clamp: atanh of a float32 action near +-1 cannot resolve |z| > 4)
    rng = np.random.RandomState(0)
    obs = rng.randn(37, S).astype(np.float32)
    pol = _policy64(agent, net, S, A, 64)
    if cont:
        act, mu_raw, ls_raw = agent._net.act_continuous(obs, training=False, want_heads=True)
        with torch.no_grad():
            mu, std, _ = pol(torch.from_numpy(obs).double())
        np.testing.assert_allclose(np.clip(mu_raw, -5, 5), npy(mu), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(np.exp(np.tanh(ls_raw)), npy(std), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(act, np.tanh(np.clip(mu_raw, -5, 5)), rtol=1e-6, atol=1e-6)
        one = np.repeat(obs[:1], 64, 0)
        zs = np.concatenate([np.arctanh(np.clip(agent.act(one, True)["action"].astype(np.float64), -1 + 1e-7, 1 - 1e-7)) for _ in range(300)], 0)
        ok = np.abs(npy(mu)[0]) < 2.5  # the sample statistics come back through tanh / atanh in float32
        assert ok.sum() >= A // 2
        np.testing.assert_allclose(zs.mean(0)[ok], npy(mu)[0][ok], atol=0.06)
        np.testing.assert_allclose(zs.std(0)[ok], npy(std)[0][ok], rtol=0.06)
    else:
        act, logits, val = agent._net.act_discrete(obs, training=False, want_logits=True)
        with torch.no_grad():
            pi, v = pol(torch.from_numpy(obs).double())
        p_ours = np.exp(logits - logits.max(1, keepdims=True))
        np.testing.assert_allclose(p_ours / p_ours.sum(1, keepdims=True), npy(pi), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(val, npy(v), rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(act.reshape(-1), logits.argmax(1))
        one = np.repeat(obs[:1], 64, 0)
        cnt = np.bincount(np.concatenate([agent.act(one, True)["action"].reshape(-1) for _ in range(300)]), minlength=A) / (300 * 64)
        np.testing.assert_allclose(cnt, npy(pi)[0], atol=0.02)


def test_native_collector_32_workers_wide_action_space_end_to_end():
    """configs[4]'s worker count (config.ppo.mujoco: 32 workers) on ONE GPU with an action space wider than Hopper's (A = 6): the native
    collector (one acting forward per timestep: 2 A + 1 = 13 head outputs are beyond the persistent kernel's 12) feeds the learner, the stored
    transitions replay through the oracle env bit for bit, learn() consumes them on the separate-call path."""
    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import NativeCollector
    from oracle.jorldy_oracle import ControlOracle

    S, A, T, W = 17, 6, 24, 32
    torch.manual_seed(3)
    agent = Agent("ppo", state_size=S, action_size=A, hidden_size=64, network="continuous_policy_value", n_step=T, batch_size=256, n_epoch=2, device="cuda", backend="native", seed=5)
    agent.memory.first_store = False
    col = NativeCollector(ops.ControlVec(W, S, A, seed=9), agent, W)
    col.run(T)
    torch.cuda.synchronize()
    st = agent.memory._store
    assert st.size == W * T
    cols = {k: npy(st.column(k)[: W * T]) for k in ("state", "action", "reward", "next_state", "done")}
    orc = ControlOracle(W, S, A, seed=9)
    a = cols["action"].reshape(W, T, A)
    for t in range(T):
        obs = orc.obs()
        nxt, rew, done = orc.step(a[:, t])
        rows = np.arange(W) * T + t
        np.testing.assert_array_equal(cols["state"][rows], obs)
        np.testing.assert_array_equal(cols["next_state"][rows], nxt)
        np.testing.assert_array_equal(cols["reward"][rows, 0], rew)
    res = agent.process(None, T)
    assert set(res) == {"actor_loss", "critic_loss", "entropy_loss", "max_ratio", "min_prob", "mean_ret"} and all(np.isfinite(v) for v in res.values())


def test_ppo_native_data_parallel_path_single_rank_rccl():
    """The DP code path (eager native kernels + one RCCL all-reduce of the flat gradient bucket per
    minibatch) on a 1-rank nccl group: must equal the graph-replayed single-learner result."""
    import socket

    import torch.distributed as dist

    from jorldy_amd.core.agent import Agent
    from jorldy_amd.parallel import attach_data_parallel

    z = load("ppo_disc_cartpole")
    S, A, H, W, T, B, E, cont = [int(x) for x in z["cfg"]]
    gamma, lam, eps, vf, ent, clip, lr = z["hyper"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cols = {k: z[f"in_{k}"] for k in ["state", "next_state", "reward", "done", "action"]}
        outs = []
        for dp in (False, True):
            agent = Agent("ppo", state_size=S, action_size=A, hidden_size=H, optim_config={"name": "adam", "lr": lr}, batch_size=B, n_step=T, n_epoch=E,
                          _lambda=lam, epsilon_clip=eps, vf_coef=vf, ent_coef=ent, clip_grad_norm=clip, gamma=gamma, run_step=100000, device="cuda", backend="native")
            agent.network.load_state_dict(_sd(z, "sd0/"))
            agent.memory.first_store = False
            if dp:
                attach_data_parallel(agent, dist)
            np.random.seed(int(z["np_seed"]))
            agent.process(cols, T)
            outs.append(agent._net.params.clone())
        # same kernels except the global norm (fused per-GEMM partials vs the gradnorm pass after the
        # all-reduce): equal up to the rounding of that one reduction
        d = (outs[0] - outs[1]).abs()
        margins.lt(float((d > 2e-6).float().mean()), 0.005, "fraction of weights > 2e-6 apart (DP hook vs none)")
        margins.leq(float(d.max()), 2.1 * lr * int(z["n_minibatch"]), "worst weight difference (DP hook vs none)")
        _cmp_sd(agent.network, _sd(z, "sd1/"), lr, int(z["n_minibatch"]), atol=3e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("version", [1, 2])
@pytest.mark.parametrize("name,steps", [("per", 50), ("per", 12), ("dqn", 50), ("dqn", 12)])
def test_full_checkpoint_resumes_bit_identically(tmp_path, name, steps, version):
    """save_full/load_full: buffer rows, sum tree, counters, beta/epsilon and RNG state survive, so a resumed agent
    continues EXACTLY like the uninterrupted one (the reference restarts from an empty buffer).
    version 2 = manifest + raw column files streamed from HBM, version 1 = round 1's resume.pt pickle (must stay
    loadable); 50 steps wrap the 32-slot ring, 12 steps leave fewer rows than the agents' deferred-store threshold
    (a restore through the deferred path would append them BEHIND the restored ring position)."""
    import json

    from jorldy_amd.core.agent import Agent

    S, A = 6, 3
    mk = lambda: Agent(name, state_size=S, action_size=A, hidden_size=16, batch_size=8, start_train_step=5, buffer_size=32, run_step=200, learn_period=2, device="cuda",
                       use_graph=False)  # replayed-graph GEMMs may pick another hipBLASLt algorithm than the eager first call: last-bit differences
    np.random.seed(5)
    torch.manual_seed(5)
    a1 = mk()
    a1.memory.first_store = False
    assert a1.memory.defer_rows >= 12
    _interact(_MockEnv(S, A), a1, steps)
    a1.save_full(str(tmp_path), version=version)
    if version == 2:
        man = json.load(open(tmp_path / "resume" / "manifest.json"))
        assert man["format"] == "jorldy_amd.resume" and man["version"] == 2 and man["memory"]["buffer_counter"] == a1.memory.size
        assert all((tmp_path / "resume" / c["file"]).stat().st_size == c["rows"] * int(np.prod(c["shape"])) * np.dtype(c["dtype"]).itemsize for c in man["memory"]["columns"])
        assert not (tmp_path / "resume.pt").exists()
    else:
        assert (tmp_path / "resume.pt").exists()
    a2 = mk()
    a2.memory.first_store = False
    a2.load_full(str(tmp_path))
    assert a2.memory.size == a1.memory.size and a2.memory.buffer_index == a1.memory.buffer_index and a2.time_t == a1.time_t
    assert not a2.memory._pending
    for k in a1.memory._store.names:  # every stored row, slot by slot
        assert torch.equal(a1.memory._store.column(k)[: a1.memory.size], a2.memory._store.column(k)[: a2.memory.size]), k
    per = name == "per"
    if per:
        np.testing.assert_array_equal(a2.memory.sum_tree, a1.memory.sum_tree)
        assert a2.memory.max_priority == a1.memory.max_priority and a2.beta == a1.beta
    assert a2.epsilon == a1.epsilon
    env1, env2 = _MockEnv(S, A), _MockEnv(S, A)
    for ag, env in ((a1, env1), (a2, env2)):  # more stores + learns after the resume: same ring slots, same samples
        np.random.seed(99)
        torch.manual_seed(99)
        _interact(env, ag, 6, start=steps)
    for (k, v1), (_, v2) in zip(a1.network.state_dict().items(), a2.network.state_dict().items()):
        torch.testing.assert_close(v1, v2, rtol=0, atol=0)
    for k in a1.memory._store.names:
        assert torch.equal(a1.memory._store.column(k)[: a1.memory.size], a2.memory._store.column(k)[: a2.memory.size]), k
    if per:
        np.testing.assert_array_equal(a2.memory.sum_tree, a1.memory.sum_tree)


def test_resume_manifest_rejects_unknown_version(tmp_path):
    import json

    from jorldy_amd.core.agent import Agent

    a = Agent("dqn", state_size=4, action_size=2, hidden_size=16, batch_size=4, buffer_size=16, device="cuda")
    a.save_full(str(tmp_path))
    mp = tmp_path / "resume" / "manifest.json"
    man = json.load(open(mp))
    man["version"] = 3
    json.dump(man, open(mp, "w"))
    with pytest.raises(ValueError, match="unsupported resume manifest"):
        a.load_full(str(tmp_path))


@pytest.mark.parametrize("name,extra", [("dqn", {}), ("per", dict(learn_period=1)), ("ape_x", dict(n_step=3, num_workers=4, learn_period=1)),
                                         ("c51", dict(num_support=21, v_min=-2, v_max=5))])
def test_td_agents_graph_replay_equals_eager(name, extra):
    """learn() captured into one hipGraph (gather, forwards, HIP loss kernel, backward, capturable
    optimizer step, priority write-back) must reproduce the eager sequence update for update."""
    from jorldy_amd.core.agent import Agent

    fx = {"dqn": "dqn", "per": "per", "ape_x": "ape_x", "c51": "c51"}[name]
    z = load(fx)
    res = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        agent = Agent(name, state_size=int(_h(z, "S")), action_size=int(_h(z, "A")), hidden_size=int(_h(z, "H")), optim_config={"name": "adam", "lr": 1e-3},
                      buffer_size=256, batch_size=int(_h(z, "B")), start_train_step=0, target_update_period=10000, run_step=1000, device="cuda",
                      use_graph=use_graph, **extra)
        agent.network.load_state_dict(_sd(z, "sd0/"))
        agent.target_network.load_state_dict(_sd(z, "sdt/"))
        _fill_from_fixture(agent, z, name in ("per", "ape_x"))
        np.random.seed(7)
        out = []
        for it in range(5):
            r = agent.learn()
            agent.learning_rate_decay(10 * (it + 1))
            out.append(r["loss"])
        if use_graph:
            assert agent._graph is not None, "learn() was not captured"
        res.append((out, torch.cat([p.detach().reshape(-1) for p in agent.network.parameters()]).clone(),
                    agent.memory.sum_tree if hasattr(agent.memory, "_tree") else None))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-5, atol=1e-6)
    if res[0][2] is not None:
        np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-4, atol=1e-7)  # priorities = |fp32 TD error|^alpha


def test_deferred_stores_are_invisible_per_buffer():
    """Coalesced per-step stores (ReplayBuffer.defer_rows) == immediate stores: same ring contents, same sum
    tree (bit-exact), same samples -- across ring wrap, mixed actor-side / max priorities and priority updates."""
    from jorldy_amd.core.buffer import PERBuffer

    rng = np.random.RandomState(3)
    bufs = [PERBuffer(24, 0.1, device="cuda"), PERBuffer(24, 0.1, device="cuda")]
    bufs[1].defer_rows = 4
    for b in bufs:
        b.first_store = False
    for step in range(70):
        t = {"state": rng.randn(1, 5).astype(np.float32), "action": rng.randint(0, 3, size=(1, 1)), "reward": rng.randn(1, 1),
             "next_state": rng.randn(1, 5).astype(np.float32), "done": np.asarray([[rng.rand() < 0.2]])}
        if step % 3 == 0:
            t["priority"] = np.asarray([[rng.rand() + 0.1]])
        for b in bufs:
            b.store([dict(t)])
        assert bufs[0].size == bufs[1].size and bufs[0].buffer_index == bufs[1].buffer_index
        if step % 7 == 6:
            outs = []
            for b in bufs:
                np.random.seed(step)
                tr, w, idx, sp, mp = b.sample(0.5, 6)
                b.update_priorities(idx, (w * 0 + 1.0 + 0.01 * step).double() ** 0.5)
                outs.append((tr, w, idx, float(sp), float(mp)))
            assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][3:] == outs[1][3:]
            for k in outs[0][0]:
                assert torch.equal(outs[0][0][k], outs[1][0][k]), k
    for b in bufs:
        b.store([dict(t)])
    assert bufs[1]._pending_rows > 0 and bufs[0]._pending_rows == 0  # something is still held ...
    np.testing.assert_array_equal(bufs[0].sum_tree, bufs[1].sum_tree)  # ... and reading the tree flushes it
    assert bufs[0].max_priority == bufs[1].max_priority and bufs[0].tree_index == bufs[1].tree_index
    sd0, sd1 = bufs[0].state_dict(), bufs[1].state_dict()
    for k in sd0["columns"]:
        np.testing.assert_array_equal(sd0["columns"][k], sd1["columns"][k])


def test_rainbow_native_data_parallel_hook_single_rank_rccl():
    """attach_data_parallel on a 1-rank RCCL group: the all-reduce of the flat gradient bucket sits between
    backward and Adam (captured into the learn() graph when RCCL allows it); with world size 1 the mean is the
    identity, so the run must reproduce the hook-free learner bit for bit."""
    import torch.distributed as dist

    from jorldy_amd.core.agent import Agent
    from jorldy_amd.parallel import attach_data_parallel

    z = load("rainbow")
    H, A, K = int(_h(z, "H")), int(_h(z, "A")), int(_h(z, "num_support"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        res = []
        for dp in (False, True):
            torch.manual_seed(0)
            agent = Agent("rainbow", state_size=int(z["hyper/S"]), action_size=A, hidden_size=H, optim_config={"name": "adam", "lr": 1e-3}, buffer_size=256,
                          batch_size=int(_h(z, "B")), start_train_step=0, run_step=1000, n_step=3, num_support=K, device="cuda")
            agent.network.load_state_dict(_sd(z, "sd0/"))
            agent.target_network.load_state_dict(_sd(z, "sdt/"))
            _fill_from_fixture(agent, z, True)
            if dp:
                attach_data_parallel(agent, dist)
                assert agent.grad_sync is not None and agent.memory._shards is not None  # PER shard: global IS-weight normalisation
            np.random.seed(3)
            torch.manual_seed(4)
            losses = [agent.learn()["loss"] for _ in range(4)]
            res.append((losses, agent._net.params.clone()))
        # one shard = the whole logical buffer: the sharded weights (float64 torch ops + all-gather) must equal the
        # sum-tree kernel's local normalisation up to the last bit of pow()
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-6)
        torch.testing.assert_close(res[0][1], res[1][1], rtol=0, atol=1e-6)
    finally:
        dist.destroy_process_group()


def test_nstep_assemblers_of_the_agents_match_reference_fixture():
    """interact_callback of Rainbow / Multistep / Ape-X (rainbow.py:294-308, multistep.py:90-104,
    ape_x.py:174-199) against the windows the reference emitted for the same 12 raw transitions (two episode
    ends inside: windows straddle them; Ape-X adds the actor-side priority |G_n - q_t|)."""
    from jorldy_amd.core.agent import Agent

    z = load("nstep")
    keys = ["state", "action", "reward", "next_state", "done"]
    S, A = z["in_state"].shape[1], 2
    agents = {
        "rainbow": Agent("rainbow", state_size=S, action_size=A, hidden_size=8, n_step=3, buffer_size=16, device="cuda"),
        "multistep": Agent("multistep", state_size=S, action_size=A, hidden_size=8, n_step=3, buffer_size=16, device="cuda"),
        "apex": Agent("ape_x", state_size=S, action_size=A, hidden_size=8, n_step=3, buffer_size=16, num_workers=4, device="cuda"),
    }
    for name, agent in agents.items():
        ref = "rainbow" if name == "multistep" else name  # multistep.py's assembler is the same window as rainbow.py's
        out, emitted = {}, []
        for i in range(z["in_state"].shape[0]):
            t = {k: z[f"in_{k}"][i : i + 1] for k in keys}
            if name == "apex":
                t["q"] = z["in_q"][i : i + 1]
            e = agent.interact_callback(t)
            emitted.append(bool(e))
            for k, v in e.items():
                out.setdefault(k, []).append(np.asarray(v))
        np.testing.assert_array_equal(emitted, z[f"{ref}_emitted"])
        for k, v in out.items():
            np.testing.assert_array_equal(np.concatenate(v, 0), z[f"{ref}_{k}"], err_msg=f"{name}.{k}")


def test_multimodal_list_valued_keys_through_the_device_store():
    """base.py:42-56 stack_transition stacks list-valued keys (multimodal observations: [image, vector]) per
    element; the device store keeps one column per element (uint8 stays uint8) and gather returns lists."""
    from jorldy_amd.core.buffer import ReplayBuffer
    from oracle import jorldy_oracle as O

    rng = np.random.RandomState(0)
    mk = lambda: [rng.randint(0, 256, size=(1, 2, 6, 5)).astype(np.uint8), rng.randn(1, 3).astype(np.float32)]
    trs = [{"state": mk(), "action": rng.randint(0, 4, size=(1, 1)), "reward": rng.randn(1, 1), "next_state": mk(), "done": np.asarray([[i % 5 == 4]])} for i in range(11)]
    buf = ReplayBuffer(8, device="cuda")  # wraps
    ora = O.ReplayOracle(8)
    for i in range(0, 11, 3):
        buf.store(trs[i : i + 3])
        ora.store(trs[i : i + 3])
    assert buf.size == ora.size == 8 and buf.buffer_index == ora.buffer_index
    np.random.seed(4)
    want = ora.sample(6)
    np.random.seed(4)
    got = buf.sample(6, as_float=False)
    assert isinstance(got["state"], list) and got["state"][0].dtype == torch.uint8
    for k in ("state", "next_state"):
        for a, b in zip(got[k], want[k]):
            np.testing.assert_array_equal(a.cpu().numpy(), b)
    for k in ("action", "reward", "done"):
        # float64 rewards become fp32 on the device, as in the reference's as_tensor (base.py:61-73)
        np.testing.assert_array_equal(got[k].cpu().numpy().astype(np.float32), np.asarray(want[k]).astype(np.float32))


def test_staging_ring_drain_equals_direct_stores_and_survives_concurrent_actors():
    """PERBuffer.make_ring / drain (jh_ring_drain: hipMemcpyAsync from the pinned ring slots into the device ring
    + leaves with the actors' priorities): (1) one producer, several drains incl. both wrap-arounds == the same rows
    through store_soa, bit for bit (rows, tree, counters); (2) 6 concurrent actor threads while the learner drains
    and samples: every row lands exactly once and the tree's root is the sum of the priorities."""
    import threading

    from jorldy_amd.core.buffer import PERBuffer

    rng = np.random.RandomState(1)

    def batch(n, base):
        ids = np.arange(base, base + n)
        return {"state": rng.randint(0, 256, size=(n, 2, 4, 4)).astype(np.uint8), "action": ids.reshape(-1, 1), "reward": rng.randn(n, 3, 1).astype(np.float32),
                "next_state": rng.randint(0, 256, size=(n, 2, 4, 4)).astype(np.uint8), "done": (rng.rand(n, 3, 1) < 0.1)}, rng.rand(n) + 0.05

    direct, ringed = PERBuffer(40, 0.1, device="cuda"), PERBuffer(40, 0.1, device="cuda")
    direct.first_store = ringed.first_store = False
    ex, _ = batch(1, 0)
    ring = ringed.make_ring(16, example=ex, with_priority=True)
    base = 0
    for n in (7, 9, 16, 3, 11, 16, 5):  # 67 rows through a 16-slot ring into a 40-slot store
        cols, prio = batch(n, base)
        base += n
        direct.store_soa(cols, prio)
        ring.produce(ringed.ring_columns(cols), prio, timeout_ms=5000)
        assert ringed.drain() == n
        ring.reclaim(wait=True)  # single-threaded here: nobody else would recycle the slots before the next produce
        assert ringed.size == direct.size and ringed.buffer_index == direct.buffer_index
    np.testing.assert_array_equal(ringed.sum_tree, direct.sum_tree)
    assert ringed.max_priority == direct.max_priority and ringed.tree_index == direct.tree_index
    a, b = direct.state_dict()["columns"], ringed.state_dict()["columns"]
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)

    big = PERBuffer(4096, 1e-3, device="cuda")
    big.first_store = False
    ring = big.make_ring(256, example=ex, with_priority=True)
    P, CH, ROWS = 6, 40, 8
    sent = {}

    def actor(pid):
        r = np.random.RandomState(100 + pid)
        for c in range(CH):
            ids = np.arange((pid * CH + c) * ROWS, (pid * CH + c + 1) * ROWS)
            cols = {"state": r.randint(0, 256, size=(ROWS, 2, 4, 4)).astype(np.uint8), "action": ids.reshape(-1, 1), "reward": np.zeros((ROWS, 3, 1), np.float32),
                    "next_state": np.zeros((ROWS, 2, 4, 4), np.uint8), "done": np.zeros((ROWS, 3, 1), bool)}
            prio = 0.5 + (ids % 7)
            sent[(pid, c)] = (ids, cols["state"].copy(), prio)
            ring.produce(big.ring_columns(cols), prio, timeout_ms=30000)

    threads = [threading.Thread(target=actor, args=(p,)) for p in range(P)]
    for t in threads:
        t.start()
    total, got = P * CH * ROWS, 0
    import time

    t0 = time.time()
    while got < total and time.time() - t0 < 60:
        got += big.drain()
        if big.size >= 16:
            big.sample(0.4, 16)  # the learner keeps sampling while actors produce
    for t in threads:
        t.join(timeout=30)
    assert got == total == big.size
    cols = big.state_dict()["columns"]
    ids = cols["action"][:, 0]
    assert sorted(ids.tolist()) == list(range(total))
    by_id = {int(i): k for k, i in enumerate(ids)}
    for (pid, c), (sid, st, prio) in sent.items():
        rows = [by_id[int(i)] for i in sid]
        assert rows == list(range(rows[0], rows[0] + ROWS))  # one produce() call stays contiguous
        np.testing.assert_array_equal(cols["state"][rows], st)
    tree = big.sum_tree
    np.testing.assert_allclose(tree[0], sum(float(p.sum()) for _, _, p in sent.values()), rtol=1e-12)
    leaves = tree[big.first_leaf_index : big.first_leaf_index + total]
    np.testing.assert_array_equal(leaves, 0.5 + (ids % 7))  # every leaf carries ITS row's actor-side priority


def _frame_stream(rng, steps, C=4, H=12, W=10, n=3, p_done=0.08):
    """Frame-stacking env wrapper + n-step assembler as the reference wires them (atari.py:147-149 stack of the last
    C frames, reset repeats the first frame; rainbow.py:294-308 windows that straddle episode ends)."""
    from collections import deque

    frames, out, win = None, [], deque(maxlen=n)
    for t in range(steps):
        if frames is None:
            f0 = rng.randint(0, 256, size=(H, W)).astype(np.uint8)
            frames = deque([f0] * C, maxlen=C)
        state = np.stack(frames, 0)[None]
        done = rng.rand() < p_done
        frames.append(rng.randint(0, 256, size=(H, W)).astype(np.uint8))
        nxt = np.stack(frames, 0)[None]
        win.append({"state": state, "action": np.asarray([[t % 4]]), "reward": np.asarray([[float(t)]]), "next_state": nxt, "done": np.asarray([[done]])})
        if done:
            frames = None
        if len(win) == n:
            out.append({"state": win[0]["state"], "action": win[0]["action"], "next_state": win[-1]["next_state"],
                        "reward": np.stack([w["reward"] for w in win], 1), "done": np.stack([w["done"] for w in win], 1)})
    return out


def test_frame_dedup_replay_is_invisible_and_stores_one_frame_per_step():
    """frame_dedup=True (single frames in a device pool + slot numbers per transition, frames recognised by content
    hash): samples, trees and checkpoints equal the plain buffer's bit for bit across ring wrap (slot recycling),
    single and bulk stores; ~1 new frame per env step is uploaded instead of 2 x C."""
    from jorldy_amd.core.buffer import PERBuffer

    rng = np.random.RandomState(7)
    trs = _frame_stream(rng, 260)
    plain, dd = PERBuffer(64, 0.05, device="cuda"), PERBuffer(64, 0.05, device="cuda", frame_dedup=True, frame_pool_factor=1.6)
    for b in (plain, dd):
        b.first_store = False
        b.defer_rows = 4
    i = 0
    for step, chunk in enumerate([1, 1, 1, 5, 1, 1, 17, 1, 1, 1, 1, 30, 1, 1, 1] * 40):
        if i + chunk > len(trs):
            break
        for b in (plain, dd):
            b.store(trs[i : i + chunk])
        i += chunk
        if step % 6 == 5:
            outs = []
            for b in (plain, dd):
                np.random.seed(step)
                tr, w, idx, sp, mp = b.sample(0.5, 8, as_float=(step % 12 == 5))
                b.update_priorities(idx, w.double() * 0 + 0.3 + 0.01 * step)
                outs.append((tr, w, idx))
            assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][1], outs[1][1])
            for k in outs[0][0]:
                assert outs[0][0][k].dtype == outs[1][0][k].dtype and torch.equal(outs[0][0][k], outs[1][0][k]), k
    assert i > 200 and dd.size == plain.size == 64
    np.testing.assert_array_equal(plain.sum_tree, dd.sum_tree)
    a, b = plain.state_dict(), dd.state_dict()
    for k in a["columns"]:
        np.testing.assert_array_equal(a["columns"][k], b["columns"][k], err_msg=k)
    st = dd._frames.stats()
    per_tr = st["frames_uploaded"] / i
    assert per_tr < 1.6, st  # 2 x C = 8 without de-duplication; ~1 + (C - 1) * P(reset)
    assert st["pool_in_use"] <= 64 * 1.6 + 80
    # round trip through a checkpoint (portable: full stacks) into a fresh de-duplicating buffer
    dd2 = PERBuffer(64, 0.05, device="cuda", frame_dedup=True)
    dd2.load_state_dict(b)
    c = dd2.state_dict()
    for k in a["columns"]:
        np.testing.assert_array_equal(a["columns"][k], c["columns"][k], err_msg=k)
    np.testing.assert_array_equal(dd2.sum_tree, plain.sum_tree)
    # the streamed form (resume format version 2): de-duplicated -> files of full stacks -> plain AND de-duplicating buffers
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        meta = dd.save_stream(d)
        assert meta["frame_dedup"] and all(c["rows"] == dd.size for c in meta["columns"])
        for tgt in (PERBuffer(64, 0.05, device="cuda"), PERBuffer(64, 0.05, device="cuda", frame_dedup=True)):
            tgt.load_stream(d, meta)
            e = tgt.state_dict()
            for k in a["columns"]:
                np.testing.assert_array_equal(a["columns"][k], e["columns"][k], err_msg=k)
            np.testing.assert_array_equal(tgt.sum_tree, plain.sum_tree)
            assert tgt.buffer_index == plain.buffer_index and tgt.size == plain.size


def test_rainbow_native_learns_identically_from_a_deduplicated_replay():
    """Rainbow (native CNN backend, hipGraph learn) on frame_dedup storage == on plain storage: same losses, same weights."""
    from jorldy_amd.core.agent import Agent

    rng = np.random.RandomState(3)
    trs = _frame_stream(rng, 120, C=4, H=44, W=52, n=3)
    res = []
    for dedup in (False, True):
        torch.manual_seed(0)
        agent = Agent("rainbow", state_size=(4, 44, 52), action_size=4, hidden_size=32, head="cnn", optim_config={"name": "adam", "lr": 1e-3}, buffer_size=96,
                      batch_size=8, start_train_step=0, run_step=1000, n_step=3, num_support=11, v_min=-1, v_max=10, device="cuda", frame_dedup=dedup)
        agent.memory.first_store = False
        np.random.seed(5)
        torch.manual_seed(6)
        losses = []
        for t in trs:
            agent.memory.store([t])
            if agent.memory.size >= 16 and len(losses) < 12 and agent.memory.size % 6 == 0:
                losses.append(agent.learn()["loss"])
        res.append((losses, agent._net.params.clone()))
        if dedup:
            assert agent._graph is not None and agent.memory._frames is not None
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_full_checkpoint_restores_the_native_rng_streams(tmp_path):
    """Rainbow's learner noise (ops.NormalSource: seed + call counter in device memory) and PPO's host-side action
    sampling stream (seed, acting-step counter) are part of resume format 2: a resumed run draws the same NoisyNet noise /
    samples the same actions as the uninterrupted one."""
    from jorldy_amd.core.agent import Agent

    z = load("rainbow")
    H, A, K = int(_h(z, "H")), int(_h(z, "A")), int(_h(z, "num_support"))

    def mk():
        torch.manual_seed(0)
        np.random.seed(0)
        ag = Agent("rainbow", state_size=int(z["hyper/S"]), action_size=A, hidden_size=H, optim_config={"name": "adam", "lr": 1e-3}, buffer_size=256,
                   batch_size=int(_h(z, "B")), start_train_step=0, run_step=1000, n_step=3, num_support=K, device="cuda", backend="native", use_graph=False)
        ag.network.load_state_dict(_sd(z, "sd0/"))
        ag.target_network.load_state_dict(_sd(z, "sdt/"))
        _fill_from_fixture(ag, z, True)
        return ag

    a = mk()
    torch.manual_seed(3)
    for _ in range(4):
        a.learn()
    os.makedirs(tmp_path / "rb")
    a.save_full(str(tmp_path / "rb"))
    want = [a.learn()["loss"] for _ in range(4)]
    b = mk()
    torch.manual_seed(12345)  # whatever the process' torch generator holds: the restored stream must not depend on it
    b.load_full(str(tmp_path / "rb"))
    got = [b.learn()["loss"] for _ in range(4)]
    assert want == got
    torch.testing.assert_close(a._net.params, b._net.params, rtol=0, atol=0)

    def mkp():
        torch.manual_seed(0)
        return Agent("ppo", state_size=4, action_size=2, hidden_size=64, n_step=16, batch_size=16, device="cuda", backend="native", seed=9)

    p = mkp()
    obs = np.random.RandomState(1).randn(8, 4).astype(np.float32)
    for _ in range(5):
        p.act(obs, training=True)
    os.makedirs(tmp_path / "ppo")
    p.save_full(str(tmp_path / "ppo"))
    want_a = [p.act(obs, training=True)["action"].copy() for _ in range(20)]
    q = mkp()
    q.load_full(str(tmp_path / "ppo"))
    assert q._net.act_rng()[1] == 5
    got_a = [q.act(obs, training=True)["action"].copy() for _ in range(20)]
    assert all(np.array_equal(x, y) for x, y in zip(want_a, got_a)) and len({a.tobytes() for a in want_a}) > 1
