"""GPU parity tests of the native value networks (jh_rbnet_*: implicit-GEMM convolutions, noisy dueling heads, backward, Adam /
centered RMSprop) against FLOAT64 ground truth: the reference's modules (core/network/rainbow.py:8-94, dueling.py:8-35,
q_network.py:8-20, head.py:6-61; restated in tests/mirror) evaluated on the CPU in float64 with the same parameters, inputs and
NoisyNet draws, and torch-CPU float32 (the oracle's arithmetic) beside it.  Criterion: tests/fp64_truth.py --
|ours - exact| <= max(1e-5, 2 x |torch_cpu_fp32 - exact|) per tensor for values / gradients; an optimizer step is checked as
arithmetic (float64 torch.optim fed OUR gradient must give our weights / moments up to fp32 rounding).  No GPU library (MIOpen / rocBLAS) is a comparator anywhere in this file."""
import copy

import numpy as np
import pytest

import fp64_truth as T
import margins

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north star: 1e-5 in fp32, relative to the tensor's largest entry


def _net64(name, *args, seed=0, **kw):
    """The reference module in float64 with fp32-representable, slightly de-symmetrised parameters, and its float32 twin."""
    import torch
    from mirror.networks import Network

    torch.manual_seed(seed)
    m = Network(name, *args, **kw).double()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    T.round_to_fp32_(m)
    return m, T.as32(m)


def _mk(head, state_size, A, K, H, B, seed=0, noise_type="factorized"):
    from jorldy_amd import ops

    ref64, ref32 = _net64("rainbow", state_size, A, K, noise_type, D_hidden=H, head=head, seed=seed)
    tgt64, tgt32 = _net64("rainbow", state_size, A, K, noise_type, D_hidden=H, head=head, seed=seed + 100)
    nat = ops.RainbowNet(state_size, A, K, H, head, B, "cuda:0", noise_type=noise_type)
    nat.import_state(ref32.state_dict(), nat.params)
    nat.import_state(tgt32.state_dict(), nat.target)
    return (ref64, ref32), (tgt64, tgt32), nat


def _inputs(head, S, rows, g):
    """-> (device tensor for the native net: uint8 frames / fp32 vectors, the same values as float64 on the CPU)."""
    import torch

    if head == "cnn":
        x = torch.randint(0, 256, (rows,) + tuple(S), dtype=torch.uint8, generator=g)
    else:
        x = torch.randn(rows, S, generator=g)
    return x.cuda(), x.double()


def _noise_dicts(nat, noise, dtype):
    """flat noise set -> the {tag: (e_in, e_out)} dict the mirror modules take."""
    H, NA = nat.H, nat.A * nat.K
    out = []
    for s in range(noise.shape[0]):
        e, o, d = noise[s].to(dtype), 0, {}
        for tag, n_out in (("a1", H), ("v1", H), ("a2", NA), ("v2", nat.K)):
            d[tag] = (e[o : o + H], e[o + H : o + H + n_out])
            o += H + n_out
        assert o == nat.noise_len
        out.append(d)
    return out


def _grads_vs_exact(nat, m64, m32, tol=TOL, tag=""):
    """-> our raw gradient {name: tensor} (checked against float64)."""
    grads = nat.export_state(nat.grads)
    p32 = dict(m32.named_parameters())
    for k, p in m64.named_parameters():
        T.vs_exact(grads[k], p.grad, p32[k].grad, tol, f"{tag}grad {k}")
    return grads


CASES = [
    ("mlp", 4, 3, 51, 32, 32),
    ("mlp", 6, 2, 11, 64, 5),           # S not a multiple of 4, ragged batch
    ("cnn", (4, 44, 52), 3, 51, 32, 8),  # non-square image, 2x3 feature map
    ("cnn", (4, 84, 84), 4, 51, 512, 32),  # config.rainbow.atari shapes
    ("cnn", (4, 48, 52), 2, 7, 32, 6),   # 11 x 12 = 132 conv1 pixels: not a multiple of the forward kernel's 16-pixel tiles
    ("cnn", (4, 44, 48), 2, 7, 32, 5),   # 10 x 11 = 110 conv1 pixels: not a multiple of 4 -> conv1 weight gradient on the tile engine
]


@pytest.mark.parametrize("head,S,A,K,H,B", CASES)
def test_rbnet_three_forwards_and_backward_match_float64(head, S, A, K, H, B):
    import torch

    (ref64, ref32), (tgt64, tgt32), nat = _mk(head, S, A, K, H, B)
    g = torch.Generator().manual_seed(1)
    x_dev, x64 = _inputs(head, S, 2 * B, g)
    noise = torch.randn(3, nat.noise_len, generator=g)
    out = torch.empty(3, B, A, K, device="cuda")
    noise_dev = noise.cuda()  # stays alive until after backward(): the sigma gradients re-read the draws
    nat.learn_forward(x_dev, B, noise_dev, out)
    nd64, nd32 = _noise_dicts(nat, noise, torch.float64), _noise_dicts(nat, noise, torch.float32)
    x32 = x64.float()
    l0, l0_32 = ref64(x64[:B], True, nd64[0]), ref32(x32[:B], True, nd32[0])
    with torch.no_grad():
        l1, l1_32 = ref64(x64[B:], True, nd64[1]), ref32(x32[B:], True, nd32[1])
        l2, l2_32 = tgt64(x64[B:], True, nd64[2]), tgt32(x32[B:], True, nd32[2])
    T.vs_exact(out[0], l0, l0_32, TOL, "online(state)")
    T.vs_exact(out[1], l1, l1_32, TOL, "online(next_state)")
    T.vs_exact(out[2], l2, l2_32, TOL, "target(next_state)")
    gl = torch.randn(B, A, K, generator=g) / B
    l0.backward(gl.double())
    l0_32.backward(gl)
    nat.backward(gl.cuda().contiguous())
    torch.cuda.synchronize()
    _grads_vs_exact(nat, ref64, ref32)


@pytest.mark.parametrize("head,S,A,K,H,B", CASES[:3])
def test_rbnet_eval_forward_and_uint8_vs_float_input(head, S, A, K, H, B):
    import torch

    (ref64, ref32), (tgt64, tgt32), nat = _mk(head, S, A, K, H, B, seed=3)
    g = torch.Generator().manual_seed(4)
    x_dev, x64 = _inputs(head, S, B, g)
    rows = max(1, B - 3)
    got = nat.forward(x_dev[:rows].contiguous(), which=1, noise=None)
    with torch.no_grad():
        want, want32 = tgt64(x64[:rows], False), tgt32(x64[:rows].float(), False)
    T.vs_exact(got, want, want32, TOL, "target eval forward")
    if head == "cnn":  # fp32 frames (the reference's as_tensor path) give the same numbers as uint8 frames: the operands are the
        # same fp32 values (byte / 255 correctly rounded), only the summation order of layer 1 differs (dedicated uint8 kernel)
        got_f = nat.forward(x_dev[:rows].float().contiguous(), which=1, noise=None)
        T.vs_exact(got_f, want, want32, TOL, "target eval forward from fp32 frames")


def _native_state(nat):
    return nat.export_state(), nat.export_state(nat.m), nat.export_state(nat.v)


def _force(nat, truth, set_hyper, it):
    params, m, v = truth.teacher_force()
    nat.import_state(params, nat.params)
    if m is not None:
        nat.import_state(m, nat.m)
    else:
        nat.m.zero_()
    if v is not None:
        nat.import_state(v, nat.v)
    else:
        nat.v.zero_()
    set_hyper(it)


def test_rbnet_adam_matches_float64_adam_over_several_steps():
    """Adam (eps 1.5e-4 as config.rainbow.atari), a learning-rate change in the middle; four teacher-forced steps."""
    import torch

    (ref64, ref32), _, nat = _mk("mlp", 4, 3, 11, 32, 16, seed=5)
    lr = [3e-4]
    truth = T.OptimTruth(ref64, ref32, lambda ps: torch.optim.Adam(ps, lr=lr[0], eps=1.5e-4), lr[0], ("exp_avg", "exp_avg_sq"))
    g = torch.Generator().manual_seed(2)
    for it in range(4):
        _force(nat, truth, lambda step: nat.set_hyper(lr[0], 0.9, 0.999, 1.5e-4, step), it)
        x_dev, x64 = _inputs("mlp", 4, 32, g)
        noise = torch.randn(3, nat.noise_len, generator=g)
        out = torch.empty(3, 16, 3, 11, device="cuda")
        noise_dev = noise.cuda()  # alive until after backward()
        nat.learn_forward(x_dev, 16, noise_dev, out)
        gl = torch.randn(16, 3, 11, generator=g)
        for m, opt, dt in ((ref64, truth.opt64, torch.float64), (ref32, truth.opt32, torch.float32)):
            opt.zero_grad()
            m(x64[:16].to(dt), True, _noise_dicts(nat, noise, dt)[0]).backward(gl.to(dt))
        nat.backward(gl.cuda())
        raw = _grads_vs_exact(nat, ref64, ref32, tag=f"step {it} ")
        nat.adam_step()
        truth.step(None, raw, *_native_state(nat), tag=f"adam step {it}")
        if it == 1:
            lr[0] = 1e-4
            truth.lr = 1e-4
            for opt in (truth.opt64, truth.opt32):
                opt.param_groups[0]["lr"] = 1e-4


def test_rbnet_state_dict_roundtrip_and_target_sync():
    import torch

    (ref64, ref32), _, nat = _mk("cnn", (4, 44, 52), 3, 51, 32, 4, seed=7)
    sd = nat.export_state()
    assert list(sd.keys()) == list(ref32.state_dict().keys())
    for k, v in ref32.state_dict().items():
        assert sd[k].shape == v.shape and torch.equal(sd[k].cpu(), v), k
    nat.sync_target()
    sdt = nat.export_state(nat.target)
    for k, v in ref32.state_dict().items():
        assert torch.equal(sdt[k].cpu(), v), k


# ------------------------------------------------------------------ dueling / q-network kinds (Ape-X, DQN family)
def _mk_kind(kind, head, S, A, H, B, seed=0):
    from jorldy_amd import ops

    name = {"dueling": "dueling", "q": "discrete_q_network"}[kind]
    ref64, ref32 = _net64(name, S, A, D_hidden=H, head=head, seed=seed)
    tgt64, tgt32 = _net64(name, S, A, D_hidden=H, head=head, seed=seed + 100)
    nat = ops.RainbowNet(S, A, 1, H, head, B, "cuda:0", kind=kind)
    nat.import_state(ref32.state_dict(), nat.params)
    nat.import_state(tgt32.state_dict(), nat.target)
    assert list(nat.export_state().keys()) == list(ref32.state_dict().keys())
    return (ref64, ref32), (tgt64, tgt32), nat


KIND_CASES = [
    ("q", "mlp", 4, 2, 32, 32),
    ("q", "cnn", (4, 44, 52), 6, 32, 8),
    ("dueling", "mlp", 6, 3, 64, 5),
    ("dueling", "cnn", (4, 44, 52), 6, 32, 8),
    ("dueling", "cnn", (4, 84, 84), 6, 512, 64),  # config.ape_x.atari shapes (Pong: A = 6), a slice of its batch
]


@pytest.mark.parametrize("kind,head,S,A,H,B", KIND_CASES)
def test_value_net_kinds_forward_backward_and_rmsprop_match_float64(kind, head, S, A, H, B):
    """Three teacher-forced steps of config.ape_x.atari's optimizer (centered RMSprop, alpha 0.95, eps 1.5e-7; clip_grad_norm_ 0.5
    on the first step so that the clip bites, 40 = the config's afterwards): forward values, every parameter gradient, grad_avg /
    square_avg and the stepped weights against float64."""
    import torch

    (ref64, ref32), (tgt64, tgt32), nat = _mk_kind(kind, head, S, A, H, B)
    lr = 2.5e-4
    truth = T.OptimTruth(ref64, ref32, lambda ps: torch.optim.RMSprop(ps, lr=lr, alpha=0.95, eps=1.5e-7, centered=True), lr, ("grad_avg", "square_avg"))
    g = torch.Generator().manual_seed(1)
    for it in range(3):
        _force(nat, truth, lambda step: nat.set_hyper(lr, 0.95, 0.0, 1.5e-7, step, centered=True), it)
        x_dev, x64 = _inputs(head, S, 2 * B, g)
        out = torch.empty(3, B, A, 1, device="cuda")
        nat.learn_forward(x_dev, B, None, out)
        x32 = x64.float()
        q0, q0_32 = ref64(x64[:B]), ref32(x32[:B])
        with torch.no_grad():
            q1, q1_32, q2, q2_32 = ref64(x64[B:]), ref32(x32[B:]), tgt64(x64[B:]), tgt32(x32[B:])
        T.vs_exact(out[0, :, :, 0], q0, q0_32, TOL, f"step {it} online(state)")
        T.vs_exact(out[1, :, :, 0], q1, q1_32, TOL, f"step {it} online(next_state)")
        T.vs_exact(out[2, :, :, 0], q2, q2_32, TOL, f"step {it} target(next_state)")
        gl = torch.randn(B, A, generator=g)
        truth.opt64.zero_grad()
        truth.opt32.zero_grad()
        q0.backward(gl.double())
        q0_32.backward(gl)
        nat.backward(gl.cuda().contiguous())
        raw = _grads_vs_exact(nat, ref64, ref32, tag=f"step {it} ")
        clip = 40.0 if it else 0.5  # 0.5: the clip bites
        nat.optim_step("rmsprop", clip)
        truth.step(clip, raw, *_native_state(nat), tag=f"rmsprop step {it}")


def test_value_net_adam_with_clip_and_eval_forward():
    import torch

    (ref64, ref32), _, nat = _mk_kind("dueling", "mlp", 5, 4, 32, 16, seed=9)
    lr = 1e-3
    truth = T.OptimTruth(ref64, ref32, lambda ps: torch.optim.Adam(ps, lr=lr), lr, ("exp_avg", "exp_avg_sq"))
    g = torch.Generator().manual_seed(2)
    for it in range(3):
        _force(nat, truth, lambda step: nat.set_hyper(lr, 0.9, 0.999, 1e-8, step), it)
        x_dev, x64 = _inputs("mlp", 5, 32, g)
        out = torch.empty(3, 16, 4, 1, device="cuda")
        nat.learn_forward(x_dev, 16, None, out)
        gl = torch.randn(16, 4, generator=g)
        truth.opt64.zero_grad()
        truth.opt32.zero_grad()
        ref64(x64[:16]).backward(gl.double())
        ref32(x64[:16].float()).backward(gl)
        nat.backward(gl.cuda())
        raw = _grads_vs_exact(nat, ref64, ref32, tag=f"step {it} ")
        nat.optim_step("adam", 1.0)
        truth.step(1.0, raw, *_native_state(nat), tag=f"adam+clip step {it}")
    x_dev, x64 = _inputs("mlp", 5, 7, g)
    T.round_to_fp32_(ref64)
    nat.import_state({k: v.float() for k, v in ref64.state_dict().items()}, nat.params)
    with torch.no_grad():
        T.vs_exact(nat.forward(x_dev, which=0)[:, :, 0], ref64(x64), T.as32(ref64)(x64.float()), TOL, "acting forward")


def test_rbnet_independent_noise_matches_float64():
    """noise_type="independent" (utils.py:72-79: one Gaussian draw per weight): three forwards + backward."""
    import torch

    S, A, K, H, B = 5, 3, 11, 32, 8
    (ref64, ref32), _, nat = _mk("mlp", S, A, K, H, B, noise_type="independent")
    nat.import_state(ref32.state_dict(), nat.target)
    g = torch.Generator().manual_seed(1)
    x_dev, x64 = _inputs("mlp", S, 2 * B, g)
    noise = torch.randn(3, nat.noise_len, generator=g)
    assert nat.noise_len == 2 * (H * H + H) + H * A * K + A * K + H * K + K

    def dicts(dtype):
        nd = []
        for s in range(3):
            e, o, d = noise[s].to(dtype), 0, {}
            for tag, n_out in (("a1", H), ("v1", H), ("a2", A * K), ("v2", K)):
                d[tag] = (e[o : o + H * n_out].view(H, n_out), e[o + H * n_out : o + H * n_out + n_out])
                o += H * n_out + n_out
            nd.append(d)
        return nd

    nd64, nd32 = dicts(torch.float64), dicts(torch.float32)
    out = torch.empty(3, B, A, K, device="cuda")
    noise_dev = noise.cuda()  # stays alive until after backward(): the sigma gradients re-read the draws
    nat.learn_forward(x_dev, B, noise_dev, out)
    x32 = x64.float()
    l0, l0_32 = ref64(x64[:B], True, nd64[0]), ref32(x32[:B], True, nd32[0])
    with torch.no_grad():
        l1, l1_32 = ref64(x64[B:], True, nd64[1]), ref32(x32[B:], True, nd32[1])
        l2, l2_32 = ref64(x64[B:], True, nd64[2]), ref32(x32[B:], True, nd32[2])
    T.vs_exact(out[0], l0, l0_32, TOL, "online(state)")
    T.vs_exact(out[1], l1, l1_32, TOL, "online(next_state)")
    T.vs_exact(out[2], l2, l2_32, TOL, "target(next_state)")
    gl = torch.randn(B, A, K, generator=g) / B
    l0.backward(gl.double())
    l0_32.backward(gl)
    nat.backward(gl.cuda().contiguous())
    _grads_vs_exact(nat, ref64, ref32)


@pytest.mark.parametrize("head,S,A,K,H,B", [("mlp", 4, 3, 51, 32, 32), ("cnn", (4, 84, 84), 6, 51, 64, 8), ("mlp", 6, 5, 21, 16, 33)])
def test_rainbow_fused_step_is_bit_identical_to_the_separate_calls(head, S, A, K, H, B):
    """jh_rbnet_c51_step (dueling combine + C51 + gradient through the combine in one launch; statistics + PER leaf write-back in one;
    climb) followed by the deferred backward (d(sigma) and conv1's partial sums folded into the optimizer pass) against the calls it
    replaces -- jh_rbnet_learn_heads, jh_c51_loss, jh_per_update, jh_rbnet_backward, jh_rbnet_optim_step: every output byte for byte
    (logits of the three forwards, priorities, KL, statistics, the whole sum tree with duplicate leaves in the batch, the gradient
    bucket, weights and both Adam moments after the step)."""
    import torch
    from jorldy_amd import ops

    g = torch.Generator().manual_seed(11)
    (r64, r32), _, nat_a = _mk(head, S, A, K, H, B, seed=5)
    _, _, nat_b = _mk(head, S, A, K, H, B, seed=5)
    x_dev, _ = _inputs(head, S, 2 * B, g)
    noise = torch.randn(3, nat_a.noise_len, generator=g).cuda()
    n_step, N = 3, 64
    action = torch.randint(0, A, (B, 1), generator=g).float().cuda()
    reward = torch.randn(B, n_step, generator=g).cuda()
    done = (torch.rand(B, n_step, generator=g) < 0.2).float().cuda()
    w = (torch.rand(B, generator=g) + 0.1).cuda()
    idx = (torch.randint(0, N, (B,), generator=g) + (N - 1)).cuda()  # tree space; B > distinct leaves in places: duplicates
    idx[1] = idx[0]
    trees = []
    for _ in range(2):
        t = ops.SumTree(N, 1e-3, device="cuda:0")
        t.push(N, (np.arange(N, dtype=np.float64) % 7 + 0.5))
        trees.append(t)
    out = {}
    for tag, nat, tree in (("sep", nat_a, trees[0]), ("fused", nat_b, trees[1])):
        nat.set_hyper(1e-3, 0.9, 0.999, 1e-8, 0)
        nat.m.zero_(); nat.v.zero_()
        logits = torch.empty(3, B, A, K, device="cuda")
        stats = torch.zeros(8, device="cuda")
        for it in range(2):  # two steps: the second runs on moments / weights the first produced
            nat.learn_trunk(x_dev, B)
            if tag == "sep":
                lg = nat.learn_heads(B, noise, logits)
                gl, prio, kl, _ = ops.c51_loss(lg[0], lg[2], action, reward, done, -2.0, 3.0, 0.99, next_logit_online=lg[1], weights=w, alpha=0.6, n_step=n_step, stats=stats)
                tree.update(idx, prio)
                nat.backward(gl)
            else:
                nat.learn_heads_raw(B, noise)
                prio, kl, _ = nat.c51_step(tree, idx, action, reward, done, w, -2.0, 3.0, 0.99, 0.6, n_step, logits, stats=stats)
                nat.backward(None, defer=True)
            nat.optim_step("adam", None)
        torch.cuda.synchronize()
        out[tag] = dict(logits=logits.cpu(), prio=prio.cpu(), kl=kl.cpu(), stats=stats.cpu()[:5], tree=torch.from_numpy(tree.dump()), maxp=torch.tensor(tree.state()["max_priority"]),
                        grads=nat.grads.cpu().clone(), params=nat.params.cpu().clone(), m=nat.m.cpu().clone(), v=nat.v.cpu().clone())
    for k in out["sep"]:
        a, b = out["sep"][k], out["fused"][k]
        assert torch.equal(a, b), (k, float((a.double() - b.double()).abs().max()))
    assert float(out["fused"]["grads"].abs().max()) > 0 and float(out["fused"]["prio"].max()) > 0


def test_rainbow_deferred_backward_with_clipping_and_flush():
    """The deferred tails run as their own launches when clipping needs the finished gradient first (jh_rbnet_optim_step, max_norm > 0)
    and in jh_rbnet_flush_grads (what a data-parallel learner calls in front of its all-reduce): same bytes as jh_rbnet_backward."""
    import torch
    from jorldy_amd import ops

    head, S, A, K, H, B = "cnn", (4, 84, 84), 4, 51, 32, 6
    g = torch.Generator().manual_seed(3)
    nets = [_mk(head, S, A, K, H, B, seed=9)[2] for _ in range(3)]
    x_dev, _ = _inputs(head, S, 2 * B, g)
    noise = torch.randn(3, nets[0].noise_len, generator=g).cuda()
    gl = (torch.randn(B, A, K, generator=g) * 0.05).cuda()
    res = []
    for mode, nat in zip(("plain", "deferred_clip", "deferred_flush"), nets):
        nat.set_hyper(1e-3, 0.9, 0.999, 1e-8, 0)
        nat.m.zero_(); nat.v.zero_()
        logits = torch.empty(3, B, A, K, device="cuda")
        nat.learn_trunk(x_dev, B)
        nat.learn_heads(B, noise, logits)
        nat.backward(gl, defer=mode != "plain")
        if mode == "deferred_flush":
            nat.flush_grads()
            g_mid = nat.grads.cpu().clone()
        nat.optim_step("adam", 0.5)
        torch.cuda.synchronize()
        res.append(dict(grads=nat.grads.cpu().clone(), params=nat.params.cpu().clone(), m=nat.m.cpu().clone(), v=nat.v.cpu().clone()))
    for r in res[1:]:
        for k in r:
            assert torch.equal(res[0][k], r[k]), k
    assert float(g_mid.abs().max()) > 0
