"""GPU parity tests of the native Rainbow network (jh_rbnet_*: implicit-GEMM convolutions, noisy
dueling heads, backward, Adam) against the plain PyTorch fp32 mirror of the reference modules
(jorldy_amd/core/network = core/network/rainbow.py:8-94, head.py:6-61) with the same parameters,
inputs and NoisyNet draws.  fp32 tolerance: 2e-5 relative to the tensor's max magnitude (the MFMA
accumulates K in a different order than rocBLAS / MIOpen)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(head, state_size, A, K, H, B, seed=0):
    import torch
    from jorldy_amd import ops
    from jorldy_amd.core.network import Network

    torch.manual_seed(seed)
    ref = Network("rainbow", state_size, A, K, "factorized", D_hidden=H, head=head).cuda()
    tgt = Network("rainbow", state_size, A, K, "factorized", D_hidden=H, head=head).cuda()
    with torch.no_grad():
        for net in (ref, tgt):
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
    nat = ops.RainbowNet(state_size, A, K, H, head, B, "cuda:0")
    nat.import_state(ref.state_dict(), nat.params)
    nat.import_state(tgt.state_dict(), nat.target)
    return ref, tgt, nat


def _noise_dicts(nat, noise):
    """flat noise set -> the {tag: (e_in, e_out)} dict the torch mirror takes."""
    H, NA, K = nat.H, nat.A * nat.K, nat.K
    out = []
    for s in range(noise.shape[0]):
        e, o, d = noise[s], 0, {}
        for tag, n_out in (("a1", H), ("v1", H), ("a2", NA), ("v2", K)):
            d[tag] = (e[o : o + H], e[o + H : o + H + n_out])
            o += H + n_out
        assert o == nat.noise_len
        out.append(d)
    return out


def _close(a, b, tol=2e-5, what=""):
    import torch

    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{what}: max err {err:.3e} relative to max |ref| {scale:.3e}"


CASES = [
    ("mlp", 4, 3, 51, 32, 32),
    ("mlp", 6, 2, 11, 64, 5),           # S not a multiple of 4, ragged batch
    ("cnn", (4, 44, 52), 3, 51, 32, 8),  # non-square image, 2x3 feature map
    ("cnn", (4, 84, 84), 4, 51, 512, 32),  # config.rainbow.atari shapes
    ("cnn", (4, 48, 52), 2, 7, 32, 6),   # 11 x 12 = 132 conv1 pixels: not a multiple of the forward kernel's 16-pixel tiles
    ("cnn", (4, 44, 48), 2, 7, 32, 5),   # 10 x 11 = 110 conv1 pixels: not a multiple of 4 -> conv1 weight gradient on the tile engine
]


@pytest.mark.parametrize("head,S,A,K,H,B", CASES)
def test_rbnet_three_forwards_and_backward_match_torch(head, S, A, K, H, B):
    import torch

    ref, tgt, nat = _mk(head, S, A, K, H, B)
    g = torch.Generator(device="cuda").manual_seed(1)
    if head == "cnn":
        x_all = torch.randint(0, 256, (2 * B,) + tuple(S), dtype=torch.uint8, device="cuda", generator=g)
    else:
        x_all = torch.randn(2 * B, S, device="cuda", generator=g)
    noise = torch.randn(3, nat.noise_len, device="cuda", generator=g)
    out = torch.empty(3, B, A, K, device="cuda")
    nat.learn_forward(x_all, B, noise, out)
    nd = _noise_dicts(nat, noise)
    xf = x_all.float()
    l0 = ref(xf[:B], True, nd[0])
    with torch.no_grad():
        l1 = ref(xf[B:], True, nd[1])
        l2 = tgt(xf[B:], True, nd[2])
    _close(out[0], l0.detach(), what="online(state)")
    _close(out[1], l1, what="online(next_state)")
    _close(out[2], l2, what="target(next_state)")
    gl = torch.randn(B, A, K, device="cuda", generator=g) / B
    ref.zero_grad()
    l0.backward(gl)
    nat.backward(gl.contiguous())
    torch.cuda.synchronize()
    grads = nat.export_state(nat.grads)
    for k, p in ref.named_parameters():
        _close(grads[k], p.grad, tol=5e-5, what=f"grad {k}")


@pytest.mark.parametrize("head,S,A,K,H,B", CASES[:3])
def test_rbnet_eval_forward_and_uint8_vs_float_input(head, S, A, K, H, B):
    import torch

    ref, tgt, nat = _mk(head, S, A, K, H, B, seed=3)
    if head == "cnn":
        x = torch.randint(0, 256, (B,) + tuple(S), dtype=torch.uint8, device="cuda")
    else:
        x = torch.randn(B, S, device="cuda")
    rows = max(1, B - 3)
    got = nat.forward(x[:rows].contiguous(), which=1, noise=None)
    with torch.no_grad():
        want = tgt(x[:rows].float(), False)
    _close(got, want, what="target eval forward")
    if head == "cnn":  # fp32 frames (the reference's as_tensor path) give the same numbers as uint8 frames: the operands are the
        # same fp32 values (byte / 255 correctly rounded), only the summation order of layer 1 differs (dedicated uint8 kernel)
        got_f = nat.forward(x[:rows].float().contiguous(), which=1, noise=None)
        _close(got, got_f, tol=2e-6, what="uint8 vs fp32 frames")


def test_rbnet_adam_matches_torch_adam_over_several_steps():
    import torch

    ref, tgt, nat = _mk("mlp", 4, 3, 11, 32, 16, seed=5)
    opt = torch.optim.Adam(ref.parameters(), lr=3e-4, eps=1.5e-4)
    nat.set_hyper(3e-4, 0.9, 0.999, 1.5e-4, 0)
    g = torch.Generator(device="cuda").manual_seed(2)
    for it in range(4):
        x_all = torch.randn(32, 4, device="cuda", generator=g)
        noise = torch.randn(3, nat.noise_len, device="cuda", generator=g)
        out = torch.empty(3, 16, 3, 11, device="cuda")
        nat.learn_forward(x_all, 16, noise, out)
        gl = torch.randn(16, 3, 11, device="cuda", generator=g)
        l0 = ref(x_all[:16], True, _noise_dicts(nat, noise)[0])
        opt.zero_grad()
        l0.backward(gl)
        opt.step()
        nat.backward(gl)
        nat.adam_step()
        if it == 1:
            nat.set_lr(1e-4)
            opt.param_groups[0]["lr"] = 1e-4
    sd = nat.export_state()
    for k, p in ref.state_dict().items():
        assert float((sd[k] - p).abs().max()) <= 2e-6, k  # a handful of lr-sized steps


def test_rbnet_state_dict_roundtrip_and_target_sync():
    import torch

    ref, tgt, nat = _mk("cnn", (4, 44, 52), 3, 51, 32, 4, seed=7)
    sd = nat.export_state()
    assert list(sd.keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert sd[k].shape == v.shape and torch.equal(sd[k], v), k
    nat.sync_target()
    sdt = nat.export_state(nat.target)
    for k, v in ref.state_dict().items():
        assert torch.equal(sdt[k], v), k


# ------------------------------------------------------------------ dueling / q-network kinds (Ape-X, DQN family)
def _mk_kind(kind, head, S, A, H, B, seed=0):
    import torch
    from jorldy_amd import ops
    from jorldy_amd.core.network import Network

    torch.manual_seed(seed)
    name = {"dueling": "dueling", "q": "discrete_q_network"}[kind]
    ref = Network(name, S, A, D_hidden=H, head=head).cuda()
    tgt = Network(name, S, A, D_hidden=H, head=head).cuda()
    with torch.no_grad():
        for net in (ref, tgt):
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
    nat = ops.RainbowNet(S, A, 1, H, head, B, "cuda:0", kind=kind)
    nat.import_state(ref.state_dict(), nat.params)
    nat.import_state(tgt.state_dict(), nat.target)
    assert list(nat.export_state().keys()) == list(ref.state_dict().keys())
    return ref, tgt, nat


KIND_CASES = [
    ("q", "mlp", 4, 2, 32, 32),
    ("q", "cnn", (4, 44, 52), 6, 32, 8),
    ("dueling", "mlp", 6, 3, 64, 5),
    ("dueling", "cnn", (4, 44, 52), 6, 32, 8),
    ("dueling", "cnn", (4, 84, 84), 6, 512, 64),  # config.ape_x.atari shapes (Pong: A = 6), a slice of its batch
]


@pytest.mark.parametrize("kind,head,S,A,H,B", KIND_CASES)
def test_value_net_kinds_forward_backward_and_rmsprop_match_torch(kind, head, S, A, H, B):
    import torch

    ref, tgt, nat = _mk_kind(kind, head, S, A, H, B)
    opt = torch.optim.RMSprop(ref.parameters(), lr=2.5e-4, alpha=0.95, eps=1.5e-7, centered=True)
    nat.set_hyper(2.5e-4, 0.95, 0.0, 1.5e-7, 0, centered=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    for it in range(3):
        if head == "cnn":
            x_all = torch.randint(0, 256, (2 * B,) + tuple(S), dtype=torch.uint8, device="cuda", generator=g)
        else:
            x_all = torch.randn(2 * B, S, device="cuda", generator=g)
        out = torch.empty(3, B, A, 1, device="cuda")
        nat.learn_forward(x_all, B, None, out)
        xf = x_all.float()
        q0 = ref(xf[:B])
        with torch.no_grad():
            q1, q2 = ref(xf[B:]), tgt(xf[B:])
        if it == 0:
            _close(out[0, :, :, 0], q0.detach(), what="online(state)")
            _close(out[1, :, :, 0], q1, what="online(next_state)")
            _close(out[2, :, :, 0], q2, what="target(next_state)")
        gl = torch.randn(B, A, device="cuda", generator=g)
        opt.zero_grad()
        q0.backward(gl)
        nat.backward(gl.contiguous())
        if it == 0:
            grads = nat.export_state(nat.grads)
            for k, p in ref.named_parameters():
                _close(grads[k], p.grad, tol=5e-5, what=f"grad {k}")
        norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 40.0 if it else 0.5)  # 0.5: the clip bites
        opt.step()
        nat.optim_step("rmsprop", 40.0 if it else 0.5)
    sd = nat.export_state()
    for k, p in ref.state_dict().items():
        # centered RMSprop divides by sqrt(E[g^2] - E[g]^2): ill-conditioned in the first steps, so compare the
        # travel of every weight (<= a few lr) rather than demanding bit-close updates
        assert float((sd[k] - p).abs().max()) <= 3e-5, k


def test_value_net_adam_with_clip_and_eval_forward():
    import torch

    ref, tgt, nat = _mk_kind("dueling", "mlp", 5, 4, 32, 16, seed=9)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    nat.set_hyper(1e-3, 0.9, 0.999, 1e-8, 0)
    g = torch.Generator(device="cuda").manual_seed(2)
    for it in range(3):
        x_all = torch.randn(32, 5, device="cuda", generator=g)
        out = torch.empty(3, 16, 4, 1, device="cuda")
        nat.learn_forward(x_all, 16, None, out)
        gl = torch.randn(16, 4, device="cuda", generator=g)
        opt.zero_grad()
        ref(x_all[:16]).backward(gl)
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
        nat.backward(gl)
        nat.optim_step("adam", 1.0)
    sd = nat.export_state()
    for k, p in ref.state_dict().items():
        assert float((sd[k] - p).abs().max()) <= 3e-6, k
    x = torch.randn(7, 5, device="cuda", generator=g)
    _close(nat.forward(x, which=0)[:, :, 0], ref(x).detach(), what="acting forward")


@pytest.mark.gpu
@pytest.mark.parametrize("n,M,N,K", [(3, 64, 1024, 3136), (2, 512, 512, 3136), (1, 2048, 256, 64), (4, 96, 64, 4096)])
def test_tgemm_grouped_split_k_is_exact_launch_after_launch(n, M, N, K):
    """Grouped launches on the LDS-DMA operand path with split-K hand-offs (the Ape-X / R2D2 forward shapes), many times over fresh
    operands: the hand-off between the splits (sc1 partial stores, ticket, last arriver's sum) went wrong once in a few hundred
    launches -- 32 elements of one accumulator fragment -- until its asm loads carried their wait and its asm stores their s_nop."""
    import torch
    from jorldy_amd import ops

    g = torch.Generator(device="cuda").manual_seed(n * 1000 + M)
    for _ in range(40):
        As = [torch.randn(M, K, device="cuda", generator=g) for _ in range(n)]
        Bs = [torch.randn(N, K, device="cuda", generator=g) for _ in range(n)]
        Cs = ops.tgemm_dense_group(As, Bs)
        torch.cuda.synchronize()
        for a, b, c in zip(As, Bs, Cs):
            want = a.double() @ b.double().t()
            err = ((c.double() - want).abs() / want.abs().max())
            assert int((~(err < 1e-5)).sum()) == 0, float(err.max())


def test_tgemm_dense_random_shapes_modes_and_epilogues_match_torch():
    """The GEMM engine under every value-network layer, on 80 random problems: ragged M / N / K (not multiples of
    the 64 x 64 x 32 tile, of 4, or of anything), all four dense operand layouts, row strides that do and do not
    allow 16-byte loads, every epilogue, fused row sums, shapes that do and do not split K."""
    import torch
    from jorldy_amd import ops

    g = torch.Generator(device="cuda").manual_seed(0)
    rng = np.random.RandomState(0)
    dims = [1, 2, 3, 4, 5, 7, 8, 11, 16, 31, 32, 33, 51, 64, 65, 100, 127, 128, 204, 256, 512, 777, 1024, 3136]
    for case in range(80):
        M, N = int(rng.choice(dims[:-3])), int(rng.choice(dims[:-3]))
        K = int(rng.choice(dims)) if case % 5 else int(rng.choice([2048, 3136, 12800]))
        a_kc, b_kc = bool(rng.randint(2)), bool(rng.randint(2))
        pad_a, pad_b = int(rng.choice([0, 0, 1, 4])), int(rng.choice([0, 0, 3, 4]))
        A = torch.randn((M, K + pad_a) if a_kc else (K, M + pad_a), device="cuda", generator=g)
        Bm = torch.randn((N, K + pad_b) if b_kc else (K, N + pad_b), device="cuda", generator=g)
        a_v = A[:, :K] if a_kc else A[:, :M]
        b_v = Bm[:, :K] if b_kc else Bm[:, :N]
        a2 = a_v if a_kc else a_v.t()       # [M, K]
        b2 = b_v.t() if b_kc else b_v       # [K, N]
        epi = case % 4
        bias = torch.randn(N, device="cuda", generator=g) if epi in (1, 2) else None
        aux = torch.randn(M, N, device="cuda", generator=g) if epi == 3 else None
        want = a2.double() @ b2.double()
        if epi in (1, 2):
            want = want + bias.double()
        if epi == 2:
            want = want.clamp_min(0)
        if epi == 3:
            want = torch.where(aux > 0, want, torch.zeros_like(want))
        got, rs = ops.tgemm_dense(a_v, b_v, a_kcont=a_kc, b_kcont=b_kc, epi=epi, bias=bias, aux=aux, rowsum=True, M=M, N=N, K=K)
        scale = float(a2.abs().double().matmul(b2.abs().double()).max()) + 1e-9  # fp32 accumulation error scales with sum |a||b|
        err = float((got.double() - want).abs().max()) / scale
        assert err < 2e-6, (case, M, N, K, a_kc, b_kc, epi, err)
        rs_err = float((rs.double() - a2.double().sum(1)).abs().max()) / (float(a2.abs().double().sum(1).max()) + 1e-9)
        assert rs_err < 2e-6, (case, M, N, K, "rowsum", rs_err)


@pytest.mark.gpu
def test_tgemm_dense_lds_dma_shapes_all_layouts_match_torch():
    """Problems the LDS-DMA kernel takes (K % 32 == 0, 16-byte pieces, x-contiguous extents % 4 == 0; tiles that are and are not
    full, K ranges that do and do not split, one to three chunk buffers' worth of K) in all four dense layouts, with every epilogue
    and the fused row sums."""
    import torch
    from jorldy_amd import ops

    g = torch.Generator(device="cuda").manual_seed(1)
    rng = np.random.RandomState(1)
    for case in range(64):
        a_kc, b_kc = bool(case & 1), bool(case & 2)
        M = int(rng.choice([64, 65, 100, 512, 777] if a_kc else [64, 68, 132, 512, 1000]))
        N = int(rng.choice([64, 100, 129, 512] if b_kc else [64, 68, 260, 512]))
        K = int(rng.choice([32, 64, 96, 128, 512, 1024, 3136, 6400]))
        pad = 4 * int(rng.randint(2))
        A = torch.randn((M, K + pad) if a_kc else (K, M + pad), device="cuda", generator=g)
        Bm = torch.randn((N, K + pad) if b_kc else (K, N + pad), device="cuda", generator=g)
        a_v = A[:, :K] if a_kc else A[:, :M]
        b_v = Bm[:, :K] if b_kc else Bm[:, :N]
        a2 = a_v if a_kc else a_v.t()
        b2 = b_v.t() if b_kc else b_v
        epi = (case >> 2) % 4
        bias = torch.randn(N, device="cuda", generator=g) if epi in (1, 2) else None
        aux = torch.randn(M, N, device="cuda", generator=g) if epi == 3 else None
        want = a2.double() @ b2.double()
        if epi in (1, 2):
            want = want + bias.double()
        if epi == 2:
            want = want.clamp_min(0)
        if epi == 3:
            want = torch.where(aux > 0, want, torch.zeros_like(want))
        for rep in range(3):
            got, rs = ops.tgemm_dense(a_v, b_v, a_kcont=a_kc, b_kcont=b_kc, epi=epi, bias=bias, aux=aux, rowsum=True, M=M, N=N, K=K)
            scale = float(a2.abs().double().matmul(b2.abs().double()).max()) + 1e-9
            err = float((got.double() - want).abs().max()) / scale
            assert err < 2e-6, (case, rep, M, N, K, a_kc, b_kc, epi, err)
            rs_err = float((rs.double() - a2.double().sum(1)).abs().max()) / (float(a2.abs().double().sum(1).max()) + 1e-9)
            assert rs_err < 2e-6, (case, rep, M, N, K, "rowsum", rs_err)


def test_rbnet_independent_noise_matches_torch():
    """noise_type="independent" (utils.py:72-79: one Gaussian draw per weight): three forwards + backward."""
    import torch
    from jorldy_amd import ops
    from jorldy_amd.core.network import Network

    S, A, K, H, B = 5, 3, 11, 32, 8
    torch.manual_seed(0)
    ref = Network("rainbow", S, A, K, "independent", D_hidden=H, head="mlp").cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    nat = ops.RainbowNet(S, A, K, H, "mlp", B, "cuda:0", noise_type="independent")
    nat.import_state(ref.state_dict(), nat.params)
    nat.import_state(ref.state_dict(), nat.target)
    g = torch.Generator(device="cuda").manual_seed(1)
    x_all = torch.randn(2 * B, S, device="cuda", generator=g)
    noise = torch.randn(3, nat.noise_len, device="cuda", generator=g)
    assert nat.noise_len == 2 * (H * H + H) + H * A * K + A * K + H * K + K
    nd = []
    for s in range(3):
        e, o, d = noise[s], 0, {}
        for tag, n_out in (("a1", H), ("v1", H), ("a2", A * K), ("v2", K)):
            d[tag] = (e[o : o + H * n_out].view(H, n_out), e[o + H * n_out : o + H * n_out + n_out])
            o += H * n_out + n_out
        nd.append(d)
    out = torch.empty(3, B, A, K, device="cuda")
    nat.learn_forward(x_all, B, noise, out)
    l0 = ref(x_all[:B], True, nd[0])
    with torch.no_grad():
        l1, l2 = ref(x_all[B:], True, nd[1]), ref(x_all[B:], True, nd[2])
    _close(out[0], l0.detach(), what="online(state)")
    _close(out[1], l1, what="online(next_state)")
    _close(out[2], l2, what="target(next_state)")
    gl = torch.randn(B, A, K, device="cuda", generator=g) / B
    ref.zero_grad()
    l0.backward(gl)
    nat.backward(gl.contiguous())
    grads = nat.export_state(nat.grads)
    for k, p in ref.named_parameters():
        _close(grads[k], p.grad, tol=5e-5, what=f"grad {k}")
