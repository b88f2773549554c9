"""Encoders used by the hot-path agents.  Parameter names / shapes (the `state_dict` layout) are
identical to the reference's networks so checkpoints and `sync_in/out` payloads interchange:
  head.mlp / head.cnn          core/network/head.py:6-61
  discrete_q_network           core/network/q_network.py:8-20
  discrete_policy_value        core/network/policy_value.py:8-22
  continuous_policy_value      core/network/policy_value.py:38-57
  dueling                      core/network/dueling.py:8-35
  rainbow                      core/network/rainbow.py:8-94 (+ utils.py:55-86 noisy linear)
Unlike the reference the policy networks expose `raw(x)`: the pre-softmax logits / pre-clamp mu /
pre-tanh log_std, because the HIP loss kernels fuse those head transforms (and their backward).
"""
import torch
import torch.nn.functional as F


def orthogonal_init(layer, nonlinearity="relu"):
    """core/network/utils.py:110-124."""
    if isinstance(nonlinearity, str):
        gain = 0.01 if nonlinearity == "policy" else torch.nn.init.calculate_gain(nonlinearity)
    else:
        gain = nonlinearity
    for l in layer if isinstance(layer, list) else [layer]:
        torch.nn.init.orthogonal_(l.weight.data, gain)
        torch.nn.init.zeros_(l.bias.data)


class MLP(torch.nn.Module):
    def __init__(self, D_in, D_hidden=512):
        super().__init__()
        self.l = torch.nn.Linear(D_in, D_hidden)
        self.D_head_out = D_hidden
        orthogonal_init(self.l)

    def forward(self, x):
        return F.relu(self.l(x))


class CNN(torch.nn.Module):
    """Nature-CNN head; divides by 255 inside (head.py:46), so uint8 frames can stay uint8 in HBM
    until here."""

    def __init__(self, D_in, D_hidden=512):
        super().__init__()
        assert D_in[1] >= 36 and D_in[2] >= 36
        self.conv1 = torch.nn.Conv2d(D_in[0], 32, kernel_size=8, stride=4)
        d1 = ((D_in[1] - 8) // 4 + 1, (D_in[2] - 8) // 4 + 1)
        self.conv2 = torch.nn.Conv2d(32, 64, kernel_size=4, stride=2)
        d2 = ((d1[0] - 4) // 2 + 1, (d1[1] - 4) // 2 + 1)
        self.conv3 = torch.nn.Conv2d(64, 64, kernel_size=3, stride=1)
        d3 = (d2[0] - 3 + 1, d2[1] - 3 + 1)
        self.D_head_out = 64 * d3[0] * d3[1]
        for layer in (self.conv1, self.conv2, self.conv3):
            orthogonal_init(layer)

    def forward(self, x):
        x = x / 255.0
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        return x.view(x.size(0), -1)


head_dict = {"mlp": MLP, "cnn": CNN}


class BaseNetwork(torch.nn.Module):
    def __init__(self, D_in, D_hidden, head):
        super().__init__()
        assert head in head_dict, f"head {head!r} is outside the hot path (have {list(head_dict)})"
        self.head = head_dict[head](D_in, D_hidden)

    def encode(self, x):
        return self.head(x)


class DiscreteQ_Network(BaseNetwork):
    def __init__(self, D_in, D_out, D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.l = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.q = torch.nn.Linear(D_hidden, D_out)
        orthogonal_init(self.l)
        orthogonal_init(self.q, "linear")

    def forward(self, x):
        return self.q(F.relu(self.l(self.encode(x))))


class DiscretePolicyValue(BaseNetwork):
    def __init__(self, D_in, D_out, D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.l = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.pi = torch.nn.Linear(D_hidden, D_out)
        self.v = torch.nn.Linear(D_hidden, 1)
        orthogonal_init(self.l)
        orthogonal_init(self.pi, "policy")
        orthogonal_init(self.v, "linear")

    def raw(self, x):
        """(logits, value): pre-softmax policy head."""
        x = F.relu(self.l(self.encode(x)))
        return self.pi(x), self.v(x)

    def forward(self, x):
        logits, v = self.raw(x)
        return torch.exp(F.log_softmax(logits, dim=-1)), v


class ContinuousPolicyValue(BaseNetwork):
    def __init__(self, D_in, D_out, D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.l = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.mu = torch.nn.Linear(D_hidden, D_out)
        self.log_std = torch.nn.Linear(D_hidden, D_out)
        self.v = torch.nn.Linear(D_hidden, 1)
        orthogonal_init(self.l)
        orthogonal_init(self.mu, "linear")
        orthogonal_init(self.log_std, "tanh")
        orthogonal_init(self.v, "linear")

    def raw(self, x):
        """(mu_raw, log_std_raw, value): before clamp(+-5) / tanh."""
        x = F.relu(self.l(self.encode(x)))
        return self.mu(x), self.log_std(x), self.v(x)

    def forward(self, x):
        mu, ls, v = self.raw(x)
        return torch.clamp(mu, min=-5.0, max=5.0), torch.tanh(ls).exp(), v


class Dueling(BaseNetwork):
    def __init__(self, D_in, D_out, D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.l1_a = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.l1_v = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.l2_a = torch.nn.Linear(D_hidden, D_out)
        self.l2_v = torch.nn.Linear(D_hidden, 1)
        orthogonal_init([self.l1_a, self.l1_v])
        orthogonal_init([self.l2_a, self.l2_v], "linear")

    def forward(self, x):
        x = self.encode(x)
        x_a = self.l2_a(F.relu(self.l1_a(x)))
        x_a = x_a - x_a.mean(dim=1, keepdim=True)
        return x_a + self.l2_v(F.relu(self.l1_v(x)))


def _noisy_params(shape, noise_type):
    """core/network/utils.py:89-107."""
    if noise_type == "factorized":
        mu_init, sig_init = 1.0 / (shape[0] ** 0.5), 0.5 / (shape[0] ** 0.5)
    else:
        mu_init, sig_init = (3.0 / shape[0]) ** 0.5, 0.017
    mu_w = torch.nn.Parameter(torch.empty(shape).uniform_(-mu_init, mu_init))
    sig_w = torch.nn.Parameter(torch.full(shape, sig_init))
    mu_b = torch.nn.Parameter(torch.empty(shape[1]).uniform_(-mu_init, mu_init))
    sig_b = torch.nn.Parameter(torch.full((shape[1],), sig_init))
    return mu_w, sig_w, mu_b, sig_b


class Rainbow(BaseNetwork):
    """Dueling + noisy + categorical head.  `noise` lets a caller inject the Gaussian draws (parity
    tests); by default they are drawn ON DEVICE (the reference draws on the CPU and copies)."""

    def __init__(self, D_in, D_out, N_atom, noise_type="factorized", D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.D_out, self.N_atom, self.noise_type = D_out, N_atom, noise_type
        self.l = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.mu_w_a1, self.sig_w_a1, self.mu_b_a1, self.sig_b_a1 = _noisy_params((D_hidden, D_hidden), noise_type)
        self.mu_w_v1, self.sig_w_v1, self.mu_b_v1, self.sig_b_v1 = _noisy_params((D_hidden, D_hidden), noise_type)
        self.mu_w_a2, self.sig_w_a2, self.mu_b_a2, self.sig_b_a2 = _noisy_params((D_hidden, N_atom * D_out), noise_type)
        self.mu_w_v2, self.sig_w_v2, self.mu_b_v2, self.sig_b_v2 = _noisy_params((D_hidden, N_atom), noise_type)
        orthogonal_init(self.l)

    def _noisy(self, x, tag, is_train, noise):
        mu_w, sig_w = getattr(self, f"mu_w_{tag}"), getattr(self, f"sig_w_{tag}")
        mu_b, sig_b = getattr(self, f"mu_b_{tag}"), getattr(self, f"sig_b_{tag}")
        if not is_train:
            return torch.matmul(x, mu_w) + mu_b
        if self.noise_type == "factorized":
            if noise is not None:
                e_i, e_j = noise[tag]
            else:
                e_i = torch.randn(mu_w.size(0), device=x.device)
                e_j = torch.randn(mu_b.size(0), device=x.device)
            f_i = torch.sign(e_i) * torch.sqrt(torch.abs(e_i))
            f_j = torch.sign(e_j) * torch.sqrt(torch.abs(e_j))
            eps_w, eps_b = torch.outer(f_i, f_j), f_j
        else:
            if noise is not None:
                eps_w, eps_b = noise[tag]
            else:
                eps_w = torch.randn(mu_w.size(), device=x.device)
                eps_b = torch.randn(mu_b.size(), device=x.device)
        return torch.matmul(x, mu_w + sig_w * eps_w) + (mu_b + sig_b * eps_b)

    def forward(self, x, is_train, noise=None):
        x = F.relu(self.l(self.encode(x)))
        x_a = F.relu(self._noisy(x, "a1", is_train, noise))
        x_v = F.relu(self._noisy(x, "v1", is_train, noise))
        x_a = self._noisy(x_a, "a2", is_train, noise).reshape(-1, self.D_out, self.N_atom)
        x_a = x_a - x_a.mean(dim=1, keepdim=True)
        x_v = self._noisy(x_v, "v2", is_train, noise).reshape(-1, 1, self.N_atom)
        return x_a + x_v  # [B, A, K]


network_dict = {
    "discrete_q_network": DiscreteQ_Network,
    "discrete_policy_value": DiscretePolicyValue,
    "continuous_policy_value": ContinuousPolicyValue,
    "dueling": Dueling,
    "rainbow": Rainbow,
}


def Network(name, *args, **kwargs):
    """core/network/__init__.py:30-41 factory (hot-path networks only)."""
    if not isinstance(name, str):
        raise Exception("### name variable must be string! ###")
    key = name.lower()
    if key not in network_dict:
        raise Exception(f"### can use only follows {list(network_dict.keys())}")
    return network_dict[key](*args, **kwargs)
