"""Parameter containers of the hot-path networks: the reference's parameter names, shapes, registration order (= `state_dict`
layout, so checkpoints and `sync_in/out` payloads interchange) and initialisation -- and nothing else.  There is NO forward here:
every forward / backward / optimizer step of the product runs in libjorldy_hip (ops.PPONet, ops.RainbowNet); these modules only
provide the initial weights and, for PPO, the nn.Parameter views of the native flat bucket that `state_dict()` / the torch
checkpoint format need.  (The forward-capable restatement that the parity tests evaluate in float64 lives in tests/mirror.)
  head.mlp / head.cnn          core/network/head.py:6-61
  discrete_q_network           core/network/q_network.py:8-20
  discrete_policy_value        core/network/policy_value.py:8-22
  continuous_policy_value      core/network/policy_value.py:38-57
  dueling                      core/network/dueling.py:8-35
  rainbow                      core/network/rainbow.py:8-94 (+ utils.py:55-107 noisy linear)
"""
import torch


def orthogonal_init(layer, nonlinearity="relu"):
    """core/network/utils.py:110-124."""
    if isinstance(nonlinearity, str):
        gain = 0.01 if nonlinearity == "policy" else torch.nn.init.calculate_gain(nonlinearity)
    else:
        gain = nonlinearity
    for l in layer if isinstance(layer, list) else [layer]:
        torch.nn.init.orthogonal_(l.weight.data, gain)
        torch.nn.init.zeros_(l.bias.data)


class _Container(torch.nn.Module):
    def forward(self, *args, **kwargs):
        raise RuntimeError(f"{type(self).__name__} is a parameter container: the product computes in libjorldy_hip (ops.PPONet / ops.RainbowNet), "
                           "not in torch modules")


class MLP(_Container):
    def __init__(self, D_in, D_hidden=512):
        super().__init__()
        self.l = torch.nn.Linear(D_in, D_hidden)
        self.D_head_out = D_hidden
        orthogonal_init(self.l)


class CNN(_Container):
    """Nature-CNN head (head.py:21-61)."""

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


head_dict = {"mlp": MLP, "cnn": CNN}


class BaseNetwork(_Container):
    def __init__(self, D_in, D_hidden, head):
        super().__init__()
        assert head in head_dict, f"head {head!r} is outside the hot path (have {list(head_dict)})"
        self.head = head_dict[head](D_in, D_hidden)


class DiscreteQ_Network(BaseNetwork):
    def __init__(self, D_in, D_out, D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.l = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.q = torch.nn.Linear(D_hidden, D_out)
        orthogonal_init(self.l)
        orthogonal_init(self.q, "linear")


class DiscretePolicyValue(BaseNetwork):
    def __init__(self, D_in, D_out, D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.l = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.pi = torch.nn.Linear(D_hidden, D_out)
        self.v = torch.nn.Linear(D_hidden, 1)
        orthogonal_init(self.l)
        orthogonal_init(self.pi, "policy")
        orthogonal_init(self.v, "linear")


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


class Dueling(BaseNetwork):
    def __init__(self, D_in, D_out, D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.l1_a = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.l1_v = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.l2_a = torch.nn.Linear(D_hidden, D_out)
        self.l2_v = torch.nn.Linear(D_hidden, 1)
        orthogonal_init([self.l1_a, self.l1_v])
        orthogonal_init([self.l2_a, self.l2_v], "linear")


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
    """Dueling + noisy + categorical head (rainbow.py:8-94)."""

    def __init__(self, D_in, D_out, N_atom, noise_type="factorized", D_hidden=512, head="mlp"):
        super().__init__(D_in, D_hidden, head)
        self.D_out, self.N_atom, self.noise_type = D_out, N_atom, noise_type
        self.l = torch.nn.Linear(self.head.D_head_out, D_hidden)
        self.mu_w_a1, self.sig_w_a1, self.mu_b_a1, self.sig_b_a1 = _noisy_params((D_hidden, D_hidden), noise_type)
        self.mu_w_v1, self.sig_w_v1, self.mu_b_v1, self.sig_b_v1 = _noisy_params((D_hidden, D_hidden), noise_type)
        self.mu_w_a2, self.sig_w_a2, self.mu_b_a2, self.sig_b_a2 = _noisy_params((D_hidden, N_atom * D_out), noise_type)
        self.mu_w_v2, self.sig_w_v2, self.mu_b_v2, self.sig_b_v2 = _noisy_params((D_hidden, N_atom), noise_type)
        orthogonal_init(self.l)


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
