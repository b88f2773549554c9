"""Forward-capable restatement of the reference's hot-path networks -- TEST INFRASTRUCTURE, like oracle/: the comparator the
GPU parity tests evaluate on the CPU in float64 / float32 (tests/fp64_truth.py).  The product has no torch forward at all
(jorldy_amd/core/network holds parameter containers only); these classes add `forward` to those containers, so parameter names,
shapes and initialisation are by construction the ones the product ships.
  head.mlp / head.cnn          core/network/head.py:6-61
  discrete_q_network           core/network/q_network.py:8-20
  discrete_policy_value        core/network/policy_value.py:8-22       (+ raw(): pre-softmax logits, what the HIP loss kernels take)
  continuous_policy_value      core/network/policy_value.py:38-57      (+ raw(): pre-clamp mu, pre-tanh log_std)
  dueling                      core/network/dueling.py:8-35
  rainbow                      core/network/rainbow.py:8-94 (+ utils.py:55-86 noisy linear); `noise` injects the Gaussian draws
"""
import torch
import torch.nn.functional as F

from jorldy_amd.core import network as P


def encode(head, x):
    if isinstance(head, P.CNN):  # head.py:46 divides by 255 inside
        x = x / 255.0
        x = F.relu(head.conv1(x))
        x = F.relu(head.conv2(x))
        x = F.relu(head.conv3(x))
        return x.view(x.size(0), -1)
    return F.relu(head.l(x))


class DiscreteQ_Network(P.DiscreteQ_Network):
    def forward(self, x):
        return self.q(F.relu(self.l(encode(self.head, x))))


class DiscretePolicyValue(P.DiscretePolicyValue):
    def raw(self, x):
        x = F.relu(self.l(encode(self.head, x)))
        return self.pi(x), self.v(x)

    def forward(self, x):
        logits, v = self.raw(x)
        return torch.exp(F.log_softmax(logits, dim=-1)), v


class ContinuousPolicyValue(P.ContinuousPolicyValue):
    def raw(self, x):
        x = F.relu(self.l(encode(self.head, x)))
        return self.mu(x), self.log_std(x), self.v(x)

    def forward(self, x):
        mu, ls, v = self.raw(x)
        return torch.clamp(mu, min=-5.0, max=5.0), torch.tanh(ls).exp(), v


class Dueling(P.Dueling):
    def forward(self, x):
        x = encode(self.head, x)
        x_a = self.l2_a(F.relu(self.l1_a(x)))
        x_a = x_a - x_a.mean(dim=1, keepdim=True)
        return x_a + self.l2_v(F.relu(self.l1_v(x)))


class Rainbow(P.Rainbow):
    def _noisy(self, x, tag, is_train, noise):
        mu_w, sig_w = getattr(self, f"mu_w_{tag}"), getattr(self, f"sig_w_{tag}")
        mu_b, sig_b = getattr(self, f"mu_b_{tag}"), getattr(self, f"sig_b_{tag}")
        if not is_train:
            return torch.matmul(x, mu_w) + mu_b
        if self.noise_type == "factorized":
            if noise is not None:
                e_i, e_j = noise[tag]
            else:
                e_i = torch.randn(mu_w.size(0), device=x.device, dtype=x.dtype)
                e_j = torch.randn(mu_b.size(0), device=x.device, dtype=x.dtype)
            f_i = torch.sign(e_i) * torch.sqrt(torch.abs(e_i))
            f_j = torch.sign(e_j) * torch.sqrt(torch.abs(e_j))
            eps_w, eps_b = torch.outer(f_i, f_j), f_j
        else:
            if noise is not None:
                eps_w, eps_b = noise[tag]
            else:
                eps_w = torch.randn(mu_w.size(), device=x.device, dtype=x.dtype)
                eps_b = torch.randn(mu_b.size(), device=x.device, dtype=x.dtype)
        return torch.matmul(x, mu_w + sig_w * eps_w) + (mu_b + sig_b * eps_b)

    def forward(self, x, is_train, noise=None):
        x = F.relu(self.l(encode(self.head, x)))
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
    return network_dict[name.lower()](*args, **kwargs)
