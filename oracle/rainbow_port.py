"""CPU port of the reference's Rainbow learner -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

What one `Rainbow.learn()` of `config.rainbow.atari` does on the reference's CPU path, restated with
torch-CPU ops in the reference's tensor-op style (dense one-hot projection, three network passes
with fresh factorised noise, per-sample priority write-back through the Python sum tree):
    network      core/network/rainbow.py:8-94 + head.py:21-61 (Nature CNN) + utils.py:55-107
    learn        core/agent/rainbow.py:154-253
    act / interact_callback / process   core/agent/rainbow.py:140-152, 294-308, 255-283 (the single-mode loop of run_mode.py:68-80)
    PER          core/buffer/per_buffer.py:19-101  (oracle.jorldy_oracle.PEROracle)
Used by bench.py to time the reference's learner on the bench box's host cores (`rainbow.cpu_reference`)
and pinned against the reference's own run in tests/test_oracle_golden.py::test_rainbow_port_matches_reference.
Never imported by jorldy_amd.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .jorldy_oracle import PEROracle


def _noisy(x, mw, sw, mb, sb, noise):
    """utils.py:55-83, factorised, training mode; `noise` = (eps_in, eps_out) or None to draw."""
    if noise is None:
        e_i, e_j = torch.randn(mw.shape[0]), torch.randn(mb.shape[0])
    else:
        e_i, e_j = noise
    f_i, f_j = e_i.sign() * e_i.abs().sqrt(), e_j.sign() * e_j.abs().sqrt()
    return x @ (mw + sw * torch.outer(f_i, f_j)) + (mb + sb * f_j)


class RainbowNet(torch.nn.Module):
    """Parameter names and registration order of the reference module (checkpoint-compatible)."""

    def __init__(self, state_size, A, K, H=512):
        super().__init__()
        self.A, self.K = A, K
        self.head = torch.nn.Module()
        if isinstance(state_size, (int, np.integer)):
            self.cnn = False
            self.head.l = torch.nn.Linear(state_size, H)
            feat = H
        else:
            self.cnn = True
            c, h, w = state_size
            self.head.conv1 = torch.nn.Conv2d(c, 32, 8, 4)
            self.head.conv2 = torch.nn.Conv2d(32, 64, 4, 2)
            self.head.conv3 = torch.nn.Conv2d(64, 64, 3, 1)
            d = lambda n, k, s: (n - k) // s + 1
            feat = 64 * d(d(d(h, 8, 4), 4, 2), 3, 1) * d(d(d(w, 8, 4), 4, 2), 3, 1)
        self.l = torch.nn.Linear(feat, H)
        # head.py:14,43 / rainbow.py:33: orthogonal_init(layer) = orthogonal weights with gain sqrt(2), zero biases (utils.py:110-124) --
        # what a run FROM SCRATCH starts from (the learn() fixtures load the reference's weights; the learning-curve tests do not)
        gain = torch.nn.init.calculate_gain("relu")
        for layer in list(self.head.children()) + [self.l]:
            torch.nn.init.orthogonal_(layer.weight.data, gain)
            torch.nn.init.zeros_(layer.bias.data)
        for tag, shape in (("a1", (H, H)), ("v1", (H, H)), ("a2", (H, K * A)), ("v2", (H, K))):
            bound = 1.0 / shape[0] ** 0.5
            for nm, val in (("mu_w", torch.empty(shape).uniform_(-bound, bound)), ("sig_w", torch.full(shape, 0.5 * bound)),
                            ("mu_b", torch.empty(shape[1]).uniform_(-bound, bound)), ("sig_b", torch.full((shape[1],), 0.5 * bound))):
                setattr(self, f"{nm}_{tag}", torch.nn.Parameter(val))

    def forward(self, x, noise=None):
        nz = noise or {}
        if self.cnn:
            x = x / 255.0
            x = F.relu(self.head.conv3(F.relu(self.head.conv2(F.relu(self.head.conv1(x)))))).flatten(1)
        else:
            x = F.relu(self.head.l(x))
        x = F.relu(self.l(x))
        lay = lambda t, tag: _noisy(t, getattr(self, "mu_w_" + tag), getattr(self, "sig_w_" + tag), getattr(self, "mu_b_" + tag),
                                    getattr(self, "sig_b_" + tag), nz.get(tag))
        xa = F.relu(lay(x, "a1"))  # draw order a1, v1, a2, v2 (rainbow.py:41-80)
        xv = F.relu(lay(x, "v1"))
        adv = lay(xa, "a2").reshape(-1, self.A, self.K)
        val = lay(xv, "v2").reshape(-1, 1, self.K)
        return adv - adv.mean(1, keepdim=True) + val


class RainbowPort:
    def __init__(self, state_size, action_size, hidden_size=512, lr=6.25e-5, gamma=0.99, buffer_size=1000000, batch_size=32, n_step=3,
                 alpha=0.5, beta=0.4, uniform_sample_prob=1e-3, v_min=-1.0, v_max=10.0, num_support=51):
        self.network = RainbowNet(state_size, action_size, num_support, hidden_size)
        self.target_network = RainbowNet(state_size, action_size, num_support, hidden_size)
        self.target_network.load_state_dict(self.network.state_dict())
        self.optimizer = torch.optim.Adam(self.network.parameters(), lr=lr)
        self.memory = PEROracle(buffer_size, uniform_sample_prob)
        self.A, self.K, self.B, self.n, self.gamma, self.alpha, self.beta = action_size, num_support, batch_size, n_step, gamma, alpha, beta
        self.v_min, self.v_max = v_min, v_max
        self.dz = (v_max - v_min) / (num_support - 1)
        self.z = torch.linspace(v_min, v_max, num_support).view(1, -1)
        self.noise = None  # tests inject [3] dicts tag -> (eps_in, eps_out)
        # single-mode loop state (rainbow.py:96-129): n-step window, stamps, beta annealing
        from collections import deque

        self.tmp_buffer = deque(maxlen=n_step)
        self.start_train_step, self.learn_period, self.target_update_period = 0, 4, 10000
        self.time_t = self.learn_period_stamp = self.target_update_stamp = self.num_learn = 0
        self.beta_add = (1.0 - beta) / 30_000_000

    @torch.no_grad()
    def act(self, state, training=True):  # rainbow.py:140-152 (noisy forward B = 1, logits2Q, argmax)
        if training and self.memory.size < max(self.B, self.start_train_step):
            return {"action": np.random.randint(0, self.A, size=(state.shape[0], 1))}
        _, q = self._pq(self.network(torch.as_tensor(state, dtype=torch.float32)))
        return {"action": torch.argmax(q, -1, keepdim=True).numpy()}

    def interact_callback(self, transition):  # rainbow.py:294-308
        out = {}
        self.tmp_buffer.append(transition)
        if len(self.tmp_buffer) == self.n:
            out["state"], out["action"], out["next_state"] = self.tmp_buffer[0]["state"], self.tmp_buffer[0]["action"], self.tmp_buffer[-1]["next_state"]
            for key in self.tmp_buffer[0].keys():
                if key not in ("state", "action", "next_state"):
                    out[key] = np.stack([t[key] for t in self.tmp_buffer], axis=1)
        return out

    def process(self, transitions, step):  # rainbow.py:255-283
        result = {}
        delta_t = step - self.time_t
        self.memory.store(transitions)
        self.time_t = step
        self.target_update_stamp += delta_t
        self.learn_period_stamp += delta_t
        self.beta = min(1.0, self.beta + self.beta_add * delta_t)
        if self.learn_period_stamp >= self.learn_period and self.memory.buffer_counter >= self.B and self.time_t >= self.start_train_step:
            result = self.learn()
            self.num_learn += 1
            self.learn_period_stamp -= self.learn_period
        if self.num_learn > 0 and self.target_update_stamp >= self.target_update_period:
            self.target_network.load_state_dict(self.network.state_dict())
            self.target_update_stamp -= self.target_update_period
        return result

    def _pq(self, logits):
        p = torch.exp(F.log_softmax(logits, dim=-1))
        return p, (self.z.view(1, 1, -1) * p).sum(-1)

    def learn(self):  # rainbow.py:154-253
        tr, weights, indices, sampled_p, mean_p = self.memory.sample(self.beta, self.B)
        t = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in tr.items()}  # base.py:61-73: everything becomes fp32
        nz = self.noise or [None, None, None]
        logit = self.network(t["state"], nz[0])
        p_logit, q = self._pq(logit)
        eye_a = torch.eye(self.A)
        p_act = (eye_a[t["action"].long()] @ p_logit).squeeze(1)
        B, K = self.B, self.K
        with torch.no_grad():
            _, q_next = self._pq(self.network(t["next_state"], nz[1]))
            tp, _ = self._pq(self.target_network(t["next_state"], nz[2]))
            a_star = q_next.argmax(-1, keepdim=True)
            tp_act = (eye_a[a_star.long()] @ tp).squeeze(1)
            Tz = self.z
            for i in reversed(range(self.n)):
                Tz = t["reward"][:, i].expand(-1, K) + (1 - t["done"][:, i]) * self.gamma * Tz
            b = torch.clamp(Tz - self.v_min, 0, self.v_max - self.v_min) / self.dz
            lo, up = b.floor().long(), b.ceil().long()
            eye_k = torch.eye(K)
            oh_l, oh_u = eye_k[lo], eye_k[up]  # (B, K, K) one-hot temporaries, as in the reference
            lluu = oh_l * (up - b).unsqueeze(-1) + oh_u * (b - lo).unsqueeze(-1)
            d0 = t["done"][:, 0, :]
            m = d0 * (oh_l * oh_u + lluu).mean(1) + (1 - d0) * (tp_act.unsqueeze(-1) * lluu).sum(1)
            m = m / m.sum(1, keepdim=True).clamp(min=1e-8)
        max_Q, max_logit, min_logit = q.max().item(), logit.max().item(), logit.min().item()
        KL = -(m * p_act.clamp(min=1e-8).log()).sum(-1)
        p_j = KL.pow(self.alpha)
        for i, p in zip(indices, p_j):  # B .item() syncs + B Python climbs
            self.memory.update_priority(p.item(), int(i))
        w = torch.FloatTensor(weights).unsqueeze(-1)
        loss = (w * KL).mean()  # (B,1) * (B,) broadcast: the reference's quirk, kept
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return {"loss": loss.item(), "beta": self.beta, "max_Q": max_Q, "max_logit": max_logit, "min_logit": min_logit,
                "sampled_p": sampled_p, "mean_p": mean_p}
