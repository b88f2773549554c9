import os

import numpy as np
import torch

from ... import ops
from ..buffer import RolloutBuffer
from ..buffer.base import h2d_small
from ..network import Network
from ..optimizer import Optimizer
from .base import BaseAgent


class PPO(BaseAgent):
    """core/agent/ppo.py:10-202 (+ the REINFORCE base, reinforce.py:14-142) with the learner-side hot
    path on HIP kernels:

      rollout -> GPU SoA store (pinned staging)            RolloutBuffer        ppo.py:72-74
      log pi_old, GAE scan, per-row standardisation        jh_logp_*, jh_gae    ppo.py:83-110
      minibatch gathers x[idx] + clipped surrogate +       jh_ppo_loss_*        ppo.py:122-165
        clipped value + entropy, forward AND backward
      the 5 `.item()` syncs per minibatch                  one D2H of a [n_updates, 8] stats array

    The encoder fwd/bwd, clip_grad_norm_ and Adam stay torch ops on the same stream in this layer.
    Constructor arguments, `act`, `process`, result keys, checkpoint format are the reference's.
    """

    def __init__(self, state_size, action_size, hidden_size=512, network="discrete_policy_value", head="mlp",
                 optim_config={"name": "adam"}, gamma=0.99, use_standardization=True, run_step=1e6, lr_decay=True,
                 device=None, batch_size=32, n_step=128, n_epoch=3, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0,
                 ent_coef=0.01, clip_grad_norm=1.0, num_workers=1, **kwargs):
        self.device = self._require_gpu(device)
        self.action_type = network.split("_")[0]
        assert self.action_type in ["continuous", "discrete"]
        self.network = Network(network, state_size, action_size, D_hidden=hidden_size, head=head).to(self.device)
        self.optimizer = Optimizer(**optim_config, params=self.network.parameters())
        self.gamma = gamma
        self.use_standardization = use_standardization
        self.memory = RolloutBuffer(device=self.device)
        self.run_step = run_step
        self.lr_decay = lr_decay
        self.batch_size = batch_size
        self.n_step = n_step
        self.n_epoch = n_epoch
        self._lambda = _lambda
        self.epsilon_clip = epsilon_clip
        self.vf_coef = vf_coef
        self.ent_coef = ent_coef
        self.clip_grad_norm = clip_grad_norm
        self.num_workers = num_workers
        self.time_t = 0
        self.learn_stamp = 0
        self._stats = None
        self.grad_sync = None  # data-parallel hook: jorldy_amd.parallel.FlatGradSync (RCCL all-reduce)

    @torch.no_grad()
    def act(self, state, training=True):
        self.network.train(training)
        if self.action_type == "continuous":
            mu, std, _ = self.network(self.as_tensor(state))
            z = torch.normal(mu, std) if training else mu
            action = torch.tanh(z)
        else:
            pi, _ = self.network(self.as_tensor(state))
            action = torch.multinomial(pi, 1) if training else torch.argmax(pi, dim=-1, keepdim=True)
        return {"action": action.cpu().numpy()}

    def learn(self):
        tr = self.memory.sample()  # float32 device tensors, arrival (worker-major) order
        state, action, reward = tr["state"], tr["action"], tr["reward"]
        next_state, done = tr["next_state"], tr["done"]
        M = reward.shape[0]
        cont = self.action_type == "continuous"

        with torch.no_grad():  # ppo.py:83-110
            if cont:
                mu_raw, ls_raw, value = self.network.raw(state)
                log_prob_old = ops.logp_continuous(mu_raw, ls_raw, action)
            else:
                logits, value = self.network.raw(state)
                log_prob_old = ops.logp_discrete(logits, action)
            next_value = self.network.raw(next_state)[-1]
            adv, ret = ops.gae(reward, done, value, next_value, self.n_step, self.gamma, self._lambda, self.use_standardization)
            value = value.contiguous()
            mean_ret_t = ret.mean()

        n_mb = (M + self.batch_size - 1) // self.batch_size
        n_upd = self.n_epoch * n_mb
        if self._stats is None or self._stats.shape[0] < n_upd + 1:
            self._stats = torch.zeros(n_upd + 1, 8, dtype=torch.float32, device=self.device)
        stats = self._stats
        idxs = np.arange(M)
        k = 0
        for _ in range(self.n_epoch):
            np.random.shuffle(idxs)  # ppo.py:118 -- same global-RNG call as the reference
            idxs_d = h2d_small(idxs.astype(np.int64), self.device)
            for offset in range(0, M, self.batch_size):
                idx = idxs_d[offset : offset + self.batch_size]
                _state = state.index_select(0, idx)
                if cont:
                    mu_raw, ls_raw, value_pred = self.network.raw(_state)
                    g_mu, g_ls, g_v, _ = ops.ppo_loss_continuous(mu_raw.detach(), ls_raw.detach(), value_pred.detach(), idx, action, adv, ret, value, log_prob_old, self.epsilon_clip, self.vf_coef, self.ent_coef, stats=stats[k])
                    outs, grads = [mu_raw, ls_raw, value_pred], [g_mu, g_ls, g_v]
                else:
                    logits, value_pred = self.network.raw(_state)
                    g_z, g_v, _ = ops.ppo_loss_discrete(logits.detach(), value_pred.detach(), idx, action, adv, ret, value, log_prob_old, self.epsilon_clip, self.vf_coef, self.ent_coef, stats=stats[k])
                    outs, grads = [logits, value_pred], [g_z, g_v]
                self.optimizer.zero_grad(set_to_none=True)
                torch.autograd.backward(outs, grads)
                if self.grad_sync is not None:
                    self.grad_sync()
                torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.clip_grad_norm)
                self.optimizer.step()
                k += 1
        stats[n_upd, 0] = mean_ret_t
        s = stats[: n_upd + 1].cpu().numpy().astype(np.float64)  # the only host sync of learn()
        return {
            "actor_loss": np.mean(s[:n_upd, 1]),
            "critic_loss": np.mean(s[:n_upd, 2]),
            "entropy_loss": np.mean(s[:n_upd, 3]),
            "max_ratio": float(s[:n_upd, 4].max()),
            "min_prob": float(s[:n_upd, 5].min()),
            "mean_ret": float(s[n_upd, 0]),
        }

    def process(self, transitions, step):
        """ppo.py:187-202.  `transitions` is the reference's List[Dict] or an SoA dict of arrays."""
        result = {}
        if isinstance(transitions, dict):
            self.memory.store_soa(transitions)
        else:
            self.memory.store(transitions)
        delta_t = step - self.time_t
        self.time_t = step
        self.learn_stamp += delta_t
        if self.learn_stamp >= self.n_step:
            result = self.learn()
            if self.lr_decay:
                self.learning_rate_decay(step)
            self.learn_stamp = 0
        return result

    def save(self, path):
        print(f"...Save model to {path}...")
        torch.save({"network": self.network.state_dict(), "optimizer": self.optimizer.state_dict()}, os.path.join(path, "ckpt"))

    def load(self, path):
        print(f"...Load model from {path}...")
        checkpoint = torch.load(os.path.join(path, "ckpt"), map_location=self.device, weights_only=False)
        self.network.load_state_dict(checkpoint["network"])
        self.optimizer.load_state_dict(checkpoint["optimizer"])
