"""CPU port of the reference's PPO sync-mode hot path -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Restates, in the reference's own style (per-transition dicts, np.stack, Python GAE loop, B=1
acting, 5 `.item()` per minibatch), what runs per sync iteration of
`main.py --sync --config config.ppo.cartpole`:
    Actor.run            manager/distributed_manager.py:76-92
    RolloutBuffer        core/buffer/rollout_buffer.py:11-21 + base.py:42-56
    PPO.act / PPO.learn  core/agent/ppo.py:55-185
Used by bench.py's `cpu_baseline` leg (kind "port") and pinned against the reference's own run in
tests/test_oracle_golden.py::test_ppo_port_matches_reference.  Never imported by jorldy_amd.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions import Categorical, Normal

from .jorldy_oracle import CartPoleOracle, RolloutOracle


class _PolicyValue(torch.nn.Module):
    """discrete/continuous policy-value net with the reference's parameter names (core/network/policy_value.py:8-57) on the MLP head
    (head.py:6-18; S an int) or the CNN head (head.py:21-61; S = (C, H, W), frames divided by 255 in the head)."""

    def __init__(self, S, A, H, continuous):
        super().__init__()
        self.continuous = continuous
        self.head = torch.nn.Module()
        self.cnn = not isinstance(S, (int, np.integer))
        gain = torch.nn.init.calculate_gain("relu")
        if self.cnn:
            self.head.conv1 = torch.nn.Conv2d(S[0], 32, kernel_size=8, stride=4)
            self.head.conv2 = torch.nn.Conv2d(32, 64, kernel_size=4, stride=2)
            self.head.conv3 = torch.nn.Conv2d(64, 64, kernel_size=3, stride=1)
            d = [((S[1] - 8) // 4 + 1), ((S[2] - 8) // 4 + 1)]
            d = [(v - 4) // 2 + 1 for v in d]
            d = [v - 3 + 1 for v in d]
            for conv in (self.head.conv1, self.head.conv2, self.head.conv3):
                torch.nn.init.orthogonal_(conv.weight.data, gain)
                torch.nn.init.zeros_(conv.bias.data)
            self.l = torch.nn.Linear(64 * d[0] * d[1], H)
        else:
            self.head.l = torch.nn.Linear(S, H)
            self.l = torch.nn.Linear(H, H)
        if continuous:
            self.mu = torch.nn.Linear(H, A)
            self.log_std = torch.nn.Linear(H, A)
        else:
            self.pi = torch.nn.Linear(H, A)
        self.v = torch.nn.Linear(H, 1)
        for lin, g in (((self.head.l, gain),) if not self.cnn else ()) + ((self.l, gain), (self.v, 1.0)):
            torch.nn.init.orthogonal_(lin.weight.data, g)
            torch.nn.init.zeros_(lin.bias.data)
        if continuous:
            torch.nn.init.orthogonal_(self.mu.weight.data, 1.0)
            torch.nn.init.orthogonal_(self.log_std.weight.data, torch.nn.init.calculate_gain("tanh"))
            torch.nn.init.zeros_(self.mu.bias.data)
            torch.nn.init.zeros_(self.log_std.bias.data)
        else:
            torch.nn.init.orthogonal_(self.pi.weight.data, 0.01)
            torch.nn.init.zeros_(self.pi.bias.data)

    def forward(self, x):
        if self.cnn:
            x = x / 255.0
            x = F.relu(self.head.conv3(F.relu(self.head.conv2(F.relu(self.head.conv1(x))))))
            x = F.relu(self.l(x.view(x.size(0), -1)))
        else:
            x = F.relu(self.l(F.relu(self.head.l(x))))
        if self.continuous:
            return torch.clamp(self.mu(x), -5.0, 5.0), torch.tanh(self.log_std(x)).exp(), self.v(x)
        return torch.exp(F.log_softmax(self.pi(x), dim=-1)), self.v(x)


class PPOPort:
    def __init__(self, state_size, action_size, hidden_size=512, continuous=False, lr=2.5e-4, gamma=0.99, batch_size=32,
                 n_step=128, n_epoch=3, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0,
                 run_step=1e6, use_standardization=True, lr_decay=True):
        self.cont = continuous
        self.network = _PolicyValue(state_size, action_size, hidden_size, continuous)
        self.optimizer = torch.optim.Adam(self.network.parameters(), lr=lr)
        self.memory = RolloutOracle()
        self.gamma, self.batch_size, self.n_step, self.n_epoch = gamma, batch_size, n_step, n_epoch
        self._lambda, self.epsilon_clip, self.vf_coef, self.ent_coef = _lambda, epsilon_clip, vf_coef, ent_coef
        self.clip_grad_norm, self.run_step, self.use_standardization, self.lr_decay = clip_grad_norm, run_step, use_standardization, lr_decay
        self.time_t = 0
        self.learn_stamp = 0

    def as_tensor(self, x):
        return torch.as_tensor(x, dtype=torch.float32)

    @torch.no_grad()
    def act(self, state, training=True):  # ppo.py:55-69
        if self.cont:
            mu, std, _ = self.network(self.as_tensor(state))
            action = torch.tanh(torch.normal(mu, std) if training else mu)
        else:
            pi, _ = self.network(self.as_tensor(state))
            action = torch.multinomial(pi, 1) if training else torch.argmax(pi, dim=-1, keepdim=True)
        return {"action": action.numpy()}

    def learn(self):  # ppo.py:71-185
        tr = self.memory.sample()
        tr = {k: self.as_tensor(v) for k, v in tr.items()}
        state, action, reward, next_state, done = tr["state"], tr["action"], tr["reward"], tr["next_state"], tr["done"]
        with torch.no_grad():
            if self.cont:
                mu, std, value = self.network(state)
                z = torch.atanh(torch.clamp(action, -1 + 1e-7, 1 - 1e-7))
                log_prob_old = Normal(mu, std).log_prob(z)
            else:
                pi, value = self.network(state)
                log_prob_old = pi.gather(1, action.long()).log()
            next_value = self.network(next_state)[-1]
            delta = reward + (1 - done) * self.gamma * next_value - value
            adv = delta.clone()
            adv, done_v = adv.view(-1, self.n_step), done.view(-1, self.n_step)
            for t in reversed(range(self.n_step - 1)):  # the Python scan the HIP kernel replaces
                adv[:, t] += (1 - done_v[:, t]) * self.gamma * self._lambda * adv[:, t + 1]
            ret = adv.view(-1, 1) + value
            if self.use_standardization:
                adv = (adv - adv.mean(dim=1, keepdim=True)) / (adv.std(dim=1, keepdim=True) + 1e-7)
            adv = adv.view(-1, 1)
        mean_ret = ret.mean().item()
        al, cl, el, ratios, probs = [], [], [], [], []
        idxs = np.arange(len(reward))
        for _ in range(self.n_epoch):
            np.random.shuffle(idxs)
            for offset in range(0, len(reward), self.batch_size):
                idx = idxs[offset : offset + self.batch_size]
                _state, _action, _value, _ret, _adv, _lpo = (x[idx] for x in (state, action, value, ret, adv, log_prob_old))
                if self.cont:
                    mu, std, value_pred = self.network(_state)
                    m = Normal(mu, std)
                    log_prob = m.log_prob(torch.atanh(torch.clamp(_action, -1 + 1e-7, 1 - 1e-7)))
                else:
                    pi, value_pred = self.network(_state)
                    m = Categorical(pi)
                    log_prob = m.log_prob(_action.squeeze(-1)).unsqueeze(-1)
                ratio = (log_prob - _lpo).sum(1, keepdim=True).exp()
                surr1 = ratio * _adv
                surr2 = torch.clamp(ratio, 1 - self.epsilon_clip, 1 + self.epsilon_clip) * _adv
                actor_loss = -torch.min(surr1, surr2).mean()
                v_clip = _value + torch.clamp(value_pred - _value, -self.epsilon_clip, self.epsilon_clip)
                critic_loss = torch.max(F.mse_loss(value_pred, _ret), F.mse_loss(v_clip, _ret)).mean()
                entropy_loss = -m.entropy().mean()
                loss = actor_loss + self.vf_coef * critic_loss + self.ent_coef * entropy_loss
                self.optimizer.zero_grad(set_to_none=True)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.clip_grad_norm)
                self.optimizer.step()
                probs.append(log_prob.exp().min().item())
                ratios.append(ratio.max().item())
                al.append(actor_loss.item())
                cl.append(critic_loss.item())
                el.append(entropy_loss.item())
        return {"actor_loss": np.mean(al), "critic_loss": np.mean(cl), "entropy_loss": np.mean(el), "max_ratio": max(ratios),
                "min_prob": min(probs), "mean_ret": mean_ret}

    def process(self, transitions, step):  # ppo.py:187-202
        result = {}
        self.memory.store(transitions)
        self.learn_stamp += step - self.time_t
        self.time_t = step
        if self.learn_stamp >= self.n_step:
            result = self.learn()
            if self.lr_decay:
                w = np.cos((np.pi / 2) * (step / self.run_step))
                for g in self.optimizer.param_groups:
                    g["lr"] = self.optimizer.defaults["lr"] * w
            self.learn_stamp = 0
        return result


class _OneEnv:
    """One worker's env with the leading batch dim of 1 the reference envs add (gym_env.py:42-66)."""

    def __init__(self, seed):
        self.env = CartPoleOracle(1, seed=seed)

    def reset_obs(self):
        return self.env.obs().astype(np.float32)

    def step(self, action):
        nxt, rew, done = self.env.step(np.asarray(action).reshape(-1))
        return nxt, rew.reshape(1, 1).astype(np.float64), done.reshape(1, 1), self.env.obs().astype(np.float32)


def sync_iteration(agent, envs, states, update_period):
    """One loop body of sync_distributed_train (run_mode.py:180-186), workers run one after another
    in-process: DistributedManager.run (worker-major concat) -> agent.process.
    Returns (result, n_transitions)."""
    transitions = []
    for w, env in enumerate(envs):  # Actor.run, distributed_manager.py:76-92
        state = states[w]
        for _ in range(update_period):
            action_dict = agent.act(state, training=True)
            next_state, reward, done, after = env.step(action_dict["action"])
            tr = {"state": state, "next_state": next_state, "reward": reward, "done": done}
            tr.update(action_dict)
            transitions.append(tr)
            state = after if done[0, 0] else next_state
        states[w] = state
    return transitions


# ---------------------------------------------------------------------------------------------------
# "8 procs" variant of the CPU baseline (SURVEY.md §8d / BASELINE.md §3): the reference's sync mode runs
# its Actors as separate processes (Ray); every iteration the learner ships the full state_dict to them
# (DistributedManager.sync, distributed_manager.py:55-60) and gets update_period x num_workers transition
# dicts back (run, :26-31).  Ray is not installable here: plain `multiprocessing` (spawn) workers with
# pipes play its part -- same per-step work in the actors, same pickled payloads in both directions.
def _worker_main(conn, seed, S, A, H, cont):
    torch.set_num_threads(1)
    agent = PPOPort(S, A, H, cont)
    env = _OneEnv(seed=seed)
    state = env.reset_obs()
    while True:
        msg = conn.recv()
        if msg is None:
            break
        weights, steps = msg
        agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})  # BaseAgent.sync_in
        out = []
        for _ in range(steps):  # Actor.run, distributed_manager.py:76-92
            action_dict = agent.act(state, training=True)
            next_state, reward, done, after = env.step(action_dict["action"])
            tr = {"state": state, "next_state": next_state, "reward": reward, "done": done}
            tr.update(action_dict)
            out.append(tr)
            state = after if done[0, 0] else next_state
        conn.send(out)


class ProcWorkers:
    """num_workers actor processes; run(agent, T) = sync(weights) + one DistributedManager.run."""

    def __init__(self, num_workers, S, A, H, cont=False):
        import multiprocessing as mp

        ctx = mp.get_context("spawn")  # the bench process holds a HIP context: never fork it
        self.conns, self.procs = [], []
        for w in range(num_workers):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(b, w, S, A, H, cont), daemon=True)
            p.start()
            self.conns.append(a)
            self.procs.append(p)

    def run(self, agent, steps):
        weights = {k: v.detach().cpu().numpy() for k, v in agent.network.state_dict().items()}  # BaseAgent.sync_out
        for c in self.conns:
            c.send((weights, steps))
        transitions = []
        for c in self.conns:  # worker-major concat
            transitions += c.recv()
        return transitions

    def close(self):
        for c in self.conns:
            try:
                c.send(None)
            except Exception:
                pass
        for p in self.procs:
            p.join(timeout=5)
