"""CPU port of the reference's DQN single-mode path -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

BASELINE.json configs[0] (config.dqn.cartpole, single actor, CPU): what one loop body of `single_train`
(run_mode.py:68-91) does for a DQN agent, restated with torch-CPU ops in the reference's style:
    network   core/network/q_network.py:8-20 + head.py:6-18   (head.l -> l -> q, ReLU between)
    act       core/agent/dqn.py:100-115       epsilon-greedy, numpy's global RNG
    learn     core/agent/dqn.py:117-151       ReplayBuffer.sample -> as_tensor -> one-hot gather -> smooth_l1 -> Adam
    process   core/agent/dqn.py:156-178       store, learn every step once step >= start_train_step, epsilon decay, hard target update
    replay    core/buffer/replay_buffer.py    (oracle.jorldy_oracle.ReplayOracle)
Pinned against the reference's own run at config.dqn.cartpole's shapes by
tests/test_oracle_golden.py::test_dqn_port_matches_reference (fixture dqn_h512, generated from the unmodified reference by
oracle/gen_golden.py).  Used by bench.py's `dqn` leg to time the reference's path on the bench box's host cores.
Never imported by jorldy_amd.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .jorldy_oracle import CartPoleOracle, ReplayOracle


class QNet(torch.nn.Module):
    """Parameter names and registration order of the reference module (checkpoint-compatible)."""

    def __init__(self, S, A, H):
        super().__init__()
        self.head = torch.nn.Module()
        self.head.l = torch.nn.Linear(S, H)
        self.l = torch.nn.Linear(H, H)
        self.q = torch.nn.Linear(H, A)
        # head.py:14, q_network.py:13-14: orthogonal_init(layer) for head.l and l (gain sqrt(2)), orthogonal_init(q, "linear") (gain 1);
        # zero biases (utils.py:110-124) -- what a run from scratch starts from
        for layer, g in ((self.head.l, torch.nn.init.calculate_gain("relu")), (self.l, torch.nn.init.calculate_gain("relu")), (self.q, torch.nn.init.calculate_gain("linear"))):
            torch.nn.init.orthogonal_(layer.weight.data, g)
            torch.nn.init.zeros_(layer.bias.data)

    def forward(self, x):
        return self.q(F.relu(self.l(F.relu(self.head.l(x)))))


class DQNPort:
    def __init__(self, state_size=4, action_size=2, hidden_size=512, lr=1e-4, gamma=0.99, epsilon_init=1.0, epsilon_min=0.01, explore_ratio=0.2,
                 buffer_size=50000, batch_size=32, start_train_step=2000, target_update_period=500, run_step=100000, lr_decay=True):
        self.network, self.target_network = QNet(state_size, action_size, hidden_size), QNet(state_size, action_size, hidden_size)
        self.target_network.load_state_dict(self.network.state_dict())
        self.optimizer = torch.optim.Adam(self.network.parameters(), lr=lr)
        self.A, self.gamma, self.B = action_size, gamma, batch_size
        self.epsilon, self.epsilon_min = epsilon_init, epsilon_min
        self.epsilon_delta = (epsilon_init - epsilon_min) / (run_step * explore_ratio)
        self.memory = ReplayOracle(buffer_size)
        self.start_train_step, self.target_update_period, self.target_update_stamp = start_train_step, target_update_period, 0
        self.num_learn, self.time_t, self.run_step, self.lr_decay = 0, 0, run_step, lr_decay

    @torch.no_grad()
    def act(self, state, training=True):  # dqn.py:100-115
        if np.random.random() < (self.epsilon if training else 0.0):
            return {"action": np.random.randint(0, self.A, size=(state.shape[0], 1))}
        return {"action": torch.argmax(self.network(torch.as_tensor(state, dtype=torch.float32)), -1, keepdim=True).numpy()}

    def learn(self):  # dqn.py:117-151
        t = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in self.memory.sample(self.B).items()}  # base.py:61-73
        one_hot = torch.eye(self.A)[t["action"].view(-1).long()]
        q = (self.network(t["state"]) * one_hot).sum(1, keepdims=True)
        with torch.no_grad():
            max_Q = torch.max(q).item()
            next_q = self.target_network(t["next_state"])
            target_q = t["reward"] + (1 - t["done"]) * self.gamma * next_q.max(1, keepdims=True).values
        loss = F.smooth_l1_loss(q, target_q)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        self.num_learn += 1
        return {"loss": loss.item(), "epsilon": self.epsilon, "max_Q": max_Q}

    def process(self, transitions, step):  # dqn.py:156-178
        result = {}
        self.memory.store(transitions)
        delta_t = step - self.time_t
        self.time_t = step
        self.target_update_stamp += delta_t
        if self.memory.size >= self.B and self.time_t >= self.start_train_step:
            result = self.learn()
            if self.lr_decay:  # base.py:93-111, cosine
                w = np.cos((np.pi / 2) * (step / self.run_step))
                for g in self.optimizer.param_groups:
                    g["lr"] = self.optimizer.defaults["lr"] * w
        if self.num_learn > 0:
            self.epsilon = max(self.epsilon_min, self.epsilon - delta_t * self.epsilon_delta)
            if self.target_update_stamp >= self.target_update_period:
                self.target_network.load_state_dict(self.network.state_dict())
                self.target_update_stamp -= self.target_update_period
        return result


def single_mode_steps(agent, env, state, step0, n_steps):
    """run_mode.py:68-91 (the loop body, without the manage-process queues): act -> env.step -> transition dict -> process.
    `env`: oracle.jorldy_oracle.CartPoleOracle(1) (gym's CartPole restated, reward shaping of core/env/gym_env.py:78).
    -> (state, number of learn() calls)"""
    n_learn = 0
    for step in range(step0 + 1, step0 + n_steps + 1):
        a = agent.act(state, True)
        nxt, rew, done = env.step(np.asarray(a["action"]).reshape(-1))
        tr = {"state": state, "next_state": nxt.astype(np.float32), "reward": rew.reshape(1, 1).astype(np.float64), "done": done.reshape(1, 1)}
        tr.update(a)
        if agent.process([tr], step):
            n_learn += 1
        state = env.obs().astype(np.float32)  # `next_state if not done else env.reset()`: the oracle env resets itself
    return state, n_learn


def make_env(seed=0):
    env = CartPoleOracle(1, seed=seed)
    return env, env.obs().astype(np.float32)
