"""Batched on-GPU acting for the value-net agents (configs[3]: Ape-X's many actors).

Reference: every actor is a process with a CPU copy of the agent; each env step is one B = 1 forward
(manager/distributed_manager.py:76-92 -> core/agent/ape_x.py:64-77 / rainbow.py:140-152 / dqn.py:76-92), and the
learner ships its whole state_dict to all of them every `update_period` (DistributedManager.sync, :55-60).

Here: the N actors' observations of one tick (uint8 frames, N x 28 KB at Atari shapes) are staged in ONE pinned slab,
copied with one async H2D, pushed through ONE forward of an acting copy of the native network (jh_rbnet_forward:
implicit-GEMM convolutions reading the uint8 frames, same kernels as the learner), and turned into actions by one
kernel (jh_value_act: Q from the outputs, per-actor epsilon-greedy with the host's random draws, first-maximum
argmax, Q of the action taken for Ape-X's actor-side priorities).  The acting copy has its own activation
workspaces and runs on its own HIP stream, so actor ticks overlap with the learner's graph replays; `sync()` is one
device-to-device copy of the flat parameter bucket (the reference's state_dict broadcast).
"""
import numpy as np
import torch

from .. import ops


class BatchedValueActors:
    """act() for n_actors environments at once on an acting copy of `agent`'s native value network.

    agent: a DQN-family / Rainbow / Ape-X agent with backend="native".  epsilons: per-actor exploration rates
    (default: Ape-X's schedule eps^(1 + i/(N-1) * alpha), ape_x.py:166-172, when the agent has one; else agent.epsilon
    for all).  Rainbow acts through its noisy layers (a fresh noise draw per tick, rainbow.py:150) and epsilon = 0."""

    def __init__(self, agent, n_actors, epsilons=None, stream=None):
        src = getattr(agent, "_net", None)
        assert src is not None and hasattr(src, "target"), "needs a native value-net agent (backend='native')"
        self.agent, self.N, self.device = agent, int(n_actors), agent.device
        state_size = (src.Cin, src.Hin, src.Win) if src.cnn else src.Cin
        self.net = ops.RainbowNet(state_size, src.A, src.K, src.H, "cnn" if src.cnn else "mlp", self.N, self.device, kind=src.kind, noise_type=src.noise_type)
        self.stream = stream or torch.cuda.Stream(device=self.device)
        self.noisy = src.kind == "rainbow"
        if epsilons is None:
            if self.noisy:
                epsilons = np.zeros(self.N)
            elif hasattr(agent, "epsilon_alpha") and self.N > 1:
                epsilons = np.asarray([agent.epsilon ** (1 + (i / (self.N - 1)) * agent.epsilon_alpha) for i in range(self.N)])
            else:
                epsilons = np.full(self.N, float(getattr(agent, "epsilon", 0.0)))
        self.eps = np.ascontiguousarray(epsilons, dtype=np.float32)
        assert self.eps.size == self.N
        self.v_min, self.v_max = float(getattr(agent, "v_min", 0.0)), float(getattr(agent, "v_max", 0.0))
        shape = (self.N,) + ((src.Cin, src.Hin, src.Win) if src.cnn else (src.Cin,))
        self.x_dtype = torch.uint8 if src.cnn else torch.float32
        self._x_pin = torch.empty(shape, dtype=self.x_dtype, pin_memory=True)  # the tick's observations, staged once
        self._x_dev = torch.empty(shape, dtype=self.x_dtype, device=self.device)
        self._logits = torch.empty(self.N, src.A, src.K, dtype=torch.float32, device=self.device)
        self._act_dev = torch.empty(self.N, dtype=torch.int64, device=self.device)
        self._q_dev = torch.empty(self.N, dtype=torch.float32, device=self.device)
        self._act_pin = torch.empty(self.N, dtype=torch.int64, pin_memory=True)
        self._q_pin = torch.empty(self.N, dtype=torch.float32, pin_memory=True)
        self._noise = torch.empty(max(1, self.net.noise_len), dtype=torch.float32, device=self.device) if self.noisy else None
        self._normal = ops.NormalSource(self.device) if self.noisy else None
        self._done = torch.cuda.Event()
        self.ticks = 0
        self.sync()

    @property
    def obs_slab(self):
        """numpy view [N, *obs_shape] of the pinned staging slab: vectorised envs can write their observations in place."""
        return self._x_pin.numpy()

    def sync(self):
        """DistributedManager.sync + BaseAgent.sync_in (base.py:75-85) for all actors: one D2D copy of the learner's
        flat parameter bucket on the acting stream (ordered after the learner's last enqueued update)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.net.params.copy_(self.agent._net.params, non_blocking=True)

    @torch.no_grad()
    def act(self, obs=None, training=True, random_actions=False):
        """obs: numpy [N, *obs_shape] (uint8 frames / float32 vectors) or None when the envs wrote into `obs_slab`.
        -> {"action": int64 [N, 1], "q": float32 [N, 1]} (blocking: the envs need the actions).
        random_actions: the warm-up branch of Rainbow.act / the agents' `memory.size < start_train_step` phase."""
        if obs is not None:
            np.copyto(self._x_pin.numpy(), obs, casting="same_kind")
        N = self.N
        greedy = not training
        eps = u = ra = None
        if not greedy and (random_actions or np.any(self.eps > 0)):
            eps = np.ones(N, np.float32) if random_actions else self.eps
            u = np.random.random(N)                                  # `np.random.random() < epsilon` per actor
            ra = np.random.randint(0, self.net.A, size=N)            # `np.random.randint(0, action_size)` per actor
        with torch.cuda.stream(self.stream):
            self._x_dev.copy_(self._x_pin, non_blocking=True)
            noise = None
            if self.noisy and training:
                noise = self._normal.fill(self._noise)
            self.net.forward(self._x_dev, which=0, noise=noise, out=self._logits)
            ops.value_act(self._logits, self.v_min, self.v_max, eps, u, ra, out=(self._act_dev, self._q_dev))
            self._act_pin.copy_(self._act_dev, non_blocking=True)
            self._q_pin.copy_(self._q_dev, non_blocking=True)
            self._done.record(self.stream)
        self._done.synchronize()
        self.ticks += 1
        return {"action": self._act_pin.numpy().reshape(N, 1).copy(), "q": self._q_pin.numpy().reshape(N, 1).copy()}


class VecNStepApeX:
    """Ape-X's n-step assembler with actor-side priorities (ape_x.py:174-199) for N actors at once: the per-actor
    deque(maxlen = n + 1) of the reference becomes rolling arrays [n + 1, N, ...]; one push per tick emits N n-step
    transitions {state_t, action_t, reward[n], done[n], next_state = state_{t+n}} and priority = |G_n - q_t| where
    G_n folds r_i + (1 - d_i) gamma ... onto q_{t+n}.  Windows straddle episode ends like the reference's."""

    def __init__(self, n_actors, n_step, gamma, obs_shape, obs_dtype=np.uint8):
        self.N, self.n, self.gamma = n_actors, n_step, gamma
        L = n_step + 1
        self.state = np.zeros((L, n_actors) + tuple(obs_shape), dtype=obs_dtype)
        self.action = np.zeros((L, n_actors, 1), np.int64)
        self.reward = np.zeros((L, n_actors, 1), np.float32)
        self.done = np.zeros((L, n_actors, 1), np.float32)
        self.q = np.zeros((L, n_actors, 1), np.float32)
        self.count = 0

    def push(self, state, action, reward, done, q):
        """-> None until n + 1 ticks have been seen, then (cols dict of N rows, priorities [N])."""
        L = self.n + 1
        i = self.count % L
        self.state[i], self.action[i], self.reward[i], self.done[i], self.q[i] = state, action, reward, done, q
        self.count += 1
        if self.count < L:
            return None
        order = [(i + 1 + k) % L for k in range(L)]  # oldest .. newest
        o0, oN = order[0], order[-1]
        tq = self.q[oN].astype(np.float32)
        for k in reversed(order[:-1]):
            tq = self.reward[k] + (1.0 - self.done[k]) * self.gamma * tq
        cols = {"state": self.state[o0], "action": self.action[o0],
                "reward": np.stack([self.reward[k] for k in order[:-1]], axis=1), "next_state": self.state[oN],
                "done": np.stack([self.done[k] for k in order[:-1]], axis=1).astype(np.uint8)}
        return cols, np.abs(tq - self.q[o0]).reshape(-1).astype(np.float64)
