import os
from collections import deque

import numpy as np
import torch

from ... import ops
from ..buffer import PERBuffer, ReplayBuffer
from ..network import Network
from .base import BaseAgent
from .native_net import NativeValueNetMixin, require_native


class DQN(NativeValueNetMixin, BaseAgent):
    """core/agent/dqn.py:14-203.  learn(): uniform sample (same numpy draw as the reference) ->
    fused gather -> Q(s), Q_target(s') -> jh_td_loss (target, Huber, dQ) -> backward -> Adam.
    One host sync per learn (loss/max_Q read-back), not three.

    The network, its backward and the optimizer step run on libjorldy_hip (ops.RainbowNet, jh_rbnet_*): discrete_q_network /
    dueling with an mlp / cnn head, plain Adam or RMSprop.  One backend: any other configuration raises
    (native_net.NATIVE_ELIGIBLE) instead of switching to library kernels.  learn() is replayed as one hipGraph."""

    action_type = "discrete"
    _td = dict(double=False, per=False, n_step=0)

    def __init__(self, state_size, action_size, hidden_size=512, optim_config={"name": "adam"},
                 network="discrete_q_network", head="mlp", gamma=0.99, epsilon_init=1.0, epsilon_min=0.1,
                 epsilon_eval=0.0, explore_ratio=0.1, buffer_size=50000, batch_size=64, start_train_step=2000,
                 target_update_period=500, device=None, run_step=1e6, num_workers=1, lr_decay=True, use_graph=True, backend=None,
                 frame_dedup=False, **kwargs):
        self.device = self._require_gpu(device)
        self.use_graph = use_graph
        self.grad_sync = None  # data-parallel hook (jorldy_amd.parallel.attach_data_parallel)
        self.graph_with_collective = os.environ.get("JH_GRAPH_DP", "1") == "1"
        self.action_size = action_size
        self.action_type = "discrete"
        require_native(backend, network, head, state_size, hidden_size, optim_config)
        self.backend = "native"
        self._net = None
        self._init_native(network, state_size, action_size, 1, hidden_size, head, batch_size, optim_config,
                          Network(network, state_size, action_size, D_hidden=hidden_size, head=head))
        self.gamma = gamma
        self.epsilon = epsilon_init
        self.epsilon_init = epsilon_init
        self.epsilon_min = epsilon_min
        self.epsilon_eval = epsilon_eval
        self.explore_step = run_step * explore_ratio
        self.epsilon_delta = (epsilon_init - epsilon_min) / self.explore_step
        self.buffer_size = buffer_size
        self._frame_dedup = bool(frame_dedup)  # image replay as single frames + slot numbers (buffer/frame_dedup.py)
        self.memory = ReplayBuffer(buffer_size, device=self.device, frame_dedup=self._frame_dedup)
        self.memory.defer_rows = 16  # per-step stores coalesce into one ring append before the next learn()
        self.batch_size = batch_size
        self.start_train_step = start_train_step
        self.target_update_stamp = 0
        self.target_update_period = target_update_period
        self.num_learn = 0
        self.time_t = 0
        self.num_workers = num_workers
        self.run_step = run_step
        self.lr_decay = lr_decay
        self.clip_grad_norm = None
        self._stats, self._stats_np = self._mapped_stats(4)
        self._static = None
        self._graph = None
        self._warm = False
        self._noise = None
        # JH_LEARN_OVERLAP=1 (opt-in, measured SLOWER): learn() as a graph with BRANCHES -- work that nothing on the critical path waits for on
        # a side stream forked from / joined to the learner's stream inside the captured body: the PER priority write-back (per_delta + tree
        # climb: 25 us at B = 32, 45 us at B = 512) next to the backward pass + optimizer, Rainbow's noisy weight sets (17 us) next to the
        # trunk.  On this stack a fork / join inside a hipGraph costs more than the 42 us it hides: Rainbow 2 969 -> 2 655 updates/s, Ape-X
        # 988 -> 967 (two alternating pairs each, round 4).  Default: one stream, one chain.
        self._overlap = os.environ.get("JH_LEARN_OVERLAP", "0") == "1"

    @torch.no_grad()
    def act(self, state, training=True):
        self.network.train(training)
        epsilon = self.epsilon if training else self.epsilon_eval
        if np.random.random() < epsilon:
            batch_size = state[0].shape[0] if isinstance(state, list) else state.shape[0]
            action = np.random.randint(0, self.action_size, size=(batch_size, 1))
        else:
            action = self._act_greedy(state)
            if action is None:
                action = torch.argmax(self.network(self.as_tensor(state)), -1, keepdim=True).cpu().numpy()
        return {"action": action}

    # ------------------------------------------------------------------------------------------
    def learning_rate_decay(self, step, optimizers=None, mode="cosine"):
        return self._native_lr_decay(step, mode)

    def _alloc_static(self):
        """Fixed-address buffers of one learn(): sampled indices / weights and the gathered batch."""
        return self._alloc_static_native()

    def _draw(self, st):
        """Host side of sampling (the reference's numpy global-RNG draws) -> st["idx"] (+ st["w"])."""
        from ..buffer.base import h2d_small

        st["idx"].copy_(h2d_small(self.memory.sample_indices(self.batch_size).astype(np.int64), self.device))
        return None

    def _idx_offset(self):
        return 0

    def _fork(self):
        """(main, side): the side stream, made to wait for everything enqueued on the current stream so far (the fork of a graph branch)."""
        main = torch.cuda.current_stream()
        side = self.__dict__.get("_side_stream")
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.device)
        side.wait_stream(main)
        return main, side

    def _write_back_priorities(self, st, prio):
        """per.py:67-70 without the B `.item()` syncs; on a side branch when overlapping: nothing before the next sample reads the tree.
        -> the side stream to join at the end of the body, or None."""
        if not self._overlap:
            self.memory.update_priorities(st["idx"], prio)
            return None
        main, side = self._fork()
        with torch.cuda.stream(side):
            self.memory.update_priorities(st["idx"], prio)
        return side

    def _learn_body(self, st):
        net, B, A = self._net, self.batch_size, self.action_size
        tr = self.memory.gather(st["idx"], idx_offset=self._idx_offset(), as_float=self._as_float(), out=st["tr"])
        lg = net.learn_forward(st["x_all"], B, None, st["logits"])  # online(s), online(s'), target(s') in shared launches
        q, next_q, next_target_q = lg[0].view(B, A), lg[1].view(B, A), lg[2].view(B, A)
        g, prio, _ = ops.td_loss(q, next_target_q, tr["action"], tr["reward"], tr["done"], self.gamma, q_next_online=next_q if self._td["double"] else None,
                                 weights=st["w"] if self._td["per"] else None, alpha=getattr(self, "alpha", 0.0),
                                 n_step=self._td["n_step"] and self.n_step, stats=self._stats)
        side = self._write_back_priorities(st, prio) if self._td["per"] else None
        net.backward(g)
        if self.grad_sync is not None:  # data-parallel learners: one all-reduce of the flat gradient bucket
            self.grad_sync.reduce_flat(net.grads)
        net.optim_step(self._opt_name, self.clip_grad_norm)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)  # join: the next sample / store sees the updated tree

    def _run_learn(self):
        """Sample on the host (eager), then run the body: eagerly the first time, captured into a hipGraph the second time,
        replayed after."""
        if self._static is None or self._static["store"] is not self.memory._store:
            self._static, self._graph = self._alloc_static(), None
        st = self._static
        self.memory.flush()  # held per-step stores -> HBM before anything (possibly a replayed graph) reads the ring
        extra = self._draw(st)
        graphable = (self.use_graph and (self._noise is None or isinstance(self._noise, str)) and not ops._PROF["lib"] and not getattr(self, "_graph_failed", False)
                     and (self.grad_sync is None or (self.graph_with_collective and getattr(self.grad_sync, "capturable", True))))
        if graphable and self._graph is None and self._warm:
            try:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with ops.graph_capture(g):  # thread_local capture, garbage collector held off, stream restored if it fails
                    self._learn_body(st)
                self._graph = g
            except Exception as e:
                self._graph, self._graph_failed, graphable = None, True, False
                torch.cuda.synchronize()
                print(f"[jorldy_amd] hipGraph capture of {type(self).__name__}.learn() failed ({type(e).__name__}: {e}); running eagerly")
        if graphable and self._graph is not None:
            self._graph.replay()
        else:
            self._learn_body(st)
            self._warm = True
        self.num_learn += 1
        self._adam_steps += 1
        return extra

    def _learn_stats(self, view, marks, tensor):
        """Run one learn(); -> (loss statistics, PER statistics or None) as host arrays."""
        tree_np = getattr(getattr(self.memory, "_tree", None), "stats_np", None) if self._td["per"] else None
        mapped = view is not None and (not self._td["per"] or tree_np is not None)
        if mapped:
            for m in marks:
                view[m] = -1.0
        stats64 = self._run_learn()
        if mapped:
            self._await_marks(view, marks, type(self).__name__ + ".learn()")
            # the sampling kernel ran (and ended) in front of the kernels that wrote `view`: its statistics are there too
            return view.copy(), (tree_np.copy() if self._td["per"] else None)
        if self._td["per"]:
            return tuple(self._read_stats(tensor, stats64))
        return self._read_stats(tensor)[0], None

    def learn(self):
        s, p = self._learn_stats(self._stats_np, (3,), self._stats)
        result = {"loss": float(s[0]), "epsilon": self.epsilon, "max_Q": float(s[1])}
        if self._td["per"]:
            result.update({"sampled_p": float(p[0]), "mean_p": float(p[1])})
        return result

    def update_target(self):
        return self._net.sync_target()

    def _store(self, transitions):
        if isinstance(transitions, dict):
            self.memory.store_soa(transitions)
        else:
            self.memory.store(transitions)

    def process(self, transitions, step):
        """dqn.py:156-178."""
        result = {}
        self._store(transitions)
        delta_t = step - self.time_t
        self.time_t = step
        self.target_update_stamp += delta_t
        if self.memory.size >= self.batch_size and self.time_t >= self.start_train_step:
            result = self.learn()
            if self.lr_decay:
                self.learning_rate_decay(step)
        if self.num_learn > 0:
            self.epsilon_decay(delta_t)
            if self.target_update_stamp >= self.target_update_period:
                self.update_target()
                self.target_update_stamp -= self.target_update_period
        return result

    def epsilon_decay(self, delta_t):
        self.epsilon = max(self.epsilon_min, self.epsilon - delta_t * self.epsilon_delta)

    def save(self, path):
        return self._native_save(path)

    def load(self, path):
        return self._native_load(path)

    def set_distributed(self, id):
        self.epsilon = id / self.num_workers
        return self


class Double(DQN):
    """core/agent/double.py:9-52."""

    _td = dict(double=True, per=False, n_step=0)


class Multistep(DQN):
    """core/agent/multistep.py:12-104."""

    _td = dict(double=False, per=False, n_step=1)

    def __init__(self, n_step=5, **kwargs):
        super().__init__(**kwargs)
        self.n_step = n_step
        self.tmp_buffer = deque(maxlen=n_step)

    def interact_callback(self, transition):
        """multistep.py:90-104: sliding window that is NOT reset at episode ends."""
        _transition = {}
        self.tmp_buffer.append(transition)
        if len(self.tmp_buffer) == self.n_step:
            _transition["state"] = self.tmp_buffer[0]["state"]
            _transition["action"] = self.tmp_buffer[0]["action"]
            _transition["next_state"] = self.tmp_buffer[-1]["next_state"]
            for key in self.tmp_buffer[0].keys():
                if key not in ["state", "action", "next_state"]:
                    _transition[key] = np.stack([t[key] for t in self.tmp_buffer], axis=1)
        return _transition


class PER(DQN):
    """core/agent/per.py:9-122: double-Q target, |td|^alpha priorities, IS-weighted MSE."""

    _td = dict(double=True, per=True, n_step=0)

    def __init__(self, alpha=0.6, beta=0.4, learn_period=16, uniform_sample_prob=1e-3, run_step=1e6, **kwargs):
        super().__init__(run_step=run_step, **kwargs)
        self.memory = PERBuffer(self.buffer_size, uniform_sample_prob, device=self.device, frame_dedup=self._frame_dedup)
        self.memory.defer_rows = 16  # per-step stores coalesce into one ring append before the next learn()
        self.alpha = alpha
        self.beta = beta
        self.beta_add = (1 - beta) / run_step
        self.learn_period = learn_period
        self.learn_period_stamp = 0

    def _draw(self, st):
        return self.memory.sample_into(self.beta, self.batch_size, st["idx"], st["w"])

    def _idx_offset(self):
        return self.memory.first_leaf_index

    def learn(self):
        result = super().learn()
        result["beta"] = self.beta
        return result

    def process(self, transitions, step):
        """per.py:89-122."""
        result = {}
        self._store(transitions)
        delta_t = step - self.time_t
        self.time_t = step
        self.target_update_stamp += delta_t
        self.learn_period_stamp += delta_t
        self.beta = min(1.0, self.beta + (self.beta_add * delta_t))
        if self.learn_period_stamp >= self.learn_period and self.memory.size >= self.batch_size and self.time_t >= self.start_train_step:
            result = self.learn()
            if self.lr_decay:
                self.learning_rate_decay(step)
            self.learn_period_stamp -= self.learn_period
        if self.num_learn > 0:
            self.epsilon_decay(delta_t)
            if self.target_update_stamp >= self.target_update_period:
                self.update_target()
                self.target_update_stamp -= self.target_update_period
        return result


class ApeX(DQN):
    """core/agent/ape_x.py:11-199: n-step double-Q, PER, grad clipping, per-actor epsilon and
    actor-side initial priorities |G_n - q_t|."""

    _td = dict(double=True, per=True, n_step=1)

    def __init__(self, epsilon=0.4, epsilon_alpha=7.0, clip_grad_norm=40.0, alpha=0.6, beta=0.4, learn_period=4,
                 uniform_sample_prob=1e-3, n_step=4, **kwargs):
        super().__init__(**kwargs)
        self.epsilon = epsilon
        self.epsilon_alpha = epsilon_alpha
        self.clip_grad_norm = clip_grad_norm
        self.num_transitions = 0
        self.alpha = alpha
        self.beta = beta
        self.learn_period = learn_period
        self.learn_period_stamp = 0
        self.uniform_sample_prob = uniform_sample_prob
        self.beta_add = (1 - beta) / self.run_step
        self.n_step = n_step
        self.memory = PERBuffer(self.buffer_size, uniform_sample_prob, device=self.device, frame_dedup=self._frame_dedup)
        self.memory.defer_rows = 16  # per-step stores coalesce into one ring append before the next learn()
        self.tmp_buffer = deque(maxlen=n_step + 1)

    @torch.no_grad()
    def act(self, state, training=True):
        self.network.train(training)
        epsilon = self.epsilon if training else self.epsilon_eval
        q = self.network(self.as_tensor(state))
        if np.random.random() < epsilon:
            batch_size = state[0].shape[0] if isinstance(state, list) else state.shape[0]
            action = np.random.randint(0, self.action_size, size=(batch_size, 1))
        else:
            action = torch.argmax(q, -1, keepdim=True).cpu().numpy()
        q = np.take(q.cpu().numpy(), action)
        return {"action": action, "q": q}

    def _draw(self, st):
        return self.memory.sample_into(self.beta, self.batch_size, st["idx"], st["w"])

    def _idx_offset(self):
        return self.memory.first_leaf_index

    def learn(self):
        r = super().learn()
        return {"loss": r["loss"], "max_Q": r["max_Q"], "sampled_p": r["sampled_p"], "mean_p": r["mean_p"],
                "num_learn": self.num_learn, "num_transitions": self.num_transitions}

    def process(self, transitions, step):
        """ape_x.py:135-164.  transitions=None: the actors publish into the staging ring (`memory.make_ring`,
        jh_ring_*) instead of handing lists over; take whatever has arrived."""
        result = {}
        delta_t = step - self.time_t
        if transitions is None:
            self.num_transitions += self.memory.drain()
        else:
            self.num_transitions += len(next(iter(transitions.values()))) if isinstance(transitions, dict) else len(transitions)
            self._store(transitions)
        self.time_t = step
        self.target_update_stamp += delta_t
        self.learn_period_stamp += delta_t
        self.beta = min(1.0, self.beta + (self.beta_add * delta_t))
        if self.learn_period_stamp >= self.learn_period and self.memory.buffer_counter >= self.batch_size and self.time_t >= self.start_train_step:
            result = self.learn()
            if self.lr_decay:
                self.learning_rate_decay(step)
            self.learn_period_stamp -= self.learn_period
        if self.num_learn > 0 and self.target_update_stamp >= self.target_update_period:
            self.update_target()
            self.target_update_stamp -= self.target_update_period
        return result

    def set_distributed(self, id):
        assert self.num_workers > 1
        self.epsilon = self.epsilon ** (1 + (id / (self.num_workers - 1)) * self.epsilon_alpha)
        return self

    def interact_callback(self, transition):
        """ape_x.py:174-199."""
        _transition = {}
        self.tmp_buffer.append(transition)
        if len(self.tmp_buffer) == self.tmp_buffer.maxlen:
            _transition["state"] = self.tmp_buffer[0]["state"]
            _transition["action"] = self.tmp_buffer[0]["action"]
            _transition["next_state"] = self.tmp_buffer[-1]["state"]
            for key in self.tmp_buffer[0].keys():
                if key not in ["state", "action", "next_state"]:
                    _transition[key] = np.stack([t[key] for t in self.tmp_buffer][:-1], axis=1)
            target_q = self.tmp_buffer[-1]["q"]
            for i in reversed(range(self.n_step)):
                target_q = self.tmp_buffer[i]["reward"] + (1 - self.tmp_buffer[i]["done"]) * self.gamma * target_q
            _transition["priority"] = abs(target_q - self.tmp_buffer[0]["q"])
            del _transition["q"]
        return _transition
