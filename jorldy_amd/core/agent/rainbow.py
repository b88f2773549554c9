import os
from collections import deque

import numpy as np
import torch

from ... import ops
from ..buffer import PERBuffer
from ..network import Network
from .dqn import DQN
from .native_net import require_native


class C51(DQN):
    """core/agent/c51.py:11-135: categorical DQN; the target net selects its own greedy action,
    1-step projection, plain mean cross-entropy."""

    def __init__(self, state_size, action_size, v_min=-10, v_max=10, num_support=51, **kwargs):
        super().__init__(state_size, action_size * num_support, **kwargs)
        self.action_size = action_size
        self.v_min, self.v_max, self.num_support = v_min, v_max, num_support
        self.delta_z = (v_max - v_min) / (num_support - 1)
        self.z = torch.linspace(v_min, v_max, num_support, device=self.device).view(1, -1)
        self._stats8, self._stats8_np = self._mapped_stats(8)

    def logits2Q(self, logits):
        _logits = logits.view(logits.shape[0], self.action_size, self.num_support)
        _logits = _logits - torch.max(_logits, -1, keepdim=True).values
        p_logit = torch.exp(torch.log_softmax(_logits, dim=-1))
        return p_logit, torch.sum(self.z.view(1, 1, -1) * p_logit, dim=-1)

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
                _, q_action = self.logits2Q(self.network(self.as_tensor(state)))
                action = torch.argmax(q_action, -1, keepdim=True).cpu().numpy()
        return {"action": action}

    def _learn_body(self, st):
        B, A, K = self.batch_size, self.action_size, self.num_support
        net = self._net  # q-network with A*K outputs on the native engine
        tr = self.memory.gather(st["idx"], as_float=self._as_float(), out=st["tr"])
        lg = net.learn_forward(st["x_all"], B, None, st["logits"])
        g, _, _, _ = ops.c51_loss(lg[0].view(B, A, K), lg[2].view(B, A, K), tr["action"], tr["reward"], tr["done"], self.v_min, self.v_max, self.gamma,
                                  shift_max=True, stats=self._stats8)
        net.backward(g.view(B, A * K))
        if self.grad_sync is not None:
            self.grad_sync.reduce_flat(net.grads)
        net.optim_step(self._opt_name, self.clip_grad_norm)

    def learn(self):
        s, _ = self._learn_stats(self._stats8_np, (5, 7), self._stats8)
        return {"loss": float(s[0]), "epsilon": self.epsilon, "max_Q": float(s[1]), "max_logit": float(s[2]), "min_logit": float(s[3])}


class Rainbow(DQN):
    """core/agent/rainbow.py:14-308: noisy dueling categorical net, n-step double-Q projection, PER
    with priorities KL^alpha.  The projection + KL + backward-to-logits + priorities are one HIP
    kernel (jh_c51_loss); priorities go straight into the device sum tree (no B `.item()` syncs).

    The network itself runs on libjorldy_hip (ops.RainbowNet; factorised or independent noise, mlp / cnn head, plain Adam /
    RMSprop): convolutions as implicit GEMMs on the fp32 MFMA reading the uint8 frames straight out of the replay store, the
    three forwards of learn() share their launches, backward + Adam are native; one hipGraph per learn().  One backend: any
    other configuration raises (native_net.NATIVE_ELIGIBLE)."""

    def __init__(self, state_size, action_size, hidden_size=512, network="rainbow", head="mlp",
                 optim_config={"name": "adam"}, gamma=0.99, buffer_size=50000, batch_size=64, start_train_step=2000,
                 target_update_period=500, run_step=1e6, lr_decay=True, n_step=4, alpha=0.6, beta=0.4, learn_period=4,
                 uniform_sample_prob=1e-3, noise_type="factorized", v_min=-10, v_max=10, num_support=51, device=None,
                 use_graph=True, backend=None, frame_dedup=False, **kwargs):
        self.device = self._require_gpu(device)
        self.use_graph = use_graph
        self.grad_sync = None  # data-parallel hook (jorldy_amd.parallel.attach_data_parallel)
        self.graph_with_collective = os.environ.get("JH_GRAPH_DP", "1") == "1"
        self._static, self._graph, self._warm, self.clip_grad_norm = None, None, False, None
        self._overlap = os.environ.get("JH_LEARN_OVERLAP", "0") == "1"  # opt-in graph branches (dqn.py: measured slower): noise sets || trunk, PER write-back || backward
        self._fused_step = os.environ.get("JH_RB_FUSED_STEP", "1") == "1"  # jh_rbnet_c51_step + deferred backward tails (0: the separate calls, an A/B switch)
        self._td = dict(double=True, per=True, n_step=1)
        self.action_size = action_size
        self.action_type = "discrete"
        require_native(backend, network, head, state_size, hidden_size, optim_config, noise_type)
        self.backend = "native"
        self._net = None
        self._init_native(network, state_size, action_size, num_support, hidden_size, head, batch_size, optim_config,
                          Network(network, state_size, action_size, num_support, noise_type, D_hidden=hidden_size, head=head), noise_type=noise_type)
        self.gamma = gamma
        self.batch_size = batch_size
        self.start_train_step = start_train_step
        self.target_update_stamp = 0
        self.target_update_period = target_update_period
        self.num_learn = 0
        self.time_t = 0
        self.run_step = run_step
        self.lr_decay = lr_decay
        self.n_step = n_step
        self.tmp_buffer = deque(maxlen=n_step)
        self.alpha = alpha
        self.beta = beta
        self.learn_period = learn_period
        self.learn_period_stamp = 0
        self.uniform_sample_prob = uniform_sample_prob
        self.beta_add = (1 - beta) / run_step
        self.v_min, self.v_max, self.num_support = v_min, v_max, num_support
        self.memory = PERBuffer(buffer_size, uniform_sample_prob, device=self.device, frame_dedup=frame_dedup)
        self.memory.defer_rows = 16  # per-step stores coalesce into one ring append before the next learn()
        self.delta_z = (v_max - v_min) / (num_support - 1)
        self.z = torch.linspace(v_min, v_max, num_support, device=self.device).view(1, -1)
        self.epsilon = 0.0
        self._stats8, self._stats8_np = self._mapped_stats(8)
        self._stats, self._stats_np = self._mapped_stats(4)
        self._noise = None  # parity tests inject the Gaussian draws here (a list forces the eager path; "static" = already in st["noise"], graph allowed)

    def logits2Q(self, logits):
        _logits = logits.view(logits.shape[0], self.action_size, self.num_support)
        p_logit = torch.exp(torch.log_softmax(_logits, dim=-1))
        return p_logit, torch.sum(self.z.view(1, 1, -1) * p_logit, dim=-1)

    @torch.no_grad()
    def act(self, state, training=True):
        self.network.train(training)
        if training and self.memory.size < max(self.batch_size, self.start_train_step):
            batch_size = state[0].shape[0] if isinstance(state, list) else state.shape[0]
            action = np.random.randint(0, self.action_size, size=(batch_size, 1))
        else:
            action = self._act_greedy(state, training)
            if action is None:
                _, q_action = self.logits2Q(self.network(self.as_tensor(state), training))
                action = torch.argmax(q_action, -1, keepdim=True).cpu().numpy()
        return {"action": action}

    def _draw(self, st):
        return self.memory.sample_into(self.beta, self.batch_size, st["idx"], st["w"])

    def _idx_offset(self):
        return self.memory.first_leaf_index

    # ---- plumbing: native_net.NativeValueNetMixin via DQN -------------------
    def _draw_noise(self, st):
        """Three independent draws: network(s), network(s'), target_network(s') (rainbow.py:160-186) -> st["noise"]."""
        if self._noise is None:
            if getattr(self, "_normal", None) is None:
                self._normal = ops.NormalSource(self.device)
            self._normal.fill(st["noise"])
        elif self._noise != "static":  # "static": the test wrote the draws into st["noise"] itself (replayable: nothing to do here)
            for i in range(3):
                self.network.pack_noise(self._noise[i], st["noise"][i])

    def _learn_body(self, st):
        net, B = self._net, self.batch_size
        if self._overlap:
            # branch 1: the noise draw and the three noisy weight sets (normal fill 5 us + a 12-us launch) beside gather + trunk
            main, side = self._fork()
            with torch.cuda.stream(side):
                self._draw_noise(st)
                net.prepare_noise(st["noise"])
        else:
            self._draw_noise(st)
        tr = self.memory.gather(st["idx"], idx_offset=self.memory.first_leaf_index, as_float=self._as_float(), out=st["tr"])
        net.learn_trunk(st["x_all"], B)
        if self._overlap:
            main.wait_stream(side)
        if self._fused_step and not self._overlap and B <= 1024:
            # rainbow.py:160-235 in three launches (jh_rbnet_c51_step): the dueling combine of the three forwards and the gradient back through it
            # live inside the loss kernel, the priorities' leaf write-back (rainbow.py:230-231) inside the statistics kernel, then the climb
            net.learn_heads_raw(B, st["noise"])
            self.memory.flush()
            net.c51_step(self.memory._tree, st["idx"], tr["action"], tr["reward"], tr["done"], st["w"], self.v_min, self.v_max, self.gamma, self.alpha,
                         self.n_step, st["logits"], stats=self._stats8)
            # d(sigma) = d(mu) * eps and conv1's partial sums ride in the optimizer's pass unless somebody reads the bucket first
            net.backward(None, defer=self.grad_sync is None)
            if self.grad_sync is not None:
                self.grad_sync.reduce_flat(net.grads)
            net.optim_step(self._opt_name, self.clip_grad_norm)
            return
        lg = net.learn_heads(B, st["noise"], st["logits"])
        g, prio, _, _ = ops.c51_loss(lg[0], lg[2], tr["action"], tr["reward"], tr["done"], self.v_min, self.v_max, self.gamma,
                                     next_logit_online=lg[1], weights=st["w"], alpha=self.alpha, n_step=self.n_step, stats=self._stats8)
        side = self._write_back_priorities(st, prio)  # rainbow.py:230-231; branch 2: beside the backward pass + optimizer
        net.backward(g)
        if self.grad_sync is not None:  # data-parallel learners: one all-reduce of the flat gradient bucket
            self.grad_sync.reduce_flat(net.grads)
        net.optim_step(self._opt_name, self.clip_grad_norm)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    def _resume_extra_attrs(self):
        """The learner's noise stream (ops.NormalSource: seed + call counter in device memory): without it a resumed run would
        re-seed from torch's generator and draw other NoisyNet noise than the uninterrupted one."""
        src = getattr(self, "_normal", None)
        if src is None:
            return {}
        st = src.state.cpu().tolist()
        return {"normal_source": [str(int(v)) for v in st]}

    def _resume_load_extra_attrs(self, d):
        if "normal_source" in d and self._net is not None:
            self._normal = ops.NormalSource(self.device, seed=0)
            self._normal.state.copy_(torch.tensor([int(v) for v in d["normal_source"]], dtype=torch.int64))

    def learn(self):
        s, p = self._learn_stats(self._stats8_np, (5, 7), self._stats8)
        return {"loss": float(s[0]), "beta": self.beta, "max_Q": float(s[1]), "max_logit": float(s[2]), "min_logit": float(s[3]),
                "sampled_p": float(p[0]), "mean_p": float(p[1])}

    def process(self, transitions, step):
        """rainbow.py:255-283."""
        result = {}
        delta_t = step - self.time_t
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

    def interact_callback(self, transition):
        """rainbow.py:294-308."""
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
