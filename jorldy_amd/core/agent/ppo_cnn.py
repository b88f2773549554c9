import os

import numpy as np
import torch

from ... import _lib as L
from ... import np_rng, ops
from ..buffer import RolloutBuffer
from ..network import Network
from ..optimizer import Optimizer
from .native_net import NativeNet, NativeValueNetMixin
from .ppo import PPO

CNN_ELIGIBLE = ("PPO on the CNN head runs on libjorldy_hip's convolutional value-network engine (jh_rbnet_*): network 'discrete_policy_value', head 'cnn', "
                "state_size (C, H, W) with H, W >= 36, hidden_size % 4 == 0, optim_config {'name': 'adam', lr, betas, eps} (config.ppo.atari, config.ppo.procgen)")


class PolicyValueNet(NativeNet):
    """`agent.network` of the policy-value net on the CNN head: state_dict in the reference's keys (head.conv*, l, pi, v: policy_value.py:8-22) over the
    library's flat bucket; calling it returns (pi, v) like the reference module's forward."""

    @torch.no_grad()
    def __call__(self, x):
        net = self._net
        x = x.contiguous()
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.to(torch.float32)
        outs = [net.forward(x[o : o + net.maxB], self._which, None) for o in range(0, x.shape[0], net.maxB)]
        out = (outs[0] if len(outs) == 1 else torch.cat(outs, 0)).view(-1, net.A)
        n = net.n_actions
        return torch.softmax(out[:, :n], dim=-1), out[:, n:].clone()


class PPOConv(NativeValueNetMixin, PPO):
    """core/agent/ppo.py:10-202 for `head="cnn"` (config/ppo/atari.py, config/ppo/procgen.py; network/head.py:21-61 under policy_value.py:8-22).

    The network is conv 8x8/4 -> conv 4x4/2 -> conv 3x3/1 -> Linear(F, hidden) -> (pi | v): the q-network of the DQN family with one more row in its
    last layer, so it runs on the SAME engine (jh_rbnet kind q with A + 1 outputs: implicit-GEMM convolutions on uint8 frames, grouped fp32-MFMA GEMMs,
    native backward, clip_grad_norm_ + Adam in the optimizer launch).  Around it, per learn():

      rollout frames stay uint8 in HBM (RolloutBuffer)                         ppo.py:72-74
      no-grad passes over [state; next_state] in slabs of `forward_rows`       jh_rbnet_forward + jh_heads_unpack      ppo.py:83-94
      log pi_old, GAE, mean return                                             jh_logp_discrete, jh_gae, jh_mean_f32   ppo.py:92-112
      per minibatch: frame rows x[idx] (uint8 gather) -> forward (kept)        jh_store_gather, jh_rbnet_forward_keep  ppo.py:118-135
        -> clipped surrogate + clipped value + entropy, fwd AND bwd            jh_ppo_loss_packed                      ppo.py:137-165
        -> backward -> clip_grad_norm_ + Adam                                  jh_rbnet_backward, jh_rbnet_optim_step  ppo.py:167-169
      the `.item()` statistics                                                 one read of a [n_updates + 1, 8] array

    act() (ppo.py:55-69): frames -> pinned slab -> H2D -> forward -> jh_policy_act_discrete (inverse-CDF sampling on the device, the counter-based stream of
    the MLP policy's sampler) -> actions in device-mapped memory.  Constructor arguments, result keys and the checkpoint format are the reference's."""

    def __init__(self, state_size, action_size, hidden_size=512, network="discrete_policy_value", head="cnn", optim_config={"name": "adam"}, gamma=0.99,
                 use_standardization=True, run_step=1e6, lr_decay=True, device=None, batch_size=32, n_step=128, n_epoch=3, _lambda=0.95, epsilon_clip=0.1,
                 vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, num_workers=1, backend=None, use_graph=True, seed=0, forward_rows=256, **kwargs):
        self.device = self._require_gpu(device)
        if backend not in (None, "auto", "native"):
            raise ValueError(f"backend={backend!r}: jorldy_amd has one backend (libjorldy_hip)")
        ok = (network == "discrete_policy_value" and head == "cnn" and not np.isscalar(state_size) and len(state_size) == 3
              and all(np.isscalar(v) for v in state_size) and min(state_size[1:]) >= 36 and hidden_size % 4 == 0
              and optim_config.get("name", "adam").lower() == "adam" and set(optim_config) <= {"name", "lr", "betas", "eps"})
        if not ok:
            raise ValueError(f"{CNN_ELIGIBLE}; got network={network!r}, head={head!r}, state_size={state_size!r}, hidden_size={hidden_size}, optim_config={optim_config!r}")
        self.action_type = "discrete"
        self.state_size, self.action_size = [int(v) for v in state_size], int(action_size)
        self.gamma, self.use_standardization, self.run_step, self.lr_decay = gamma, use_standardization, run_step, lr_decay
        self.batch_size, self.n_step, self.n_epoch, self._lambda = batch_size, n_step, n_epoch, _lambda
        self.epsilon_clip, self.vf_coef, self.ent_coef, self.clip_grad_norm, self.num_workers = epsilon_clip, vf_coef, ent_coef, clip_grad_norm, num_workers
        self.time_t = self.learn_stamp = 0
        self.memory = RolloutBuffer(device=self.device)
        self.grad_sync = None
        self.dp_exact_critic = os.environ.get("JH_DP_EXACT_CRITIC", "1") == "1"
        self.backend, self.use_graph = "native", use_graph
        self.graph_with_collective = os.environ.get("JH_GRAPH_DP", "1") == "1"
        self._graph, self._graphs, self._static, self._stats = None, {}, None, None
        self._seed, self._act_ctr = int(seed), 0
        self._ride, self._ride_wait = None, {}  # (PPO._upload_idx: nothing rides on a collector's commit launch here)
        # the reference's initialisation (orthogonal, utils.py:110-124) drawn by the reference's module on the host, then moved into the library's bucket
        torch_net = Network(network, self.state_size, self.action_size, D_hidden=hidden_size, head=head)
        self._net = ops.RainbowNet(self.state_size, self.action_size, 1, hidden_size, "cnn", max(int(batch_size), int(forward_rows)), self.device, kind="pv")
        self._net.import_state(torch_net.state_dict())
        self.network = PolicyValueNet(self._net, 0)
        self._optim_config = dict(optim_config)
        self._opt_name = "adam"
        d = Optimizer(**optim_config, params=[torch.nn.Parameter(torch.zeros(1))]).defaults
        self._lr0, self._lr_now, self._adam_steps = float(d["lr"]), float(d["lr"]), 0
        self._set_native_hyper(d, 0)
        self.optimizer = None

    # ---------------------------------------------------------------------------------- act
    @torch.no_grad()
    def act(self, state, training=True):
        if isinstance(state, list):
            raise NotImplementedError("PPO: list-valued (multimodal) observations are outside the native policy-value net")
        if torch.is_tensor(state):  # the reference's as_tensor (base.py:61-73) takes tensors too
            state = state.detach().cpu().numpy()
        x = np.asarray(state)
        if x.dtype != np.uint8:
            x = x.astype(np.float32, copy=False)
        N, net = int(x.shape[0]), self._net
        key = (N, tuple(x.shape[1:]), x.dtype == np.uint8)
        a = self.__dict__.get("_actbuf")
        if a is None or a["key"] != key:
            dt = torch.uint8 if key[2] else torch.float32
            am = ops.PinnedBuffer((N,), np.int64, self.device.index)
            a = dict(key=key, x_pin=torch.empty((N,) + key[1], dtype=dt, pin_memory=True), x_dev=torch.empty((N,) + key[1], dtype=dt, device=self.device),
                     heads=torch.empty(N, net.A, dtype=torch.float32, device=self.device), am=am,
                     act_dev=ops._wrap_device(am.dev_ptr.value, (N,), torch.int64, self.device, owner=am), words=am.np.view(np.uint32),
                     marks=np.arange(0, 2 * N, 2, dtype=np.int32))
            self._actbuf = a
        np.copyto(a["x_pin"].numpy(), x)
        a["x_dev"].copy_(a["x_pin"], non_blocking=True)
        for o in range(0, N, net.maxB):
            net.forward(a["x_dev"][o : o + net.maxB], 0, None, out=a["heads"][o : o + net.maxB])
        a["am"].np[:] = -1  # arrival marks (actions are >= 0)
        ops.policy_act_discrete(a["heads"], net.n_actions, self._seed, self._act_ctr, training, a["act_dev"])
        self._act_ctr += 1
        if L.load().jh_host_wait_words(L.ptr(a["words"]), L.ptr(a["marks"]), N, 0xFFFFFFFF, 5.0) != 0:
            torch.cuda.current_stream(self.device).synchronize()
            if (a["am"].np < 0).any():
                raise RuntimeError("act(): the actions never arrived (failed launch?)")
        return {"action": a["am"].np.reshape(N, 1).copy()}

    # ---------------------------------------------------------------------------------- learn
    def _alloc_static(self, M):
        net, E, B = self._net, self.n_epoch, self.batch_size
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=self.device)
        n_upd = E * ((M + B - 1) // B)
        probe = self.memory._store.gather(torch.zeros(1, dtype=torch.int64, device=self.device), names=["state"], as_float=False)["state"]
        frame, dt = tuple(probe.shape[1:]), probe.dtype
        x_all = torch.empty((2 * M,) + frame, dtype=dt, device=self.device)
        st = dict(M=M, n_upd=n_upd, x_all=x_all, tr={"state": x_all[:M], "next_state": x_all[M:], "action": f(M, 1), "reward": f(M, 1), "done": f(M, 1)},
                  arange=torch.arange(M, dtype=torch.int64, device=self.device), idx=torch.zeros(E * M, dtype=torch.int64, device=self.device),
                  packed=f(2 * M, net.A), h0=f(2 * M, net.n_actions), v=f(2 * M), logp_old=f(M, 1), adv=f(M, 1), ret=f(M, 1),
                  x_mb=torch.empty((B,) + frame, dtype=dt, device=self.device), heads_mb=f(B, net.A), grad_mb=torch.zeros(B, net.A, dtype=torch.float32, device=self.device),
                  stats=torch.zeros(n_upd + 1, 8, dtype=torch.float32, device=self.device), dp_work=torch.zeros(n_upd, B + 16, dtype=torch.float32, device=self.device))
        st["idx_pin"] = [ops.PinnedBuffer((E * M,), np.int64, self.device.index) for _ in range(2)]
        st["idx_alias"] = [ops._wrap_device(p.dev_ptr.value, (E * M,), torch.int64, self.device, owner=p) for p in st["idx_pin"]]
        st["idx_ev"], st["idx_k"], st["idx_ready"] = [None, None], 0, False
        self._stats = st["stats"]
        return st

    def _enqueue_pre(self, st, captured=False):
        """ppo.py:83-112: the rollout's rows, the two no-grad passes, log pi_old, GAE, mean return."""
        net, M, tr = self._net, st["M"], st["tr"]
        self.memory._store.gather(st["arange"], as_float={"state": False, "next_state": False}, out={k: tr[k] for k in tr})
        for o in range(0, 2 * M, net.maxB):
            net.forward(st["x_all"][o : o + net.maxB], 0, None, out=st["packed"][o : o + net.maxB])
        ops.heads_unpack(st["packed"], net.n_actions, st["h0"], None, st["v"])
        ops.logp_discrete(st["h0"][:M], tr["action"], out=st["logp_old"])
        ops.gae(tr["reward"], tr["done"], st["v"][:M], st["v"][M:], self.n_step, self.gamma, self._lambda, self.use_standardization, out=(st["adv"], st["ret"]))
        ops.mean_into(st["ret"], st["stats"][st["n_upd"], 0:1])

    def _enqueue_main(self, st):
        """ppo.py:114-169: the minibatch updates of all epochs (st["idx"] = the epochs' shuffles)."""
        net, M, B, tr = self._net, st["M"], self.batch_size, st["tr"]
        exact = self.grad_sync is not None and self.dp_exact_critic
        k = 0
        for e in range(self.n_epoch):
            for offset in range(0, M, B):
                b = min(B, M - offset)
                idx = st["idx"][e * M + offset : e * M + offset + b]
                x = st["x_mb"][:b]
                self.memory._store.gather(idx, names=["state"], as_float=False, out={"state": x})
                heads, grad = st["heads_mb"][:b], st["grad_mb"][:b]
                net.forward_keep(x, heads)
                ops.ppo_loss_packed(heads, net.n_actions, idx, tr["action"], st["adv"], st["ret"], st["v"][:M], st["logp_old"], self.epsilon_clip, self.vf_coef, self.ent_coef,
                                    grad, st["stats"][k], reduce_mean=self.grad_sync.reduce_flat if exact else None, work=st["dp_work"][k])
                net.backward(grad)
                if self.grad_sync is not None:
                    self.grad_sync.reduce_flat(net.grads)
                net.optim_step("adam", max_norm=self.clip_grad_norm)
                k += 1

    def learn(self):
        M, E = self.memory.size, self.n_epoch
        if self._static is None or self._static["M"] != M:
            self._static, self._graphs = self._alloc_static(M), {}
        st = self._static
        if st.get("store") is not self.memory._store:  # the rollout store was replaced (it grows by doubling): a captured graph holds the old one's addresses
            st["store"], self._graphs, self._graph = self.memory._store, {}, None
        graphable = (self.use_graph and not ops._PROF["on"] and not ops._PROF["lib"] and not getattr(self, "_graph_failed", False)
                     and (self.grad_sync is None or (self.graph_with_collective and getattr(self.grad_sync, "capturable", True))))
        if graphable and "learn" not in self._graphs and getattr(self, "_warm", False):
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with ops.graph_capture(g):
                    self._enqueue_pre(st)
                    self._enqueue_main(st)
                self._graphs["learn"] = g
            except Exception as e:
                self._graphs.clear()
                self._graph_failed, graphable = True, False
                torch.cuda.synchronize()
                print(f"[jorldy_amd] hipGraph capture of PPO(cnn).learn() failed ({type(e).__name__}: {e}); running eagerly")
        # the reference's global-RNG shuffles (ppo.py:118) of all epochs, drawn before the launches (nothing else touches np.random inside learn())
        self._upload_idx(st, lambda out: np_rng.epoch_shuffles(M, E, out))
        if graphable and "learn" in self._graphs:
            self._graphs["learn"].replay()
            self._graph = self._graphs["learn"]
        else:
            self._enqueue_pre(st)
            self._enqueue_main(st)
            self._warm = True
        self.memory._store.clear()
        self._adam_steps += st["n_upd"]
        s = self._read_stats(st["stats"])[0].astype(np.float64)  # the only host sync of learn()
        return self._result(s, st["n_upd"])

    def _drop_rides(self):
        pass

    def learning_rate_decay(self, step, optimizers=None, mode="cosine"):
        self._native_lr_decay(step, mode)

    def process(self, transitions, step):
        """ppo.py:187-202."""
        result = {}
        if transitions is None:
            pass
        elif isinstance(transitions, dict):
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

    def process_begin(self, step):
        raise NotImplementedError("the split process() is the MLP policy's (NativeCollector); PPO on the CNN head takes process()")

    process_end = process_begin

    def early_ready(self):
        return False

    # ---------------------------------------------------------------------------------- checkpoint
    def _resume_extra_attrs(self):
        return {"act_seed": str(self._seed), "act_ctr": str(self._act_ctr)}

    def _resume_load_extra_attrs(self, d):
        if "act_seed" in d:
            self._seed, self._act_ctr = int(d["act_seed"]), int(d["act_ctr"])

    def save(self, path):
        self._native_save(path)

    def load(self, path):
        self.target_network = self.network  # (_native_load mirrors the online weights into the family's target network: there is none here)
        try:
            self._native_load(path)
        finally:
            del self.target_network
