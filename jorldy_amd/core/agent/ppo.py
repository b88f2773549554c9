import os

import numpy as np
import torch

from ... import np_rng, ops
from ..buffer import RolloutBuffer
from ..network import Network
from ..optimizer import Optimizer
from .base import BaseAgent


MAX_HEAD_OUTPUTS = 40  # jh_pponet_create (csrc/jh_mlp.hip: kMaxHeadOutputs): continuous A <= 19 (Humanoid: 17), discrete A <= 39
NATIVE_ELIGIBLE = ("PPO runs on libjorldy_hip only: network in {discrete_policy_value, continuous_policy_value}, head='mlp', int state_size, "
                   "hidden_size % 16 == 0, optim_config name 'adam' without weight_decay / amsgrad, and action_size + 1 (discrete) or "
                   f"2 * action_size + 1 (continuous) <= {MAX_HEAD_OUTPUTS} head outputs (config.ppo.cartpole, config.ppo.mujoco on all its envs; up to 8 outputs "
                   "-- CartPole, Hopper -- on the four-launch minibatch update and the persistent acting kernel, beyond that on the separate calls / the tiled engine); "
                   "head='cnn' with a (C, H, W) state_size and network 'discrete_policy_value' runs on the convolutional engine (core/agent/ppo_cnn.py: config.ppo.atari, ppo.procgen)")


class PPO(BaseAgent):
    """core/agent/ppo.py:10-202 (+ the REINFORCE base, reinforce.py:14-142) with the learner-side hot
    path on HIP kernels:

      rollout -> GPU SoA store (pinned staging)            RolloutBuffer        ppo.py:72-74
      log pi_old, GAE scan, per-row standardisation        jh_logp_*, jh_gae    ppo.py:83-110
      minibatch gathers x[idx] + clipped surrogate +       jh_ppo_loss_*        ppo.py:122-165
        clipped value + entropy, forward AND backward
      MLP encoder fwd/bwd (fp32 MFMA), clip_grad_norm_,    jh_pponet_*          ppo.py:127-169
        Adam on flat buckets            [backend "native"]
      the 5 `.item()` syncs per minibatch                  one D2H of a [n_updates, 8] stats array

    Everything above is hand-written kernels; the whole learn() is captured in one hipGraph after the first call.
    There is ONE backend (libjorldy_hip): head="mlp", an int state_size, Adam without weight decay / amsgrad and <= 40 head
    outputs -- what configs ppo.cartpole / ppo.mujoco (Hopper, HalfCheetah, Walker, Ant, Humanoid) use.  Any other configuration raises (NATIVE_ELIGIBLE below) instead of
    silently switching to library kernels; the reference agent keeps working for those.
    Constructor arguments, `act`, `process`, result keys, checkpoint format are the reference's.
    """

    def __new__(cls, *args, **kwargs):
        # head="cnn" (config.ppo.atari / ppo.procgen): the same agent on the convolutional engine (ppo_cnn.PPOConv, a subclass: its __init__ runs)
        if cls is PPO and kwargs.get("head", args[4] if len(args) > 4 else "mlp") == "cnn":
            from .ppo_cnn import PPOConv

            return object.__new__(PPOConv)
        return object.__new__(cls)

    def __init__(self, state_size, action_size, hidden_size=512, network="discrete_policy_value", head="mlp",
                 optim_config={"name": "adam"}, gamma=0.99, use_standardization=True, run_step=1e6, lr_decay=True,
                 device=None, batch_size=32, n_step=128, n_epoch=3, _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0,
                 ent_coef=0.01, clip_grad_norm=1.0, num_workers=1, backend=None, use_graph=True, seed=0, **kwargs):
        self.device = self._require_gpu(device)
        self.action_type = network.split("_")[0]
        assert self.action_type in ["continuous", "discrete"]
        if backend not in (None, "auto", "native"):
            raise ValueError(f"backend={backend!r}: jorldy_amd has one backend (libjorldy_hip); the torch mirror of round 1-3 is test infrastructure now (tests/mirror)")
        # what the library's policy-value net covers, checked BEFORE anything is built: an unsupported configuration says so at construction
        # with the eligible list (the parameter containers of core/network only know the hot path's heads)
        shape_ok = (head == "mlp" and isinstance(state_size, (int, np.integer)) and network in ("discrete_policy_value", "continuous_policy_value")
                    and hidden_size % 16 == 0 and optim_config.get("name", "adam").lower() == "adam"
                    and (2 * action_size + 1 if self.action_type == "continuous" else action_size + 1) <= MAX_HEAD_OUTPUTS)
        if not shape_ok:
            raise ValueError(f"{NATIVE_ELIGIBLE}; got network={network!r}, head={head!r}, state_size={state_size!r}, hidden_size={hidden_size}, "
                             f"optim_config={optim_config!r}, action_size={action_size}")
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
        self.grad_sync = None  # data-parallel hook: jorldy_amd.parallel.BucketSync (RCCL all-reduce)
        # data-parallel learners: the critic's max(mean, mean) (ppo.py:147-154) taken over the GLOBAL minibatch (one more 8-byte all-reduce per
        # minibatch, between loss and backward) = exactly one learner on the concatenated batch.  JH_DP_EXACT_CRITIC=0: every rank's own max
        # (round 3's form: equal only while the value clamp is inactive)
        self.dp_exact_critic = os.environ.get("JH_DP_EXACT_CRITIC", "1") == "1"

        eligible = (
            head == "mlp" and isinstance(state_size, (int, np.integer)) and optim_config.get("name", "adam").lower() == "adam"
            and network in ("discrete_policy_value", "continuous_policy_value") and hidden_size % 16 == 0
            and not self.optimizer.defaults.get("amsgrad", False) and self.optimizer.defaults.get("weight_decay", 0) == 0
            and (2 * action_size + 1 if self.action_type == "continuous" else action_size + 1) <= MAX_HEAD_OUTPUTS
        )
        if not eligible:
            raise ValueError(f"{NATIVE_ELIGIBLE}; got network={network!r}, head={head!r}, state_size={state_size!r}, hidden_size={hidden_size}, "
                             f"optim_config={optim_config!r}, action_size={action_size}")
        self.backend = "native"
        self.use_graph = use_graph
        # four- / five-launch minibatch update (jh_pponet_ppo_update) for minibatches < 1024 rows; JH_FUSED_UPDATE=0
        # keeps the forward / loss / backward / Adam calls separate (same results; used by the A/B in bench)
        self.fused_update = os.environ.get("JH_FUSED_UPDATE", "1") == "1"
        # capture the RCCL all-reduce of the data-parallel path inside the hipGraph too (falls back to
        # eager launches if the capture is refused)
        self.graph_with_collective = os.environ.get("JH_GRAPH_DP", "1") == "1"
        self._net = None
        self._graph = None
        self._static = None
        self._adam_steps = 0
        self._captured = 0            # rows whose heads / values the collector delivered for the next learn() (jh_collector_set_capture)
        self._post_launch_hook = None  # one-shot callable run right behind learn()'s launches (NativeCollector.arm_prelaunch)
        self._lr_step = None          # process(): step for the lr decay that learn() applies behind its launches
        # a NativeCollector offers its commit launch for the small per-learn() uploads (index lists, learning rate): _ride = that collector,
        # _ride_done = a run has happened since the last learn(), _ride_wait = what was handed to it and is not known to have arrived
        self._ride, self._ride_done, self._ride_wait = None, False, {}
        self._lr_word = None
        # index lists of the next learn() drawn ahead on a copy of np.random's state (np_rng.Predraw); JH_PPO_PREDRAW=0: draw inside learn()
        self._predraw = np_rng.Predraw() if os.environ.get("JH_PPO_PREDRAW", "1") == "1" else None
        self._init_native(int(state_size), int(action_size), int(hidden_size), seed)

    # ---------------------------------------------------------------------------------- native engine
    def _init_native(self, S, A, H, seed, max_rows=4096):
        cont = self.action_type == "continuous"
        self._seed = seed
        net = ops.PPONet(S, H, A, cont, max_rows, self.device, seed=seed)
        params = list(self.network.parameters())
        assert sum(p.numel() for p in params) == net.n_params, "state_dict layout mismatch with libjorldy_hip"
        o = 0
        with torch.no_grad():
            for p in params:  # registration order == the reference's state_dict order == the flat layout
                n = p.numel()
                net.params[o : o + n].copy_(p.reshape(-1))
                p.data = net.params[o : o + n].view_as(p)       # nn.Parameters become views of the bucket
                p.grad = net.grads[o : o + n].view_as(p)
                o += n
        self._views = [(p, net.m[a : a + p.numel()].view_as(p), net.v[a : a + p.numel()].view_as(p))
                       for p, a in zip(params, np.cumsum([0] + [q.numel() for q in params[:-1]]))]
        d = self.optimizer.defaults
        net.set_hyper(d["lr"], d["betas"][0], d["betas"][1], d["eps"], step=0.0)
        self._net = net

    def _grow_native(self, rows):
        if rows <= self._net.max_rows:
            return
        old = self._net
        S, H, A = old.S, old.H, old.A
        net = ops.PPONet(S, H, A, old.cont, max(rows, 2 * old.max_rows), self.device, seed=self._seed)
        for dst, src in ((net.params, old.params), (net.grads, old.grads), (net.m, old.m), (net.v, old.v)):
            dst.copy_(src)
        o = 0
        for p in self.network.parameters():
            n = p.numel()
            p.data = net.params[o : o + n].view_as(p)
            p.grad = net.grads[o : o + n].view_as(p)
            o += n
        params = list(self.network.parameters())
        self._views = [(p, net.m[a : a + p.numel()].view_as(p), net.v[a : a + p.numel()].view_as(p))
                       for p, a in zip(params, np.cumsum([0] + [q.numel() for q in params[:-1]]))]
        d = self.optimizer.defaults
        net.set_hyper(self.optimizer.param_groups[0]["lr"], d["betas"][0], d["betas"][1], d["eps"], step=float(self._adam_steps))
        torch.cuda.synchronize()
        self._drop_rides()
        self._net, self._graph, self._static, self._graphs = net, None, None, {}

    # ---------------------------------------------------------------------------------- act
    @torch.no_grad()
    def act(self, state, training=True):
        """ppo.py:55-69 on the native acting kernels (sampling included: Philox stream `seed`)."""
        if isinstance(state, list):
            raise NotImplementedError("PPO: list-valued (multimodal) observations are outside the native policy-value net")
        self._grow_native(len(state))
        obs = np.asarray(state, dtype=np.float32)
        return {"action": self._net.act_discrete(obs, training) if self.action_type == "discrete" else self._net.act_continuous(obs, training)}

    # ---------------------------------------------------------------------------------- learn
    def learn(self):
        return self._learn_native()

    def _result(self, s, n_upd):
        return {
            "actor_loss": np.mean(s[:n_upd, 1]),
            "critic_loss": np.mean(s[:n_upd, 2]),
            "entropy_loss": np.mean(s[:n_upd, 3]),
            "max_ratio": float(s[:n_upd, 4].max()),
            "min_prob": float(s[:n_upd, 5].min()),
            "mean_ret": float(s[n_upd, 0]),
        }

    # -- native: static buffers so that the whole update sequence can be replayed as one hipGraph ----
    def _alloc_static(self, M):
        S, A = self._net.S, self._net.A
        cont = self._net.cont
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=self.device)
        n_mb = (M + self.batch_size - 1) // self.batch_size
        n_upd = self.n_epoch * n_mb
        B = self.batch_size
        # [state; next_state] and their heads are adjacent so that the two no-grad passes of ppo.py:83-94 are ONE
        # forward over 2M rows when that fits the latency-oriented kernel
        x_all, h0_all, v_all = f(2 * M, S), f(2 * M, A), f(2 * M, 1)
        h1_all = f(2 * M, A) if cont else None
        st = dict(
            M=M, n_upd=n_upd, x_all=x_all, h0_all=h0_all, h1_all=h1_all, v_all=v_all,
            tr={"state": x_all[:M], "action": f(M, A if cont else 1), "reward": f(M, 1), "next_state": x_all[M:], "done": f(M, 1)},
            arange=torch.arange(M, dtype=torch.int64, device=self.device),
            idx=torch.zeros(self.n_epoch * M, dtype=torch.int64, device=self.device),
            h0=h0_all[:M], h1=h1_all[:M] if cont else None, value=v_all[:M], nh0=h0_all[M:], nh1=h1_all[M:] if cont else None, next_value=v_all[M:],
            logp_old=f(M, A if cont else 1), adv=f(M, 1), ret=f(M, 1),
            mb_h0=f(B, A), mb_h1=f(B, A) if cont else None, mb_v=f(B, 1),
            stats=torch.zeros(n_upd + 1, 8, dtype=torch.float32, device=self.device),
        )
        if os.environ.get("JH_PPO_MAPPED_STATS", "1") == "1":
            # the [n_upd + 1][8] statistics live in device-MAPPED pinned host memory: the loss kernels write them across
            # PCIe themselves (32 bytes per update, flushed when the kernel ends), and learn() waits for the last row to
            # arrive instead of for the stream -- no D2H copy, no hipStreamSynchronize wake-up, and the host is already
            # preparing the next rollout while the last backward + Adam are still running (stream order keeps that correct)
            pin = ops.PinnedBuffer((n_upd + 1, 8), np.float32, self.device.index)
            pin.np[:] = 0.0
            st["stats_pin"] = pin
            st["stats"] = ops._wrap_device(pin.dev_ptr.value, (n_upd + 1, 8), torch.float32, self.device, owner=pin)
        # rows of every minibatch of every epoch, gathered once per learn() (jh_ppo_minibatch_rows) when all
        # minibatches take the four- / five-launch update
        E = self.n_epoch
        if self.fused_update and all(self._net.fused_ok(min(B, M - o)) for o in range(0, M, B)):
            srcs = [st["tr"]["state"], st["tr"]["action"], st["adv"], st["ret"], st["value"], st["logp_old"]]
            st["mb"] = [f(E * M, int(t.numel() // M)) for t in srcs]
            st["rows"] = ops.MinibatchRows(srcs, st["mb"])
        # the epochs' index lists are drawn straight into pinned memory (two buffers: the next learn()'s lists are drawn and
        # uploaded while this one's may still be in flight)
        st["idx_pin"] = [ops.PinnedBuffer((self.n_epoch * M,), np.int64, self.device.index) for _ in range(2)]  # device-mapped: a kernel may read them in place
        st["idx_alias"] = [ops._wrap_device(p.dev_ptr.value, (self.n_epoch * M,), torch.int64, self.device, owner=p) for p in st["idx_pin"]]
        st["idx_ev"] = [None, None]
        st["idx_k"] = 0
        st["idx_ready"] = False  # st["idx"] already holds the (pre-drawn, uploaded) lists of the coming learn()
        self._stats = st["stats"]
        return st

    def _capture_targets(self, M):
        """Device tensors the collector's acting-time capture writes for a rollout of M rows (raw policy head(s), V(s), V(s'))."""
        self._grow_native(2 * M if 2 * M <= 8192 else M)
        if self._static is None or self._static["M"] != M:
            self._drop_rides()
            self._static, self._graphs = self._alloc_static(M), {}
        st = self._static
        return st["h0"], st["h1"], st["value"], st["next_value"]

    def _drop_rides(self):
        """The static buffers (index lists) or the network (hyper block) are about to be replaced: whatever was handed to the collector's
        commit launch and has not been delivered must not be delivered any more (dangling ride-along pointers)."""
        if self._ride is not None:
            self._ride.clear_rides()
        self._ride_wait.clear()
        if self._static is not None:
            self._static["idx_ready"] = False

    def _upload_idx(self, st, draw, ride=False):
        """draw(numpy int64 view [E * M]) fills the lists in device-mapped pinned memory; they reach st["idx"] with one copy kernel now,
        or (ride) inside the commit launch of the collector's next run."""
        k = st["idx_k"]
        st["idx_k"] = 1 - k
        if st["idx_ev"][k] is not None:
            st["idx_ev"][k].synchronize()
            st["idx_ev"][k] = None
        ok = draw(st["idx_pin"][k].np)
        if ok is False:
            return False
        if ride and self._ride is not None:
            self._ride.ride_along(0, st["idx_pin"][k].dev_ptr.value, st["idx"].data_ptr(), st["idx"].numel() * 8, keep=(st["idx_pin"][k], st["idx"]))
            self._ride_wait["idx"] = k
            return True
        self._ride_wait.pop("idx", None)
        st["idx"].copy_(st["idx_alias"][k])
        ev = torch.cuda.Event()
        ev.record()
        st["idx_ev"][k] = ev
        return True

    def _enqueue_learn(self, st):
        """Everything between `memory.sample()` and the result read-back, as stream work only (no host
        sync, no allocation outside torch's graph-private pool): capturable."""
        self._enqueue_pre(st)
        self._enqueue_main(st)

    def _enqueue_pre(self, st, captured=False):
        """ppo.py:83-112: everything that does NOT depend on the epoch shuffles (replay rows, the no-grad passes,
        log pi_old, GAE, mean return).  Launched first so that the host's `np.random.shuffle` calls (ppo.py:118, ~12 us
        each at 1024 rows) run while the GPU is busy with this."""
        net, M = self._net, st["M"]
        cont = net.cont
        tr = st["tr"]
        self.memory._store.gather(st["arange"], as_float=True, out={k: tr[k] for k in tr})
        if captured:
            pass  # st["h0"] / ["h1"] / ["value"] / ["next_value"] were delivered with the rollout (acting-time capture): no forward pass
        elif 2 * M <= min(net.max_rows, 8192):
            net.forward(st["x_all"], out=(st["h0_all"], st["h1_all"], st["v_all"]))
        else:
            net.forward(tr["next_state"], out=(st["nh0"], st["nh1"], st["next_value"]))
            net.forward(tr["state"], out=(st["h0"], st["h1"], st["value"]))
        if cont:
            logp_old = ops.logp_continuous(st["h0"], st["h1"], tr["action"], out=st["logp_old"])
        else:
            logp_old = ops.logp_discrete(st["h0"], tr["action"], out=st["logp_old"])
        adv, ret = ops.gae(tr["reward"], tr["done"], st["value"], st["next_value"], self.n_step, self.gamma, self._lambda, self.use_standardization,
                           out=(st["adv"], st["ret"]))
        ops.mean_into(ret, st["stats"][st["n_upd"], 0:1])  # ppo.py:112

    def _enqueue_main(self, st):
        """ppo.py:114-169: the minibatch updates of all epochs (needs st["idx"] = the shuffles)."""
        net, M, B = self._net, st["M"], self.batch_size
        cont = net.cont
        tr, adv, ret, logp_old = st["tr"], st["adv"], st["ret"], st["logp_old"]
        k = 0
        if "rows" in st:
            # x[idx] of every epoch in one launch, then forward + loss + backward (+ clip + Adam) in 4-5 launches per
            # minibatch on consecutive rows (jh_pponet_ppo_update)
            xs, acts, advs, rets, vals, lps = st["rows"](st["idx"])
            exact = self.grad_sync is not None and self.dp_exact_critic
            if exact and "critic_sums" not in st:
                st["critic_sums"] = torch.zeros(st["n_upd"], 2, dtype=torch.float32, device=self.device)
            for e in range(self.n_epoch):
                for offset in range(0, M, B):
                    o0, o1 = e * M + offset, e * M + min(offset + B, M)
                    if exact:  # critic = max(mean(e1), mean(e2)) over the GLOBAL minibatch: 8 more bytes on the wire, before the backward
                        net.ppo_update_dp(xs[o0:o1], None, acts[o0:o1], advs[o0:o1], rets[o0:o1], vals[o0:o1], lps[o0:o1], self.epsilon_clip, self.vf_coef,
                                          self.ent_coef, st["stats"][k], self.grad_sync.reduce_flat, st["critic_sums"][k],
                                          peer=getattr(getattr(self.grad_sync, "transport", None), "peer", None))
                    else:
                        net.ppo_update(xs[o0:o1], None, acts[o0:o1], advs[o0:o1], rets[o0:o1], vals[o0:o1], lps[o0:o1], self.epsilon_clip, self.vf_coef,
                                       self.ent_coef, self.clip_grad_norm, st["stats"][k], do_adam=self.grad_sync is None)
                    if self.grad_sync is not None:
                        self.grad_sync.reduce_flat(net.grads)
                        net.adam_step(self.clip_grad_norm)
                    k += 1
            return
        exact = self.grad_sync is not None and self.dp_exact_critic
        if exact and "dp_work" not in st:
            st["dp_work"] = torch.zeros(st["n_upd"], B + 16, dtype=torch.float32, device=self.device)
        # round 6: one call per update with the loss in ONE launch whatever the minibatch size (jh_pponet_ppo_update_rows); JH_PPO_ONEPASS=0 keeps the
        # separate forward / two-pass loss / backward / Adam calls (A/B switch, bit-identical)
        onepass = not exact and os.environ.get("JH_PPO_ONEPASS", "1") == "1"
        for e in range(self.n_epoch):
            for offset in range(0, M, B):
                b = min(B, M - offset)
                idx = st["idx"][e * M + offset : e * M + offset + b]
                if onepass:
                    net.ppo_update_rows(tr["state"], idx, tr["action"], adv, ret, st["value"], logp_old, self.epsilon_clip, self.vf_coef, self.ent_coef, self.clip_grad_norm,
                                        st["stats"][k], do_adam=self.grad_sync is None)
                    if self.grad_sync is not None:
                        self.grad_sync.reduce_flat(net.grads)
                        net.adam_step(self.clip_grad_norm)
                    k += 1
                    continue
                if cont:
                    mu, ls, vp = net.forward(tr["state"], idx=idx, out=(st["mb_h0"][:b], st["mb_h1"][:b], st["mb_v"][:b]))
                    if exact:
                        g_mu, g_ls, g_v = ops.ppo_loss_dp(mu, ls, vp, idx, tr["action"], adv, ret, st["value"], logp_old, self.epsilon_clip, self.vf_coef, self.ent_coef,
                                                          st["stats"][k], self.grad_sync.reduce_flat, st["dp_work"][k])
                    else:
                        g_mu, g_ls, g_v, _ = ops.ppo_loss_continuous(mu, ls, vp, idx, tr["action"], adv, ret, st["value"], logp_old, self.epsilon_clip, self.vf_coef, self.ent_coef, stats=st["stats"][k])
                    net.backward(tr["state"], idx, g_mu, g_ls, g_v)
                else:
                    z, vp = net.forward(tr["state"], idx=idx, out=(st["mb_h0"][:b], None, st["mb_v"][:b]))
                    if exact:
                        g_z, g_v = ops.ppo_loss_dp(z, None, vp, idx, tr["action"], adv, ret, st["value"], logp_old, self.epsilon_clip, self.vf_coef, self.ent_coef,
                                                   st["stats"][k], self.grad_sync.reduce_flat, st["dp_work"][k])
                    else:
                        g_z, g_v, _ = ops.ppo_loss_discrete(z, vp, idx, tr["action"], adv, ret, st["value"], logp_old, self.epsilon_clip, self.vf_coef, self.ent_coef, stats=st["stats"][k])
                    net.backward(tr["state"], idx, g_z, None, g_v)
                if self.grad_sync is not None:
                    self.grad_sync.reduce_flat(net.grads)
                net.adam_step(self.clip_grad_norm)
                k += 1

    def _learn_native(self):
        self._learn_launch()
        return self._learn_finish()

    def early_ready(self):
        """Is the coming learn() a pure enqueue (its whole-learn graph captured, index lists drawn ahead)?  Then a collector may take
        the begin / process_begin / loop / process_end form; before that, process() (capturing a graph synchronises the device, which
        must not happen while a rollout's acting kernel is waiting for this thread's observations)."""
        st = self._static
        return bool(st is not None and st["idx_ready"] and ("one", True) in getattr(self, "_graphs", {}) and self._predraw is not None and self._predraw.valid)

    def _learn_launch(self, allow_capture=True):
        """Enqueue one learn() (everything up to and including the launches); `_learn_finish` does the host work behind them and reads
        the statistics.  Split so that a collector can enqueue the learner BEFORE its rollout's host loop (NativeCollector.begin / loop:
        the launches wait on the stream behind the acting kernel and the gated commit, and start the instant the rollout ends)."""
        M = self.memory.size
        self._grow_native(2 * M if 2 * M <= 8192 else M)
        if self._static is None or self._static["M"] != M:
            self._drop_rides()
            self._static, self._graphs = self._alloc_static(M), {}
        st = self._static
        E = self.n_epoch
        captured = self._captured == M
        self._captured = 0
        # what was handed to the collector's commit launch has arrived iff a run happened since; otherwise deliver it now
        rode, self._ride_done = self._ride_done, False
        if self._ride_wait and not rode:
            if "lr" in self._ride_wait:
                self._net.set_lr(float(self._lr_word.np[0]))
            if "idx" in self._ride_wait and st["idx_ready"]:
                st["idx"].copy_(st["idx_alias"][self._ride_wait["idx"]])
        self._ride_wait.clear()

        def shuffles():
            # the reference's global-RNG shuffles (ppo.py:118) of all epochs, drawn by numpy's own algorithm on numpy's own state
            # (np_rng.epoch_shuffles) straight into pinned memory, uploaded at once
            self._upload_idx(st, lambda out: np_rng.epoch_shuffles(M, E, out))

        pin = st.get("stats_pin")
        if pin is not None:  # arrival markers: one element of EACH 16-byte granule of the last update's row (critic and c2: means of squares, never -1)
            pin.np[st["n_upd"] - 1, 2] = -1.0
            pin.np[st["n_upd"] - 1, 7] = -1.0
        graphable = (self.use_graph and not ops._PROF["on"] and not ops._PROF["lib"] and not getattr(self, "_graph_failed", False)
                     and (self.grad_sync is None or (self.graph_with_collective and getattr(self.grad_sync, "capturable", True))))
        # index lists: pre-drawn behind the previous learn() and already uploaded (np_rng.Predraw) when np.random has not been
        # touched since -- then ONE graph holds the whole learn(); otherwise drawn now, between the pre-phase and the minibatch
        # graph (the GPU works on the no-grad passes / GAE while the host shuffles)
        have_idx = bool(st["idx_ready"]) and self._predraw is not None and self._predraw.commit(M, E)
        st["idx_ready"] = False
        split = not have_idx and os.environ.get("JH_PPO_SPLIT_GRAPH", "1") == "1"
        key = ("split" if split else "one", captured)
        graphs = getattr(self, "_graphs", None)
        if graphs is None:
            graphs = self._graphs = {}
        warm = getattr(self, "_warm_keys", None)
        if warm is None:
            warm = self._warm_keys = set()
        if graphable and allow_capture and key not in graphs and key in warm:
            try:
                torch.cuda.synchronize()
                if split:
                    gp = torch.cuda.CUDAGraph()
                    with ops.graph_capture(gp):
                        self._enqueue_pre(st, captured)
                    if "main" not in graphs:
                        g = torch.cuda.CUDAGraph()
                        with ops.graph_capture(g):  # thread_local: other threads (batched actors, staging ring) keep issuing HIP work on their own streams
                            self._enqueue_main(st)
                        graphs["main"] = g
                    graphs[key] = gp
                else:
                    g = torch.cuda.CUDAGraph()
                    with ops.graph_capture(g):
                        self._enqueue_pre(st, captured)
                        self._enqueue_main(st)
                    graphs[key] = g  # capture does not execute: replay below runs this iteration's update
            except Exception as e:  # e.g. a collective that cannot be captured: stay eager from now on
                graphs.clear()
                self._graph_failed, graphable = True, False
                torch.cuda.synchronize()
                import traceback

                print(f"[jorldy_amd] hipGraph capture of learn() failed ({type(e).__name__}: {e}); running eagerly\n{traceback.format_exc()}")
        if graphable and key in graphs:
            if split:
                graphs[key].replay()  # the GPU works on the no-grad passes / GAE ...
                shuffles()            # ... while the host shuffles
                graphs["main"].replay()
            else:
                if not have_idx:
                    shuffles()
                graphs[key].replay()
            self._graph = graphs.get("main", graphs[key])
        else:
            self._enqueue_pre(st, captured)
            if not have_idx:
                shuffles()
            self._enqueue_main(st)
            warm.add(key)
            self._warm = True
        self.memory._store.clear()
        self._adam_steps += st["n_upd"]
        self._launched = (st, pin, M, E)

    def _learn_finish(self):
        st, pin, M, E = self._launched
        self._launched = None
        # ---- behind the launches, while the GPU works: next learning rate, next index lists, the next rollout's acting kernel
        if self._lr_step is not None:
            if self.lr_decay:
                self.learning_rate_decay(self._lr_step)
            self._lr_step = None
        if self._predraw is not None:
            st["idx_ready"] = bool(self._upload_idx(st, lambda out: self._predraw.draw(M, E, out), ride=True))
        hook, self._post_launch_hook = self._post_launch_hook, None
        if hook is not None:
            hook()
        if pin is not None:
            s = self._await_mapped_stats(pin.np, st["n_upd"])
        else:
            s = self._read_stats(st["stats"])[0].astype(np.float64)  # the only host sync of learn()
        return self._result(s, st["n_upd"])

    def _await_mapped_stats(self, a, n_upd):
        """Wait until the last loss kernel's row has landed in the mapped host buffer (BaseAgent._await_marks)."""
        self._await_marks(a, ((n_upd - 1) * 8 + 2, (n_upd - 1) * 8 + 7), "PPO.learn()")  # jh_ppo.hip: jh_ppo_stats_row
        return a.astype(np.float64)

    def learning_rate_decay(self, step, optimizers=None, mode="cosine"):
        super().learning_rate_decay(step, optimizers, mode)
        if self._net is not None:
            lr = self.optimizer.param_groups[0]["lr"]
            if self._ride is not None and self._lr_step is not None:  # from inside learn(): the next run's commit launch delivers it
                if self._lr_word is None:
                    self._lr_word = ops.PinnedBuffer((1,), np.float32, self.device.index)
                self._lr_word.np[0] = lr
                self._ride.ride_along(1, self._lr_word.dev_ptr.value, self._net.hyper_ptr(), 4, keep=(self._lr_word, self._net))
                self._ride_wait["lr"] = True
            else:  # set directly (load(), an external schedule): nothing older may be delivered on top of it later
                if self._ride_wait.pop("lr", None) and self._ride is not None:
                    self._ride.ride_along(1, 0, 0, 0)
                self._net.set_lr(lr)

    def process_begin(self, step):
        """First half of `process(None, step)` for a rollout whose rows are already committed to the store -- possibly only ENQUEUED
        (NativeCollector.begin): the bookkeeping of ppo.py:187-202 and, when a learn() is due, its launches.  `process_end` returns the result."""
        assert self._net is not None, "process_begin / process_end are the native backend's split form of process()"
        delta_t = step - self.time_t
        self.time_t = step
        self.learn_stamp += delta_t
        self._begun = self.learn_stamp >= self.n_step
        if self._begun:
            self._lr_step = step
            self._learn_launch(allow_capture=False)  # a capture synchronises the device: not while the acting kernel waits for this thread

    def process_end(self):
        if not getattr(self, "_begun", False):
            return {}
        self._begun = False
        result = self._learn_finish()
        self.learn_stamp = 0
        return result

    def process(self, transitions, step):
        """ppo.py:187-202.  `transitions` is the reference's List[Dict] or an SoA dict of arrays."""
        result = {}
        if transitions is None:
            pass  # a native collector already appended them to the rollout store
        elif isinstance(transitions, dict):
            self.memory.store_soa(transitions)
        else:
            self.memory.store(transitions)
        delta_t = step - self.time_t
        self.time_t = step
        self.learn_stamp += delta_t
        if self.learn_stamp >= self.n_step:
            self._lr_step = step  # learn() applies the decay itself, right behind its launches (ppo.py:199-200)
            result = self.learn()
            self.learn_stamp = 0
        return result

    # ---------------------------------------------------------------------------------- checkpoint
    def _resume_extra_attrs(self):
        """The native sampling stream (seed, acting-step counter): a resumed run continues the same action stream."""
        if self._net is None:
            return {}
        seed, ctr = self._net.act_rng()
        return {"act_seed": str(seed), "act_ctr": str(ctr)}  # strings: uint64 does not survive JSON floats

    def _resume_load_extra_attrs(self, d):
        if self._net is not None and "act_seed" in d:
            self._net.act_rng((int(d["act_seed"]), int(d["act_ctr"])))

    def _export_optim_state(self):
        """Native Adam moments -> torch.optim.Adam state, so `ckpt` keeps the reference's format
        ({"network": state_dict, "optimizer": state_dict}, reinforce.py:128-136)."""
        if self._net is None:
            return
        for p, m, v in self._views:
            self.optimizer.state[p] = {"step": torch.tensor(float(self._adam_steps)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}

    def _import_optim_state(self):
        if self._net is None:
            return
        steps = 0
        for p, m, v in self._views:
            stt = self.optimizer.state.get(p)
            if stt:
                m.copy_(stt["exp_avg"])
                v.copy_(stt["exp_avg_sq"])
                steps = int(float(stt["step"]))
        self._adam_steps = steps
        d = self.optimizer.defaults
        if self._ride_wait.pop("lr", None) and self._ride is not None:  # a learning rate still riding with the next commit launch would overwrite this one
            self._ride.ride_along(1, 0, 0, 0)
        self._net.set_hyper(self.optimizer.param_groups[0]["lr"], d["betas"][0], d["betas"][1], d["eps"], step=float(steps))

    def save(self, path):
        print(f"...Save model to {path}...")
        self._export_optim_state()
        torch.save({"network": self.network.state_dict(), "optimizer": self.optimizer.state_dict()}, os.path.join(path, "ckpt"))

    def load(self, path):
        print(f"...Load model from {path}...")
        checkpoint = torch.load(os.path.join(path, "ckpt"), map_location=self.device, weights_only=False)
        self.network.load_state_dict(checkpoint["network"])
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        self._import_optim_state()
