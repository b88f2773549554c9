"""Native value-network backend shared by the DQN family and Rainbow: the network lives in ops.RainbowNet's
flat buckets (libjorldy_hip jh_rbnet_*: implicit-GEMM convolutions, grouped MFMA GEMMs, native backward and
optimizer step); this module keeps the agents' nn.Module-shaped surface (`agent.network(...)`, state_dict,
checkpoints in the reference's format, lr decay, target sync) on top of it."""
import os

import numpy as np
import torch

from ... import _lib as L
from ... import ops
from ..optimizer import Optimizer

_KIND_OF = {"rainbow": "rainbow", "dueling": "dueling", "discrete_q_network": "q"}


def native_supported(network, head, state_size, hidden_size, optim_config, noise_type="factorized"):
    """Can this configuration run on libjorldy_hip's value networks?"""
    name = optim_config.get("name", "adam").lower()
    if name == "adam":
        ok_opt = set(optim_config) <= {"name", "lr", "betas", "eps"}
    elif name == "rmsprop":
        ok_opt = set(optim_config) <= {"name", "lr", "alpha", "eps", "centered"}
    else:
        ok_opt = False
    ok_state = (head == "mlp" and np.isscalar(state_size)) or (
        head == "cnn" and not np.isscalar(state_size) and len(state_size) == 3 and all(np.isscalar(v) for v in state_size))
    return (network in _KIND_OF and (network != "rainbow" or noise_type in ("factorized", "independent")) and ok_state and hidden_size % 4 == 0 and ok_opt)


NATIVE_ELIGIBLE = ("the DQN family / Rainbow run on libjorldy_hip only: network in {discrete_q_network, dueling, rainbow (factorized | independent noise)}, "
                   "head 'mlp' with a scalar state_size or head 'cnn' with a (C, H, W) state_size, hidden_size % 4 == 0, optim_config "
                   "{'name': 'adam', lr, betas, eps} or {'name': 'rmsprop', lr, alpha, eps, centered} "
                   "(config.dqn / double / multistep / per / c51 / rainbow / ape_x x cartpole / atari and their shapes)")


def require_native(backend, network, head, state_size, hidden_size, optim_config, noise_type="factorized"):
    """One backend: raise with the list of eligible configurations instead of switching libraries."""
    if backend not in (None, "auto", "native"):
        raise ValueError(f"backend={backend!r}: jorldy_amd has one backend (libjorldy_hip); the torch mirror of rounds 1-3 is test infrastructure now (tests/mirror)")
    if not native_supported(network, head, state_size, hidden_size, optim_config, noise_type):
        raise ValueError(f"{NATIVE_ELIGIBLE}; got network={network!r}, head={head!r}, state_size={state_size!r}, hidden_size={hidden_size}, "
                         f"noise_type={noise_type!r}, optim_config={optim_config!r}")


class NativeNet:
    """What the agent code expects from `agent.network` / `agent.target_network` (an nn.Module) on top of
    ops.RainbowNet's flat buckets: state_dict in the reference's keys / shapes, callable forward."""

    def __init__(self, net, which):
        self._net, self._which, self.training = net, which, True

    def _bucket(self):
        return self._net.params if self._which == 0 else self._net.target

    def state_dict(self):
        return self._net.export_state(self._bucket())

    def load_state_dict(self, sd, strict=True):
        self._net.import_state(sd, self._bucket())

    def parameters(self):
        return list(self.state_dict().values())

    def named_parameters(self):
        return list(self.state_dict().items())

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *args, **kwargs):
        return self

    def pack_noise(self, noise, out=None):
        """{tag: (e_in, e_out)} / {tag: (eps_w [in, out], eps_b)} (the parity fixtures' injection format for factorised /
        independent noise) -> one flat noise set."""
        flat = torch.cat([torch.cat([noise[t][0].reshape(-1), noise[t][1].reshape(-1)]) for t in ("a1", "v1", "a2", "v2")]).to(self._net.device, torch.float32)
        assert flat.numel() == self._net.noise_len
        if out is not None:
            out.copy_(flat)
            return out
        return flat

    @torch.no_grad()
    def __call__(self, x, is_train=False, noise=None):
        net = self._net
        x = x.contiguous()
        nz = None
        if net.kind == "rainbow" and is_train:
            nz = self.pack_noise(noise) if noise is not None else torch.randn(net.noise_len, device=net.device)
        outs = [net.forward(x[o : o + net.maxB], self._which, nz) for o in range(0, x.shape[0], net.maxB)]  # one noise draw per call, like the reference
        out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        return out if net.kind == "rainbow" else out.view(out.shape[0], net.A)  # q-values [rows, A]


class NativeValueNetMixin:
    """Agent-side plumbing of the native backend (state kept on the agent: _net, _opt_name, _optim_config,
    _lr0, _lr_now, _adam_steps)."""

    def _init_native(self, network, state_size, action_size, num_support, hidden_size, head, batch_size, optim_config, torch_net, noise_type="factorized"):
        self._net = ops.RainbowNet(state_size, action_size, num_support, hidden_size, head, batch_size, self.device, kind=_KIND_OF[network], noise_type=noise_type)
        self._net.import_state(torch_net.state_dict())  # the reference's initialisation (orthogonal / uniform, utils.py:89-124)
        self._net.sync_target()
        self.network, self.target_network = NativeNet(self._net, 0), NativeNet(self._net, 1)
        self._optim_config = dict(optim_config)
        self._opt_name = optim_config.get("name", "adam").lower()
        d = Optimizer(**optim_config, params=[torch.nn.Parameter(torch.zeros(1))]).defaults
        self._lr0, self._lr_now, self._adam_steps = float(d["lr"]), float(d["lr"]), 0
        self._set_native_hyper(d, 0)
        self.optimizer = None

    def as_tensor(self, x):
        """base.py:61-73 turns everything into fp32 on the device; uint8 frames headed for the native CNN stay
        uint8 (the first convolution's operand fetch divides by 255 itself)."""
        if getattr(self, "_net", None) is not None and self._net.cnn and isinstance(x, np.ndarray) and x.dtype == np.uint8:
            return torch.as_tensor(x, device=self.device)
        return super().as_tensor(x)

    @torch.no_grad()
    def _act_greedy(self, state, is_train=False):
        """The network branch of act() (dqn.py:100-115, rainbow.py:140-152) without torch in the loop: observations -> pinned slab -> one
        async H2D -> forward of the live online network -> jh_value_act (Q from the outputs, first-maximum argmax) writing the actions
        into device-mapped memory -> the host waits for their arrival (jh_host_wait_words).  The generic form -- as_tensor (a pageable,
        synchronous copy), five torch kernels for logits2Q + argmax, `.cpu()` -- was most of a single-mode env step.
        Rainbow in training mode draws its noise exactly as NativeNet.__call__ does (torch.randn on the device: the same stream of
        draws).  -> int64 [N, 1]; None when this form does not apply (list-valued states, more rows than the network's batch)."""
        net = getattr(self, "_net", None)
        # the reference's as_tensor (base.py:61-73) takes torch tensors and any dtype: only ndarrays that copy into the slab without a lossy
        # cast come this way, everything else takes the generic path (a CUDA tensor or a float64 observation raised here)
        if net is None or not isinstance(state, np.ndarray) or state.ndim < 2:
            return None
        x = state
        if not (x.dtype == np.uint8 and net.cnn) and not np.can_cast(x.dtype, np.float32, casting="same_kind"):
            return None
        N = int(x.shape[0])
        if N < 1 or N > net.maxB:
            return None
        # [rows, actions, atoms] as the agent reads the outputs (C51 keeps its A x K outputs in a plain q-network: c51.py:27-31)
        A, K = int(self.action_size), int(getattr(self, "num_support", 1))
        if A * K != net.A * net.K:
            return None
        u8 = bool(net.cnn and x.dtype == np.uint8)
        key = (N, tuple(x.shape[1:]), u8)
        a = self.__dict__.get("_actbuf")
        if a is None or a["key"] != key:
            dt = torch.uint8 if u8 else torch.float32
            am, qm = ops.PinnedBuffer((N,), np.int64, self.device.index), ops.PinnedBuffer((N,), np.float32, self.device.index)
            a = dict(key=key, x_pin=torch.empty((N,) + tuple(x.shape[1:]), dtype=dt, pin_memory=True), x_dev=torch.empty((N,) + tuple(x.shape[1:]), dtype=dt, device=self.device),
                     logits=torch.empty(N, net.A, net.K, dtype=torch.float32, device=self.device), am=am, qm=qm,  # (viewed [N, A, K] for the act kernel)
                     act_dev=ops._wrap_device(am.dev_ptr.value, (N,), torch.int64, self.device, owner=am),
                     q_dev=ops._wrap_device(qm.dev_ptr.value, (N,), torch.float32, self.device, owner=qm), words=am.np.view(np.uint32),
                     marks=np.arange(0, 2 * N, 2, dtype=np.int32))  # low words of the int64 actions
            self._actbuf = a
        np.copyto(a["x_pin"].numpy(), x, casting="same_kind")
        a["x_dev"].copy_(a["x_pin"], non_blocking=True)
        nz = torch.randn(net.noise_len, device=net.device) if (net.kind == "rainbow" and is_train) else None
        net.forward(a["x_dev"], 0, nz, out=a["logits"])
        a["am"].np[:] = -1  # arrival marks (actions are >= 0)
        ops.value_act(a["logits"].view(N, A, K), float(getattr(self, "v_min", 0.0)), float(getattr(self, "v_max", 0.0)), out=(a["act_dev"], a["q_dev"]))
        if L.load().jh_host_wait_words(L.ptr(a["words"]), L.ptr(a["marks"]), N, 0xFFFFFFFF, 5.0) != 0:
            torch.cuda.current_stream(self.device).synchronize()
            if (a["am"].np < 0).any():
                raise RuntimeError("act(): the actions never arrived (failed launch?)")
        return a["am"].np.reshape(N, 1).copy()

    def _set_native_hyper(self, d, steps):
        if self._opt_name == "adam":
            self._net.set_hyper(d["lr"], d["betas"][0], d["betas"][1], d["eps"], steps)
        else:
            self._net.set_hyper(d["lr"], d["alpha"], 0.0, d["eps"], steps, centered=d["centered"])

    def _as_float(self):
        # frames stay uint8 until the first convolution's operand fetch; everything else fp32 (as_tensor)
        return {"state": False, "next_state": False} if (self._net is not None and self._net.cnn) else True

    def _alloc_static_native(self):
        B = self.batch_size
        idx = torch.zeros(B, dtype=torch.int64, device=self.device)
        probe = self.memory.gather(idx, idx_offset=0, as_float=self._as_float())
        x_all = torch.empty((2 * B,) + tuple(probe["state"].shape[1:]), dtype=probe["state"].dtype, device=self.device)
        tr = dict(probe)
        tr["state"], tr["next_state"] = x_all[:B], x_all[B:]  # one contiguous [state; next_state] batch
        st = dict(idx=idx, w=torch.ones(B, dtype=torch.float32, device=self.device), tr=tr, store=self.memory._store, x_all=x_all,
                  logits=torch.empty(3, B, self._net.A, self._net.K, dtype=torch.float32, device=self.device))
        if self._net.kind == "rainbow":
            st["noise"] = torch.zeros(3, self._net.noise_len, dtype=torch.float32, device=self.device)
        return st

    def _native_lr_decay(self, step, mode="cosine"):
        self._lr_now = self._lr0 * float(self._lr_weight(step, mode))
        self._net.set_lr(self._lr_now)  # a device scalar: the captured graph reads it

    def _shadow_optimizer(self):
        """The configured torch optimizer over copies of the parameters carrying the native moments: the
        reference's ckpt format ({"network", "optimizer"}, dqn.py:184-199) both ways."""
        sd = self._net.export_state()
        params = [torch.nn.Parameter(v) for v in sd.values()]
        opt = Optimizer(**self._optim_config, params=params)
        for grp in opt.param_groups:
            grp["lr"] = self._lr_now
        return opt, params, list(sd.keys())

    def _native_save(self, path):
        print(f"...Save model to {path}...")
        opt, params, keys = self._shadow_optimizer()
        if self._adam_steps > 0:
            m, v = self._net.export_state(self._net.m), self._net.export_state(self._net.v)
            for p, k in zip(params, keys):
                stt = {"step": torch.tensor(float(self._adam_steps))}
                if self._opt_name == "adam":
                    stt.update(exp_avg=m[k], exp_avg_sq=v[k])
                else:
                    stt["square_avg"] = v[k]
                    if opt.defaults["centered"]:
                        stt["grad_avg"] = m[k]
                opt.state[p] = stt
        torch.save({"network": self.network.state_dict(), "optimizer": opt.state_dict()}, os.path.join(path, "ckpt"))

    def _native_load(self, path):
        print(f"...Load model from {path}...")
        checkpoint = torch.load(os.path.join(path, "ckpt"), map_location=self.device, weights_only=False)
        self.network.load_state_dict(checkpoint["network"])
        self.target_network.load_state_dict(checkpoint["network"])
        opt, params, keys = self._shadow_optimizer()
        opt.load_state_dict(checkpoint["optimizer"])
        steps = 0
        self._net.m.zero_()
        self._net.v.zero_()
        if opt.state:
            km, kv = ("exp_avg", "exp_avg_sq") if self._opt_name == "adam" else ("grad_avg", "square_avg")
            if km in opt.state[params[0]]:
                self._net.import_state({k: opt.state[p][km] for p, k in zip(params, keys)}, self._net.m)
            self._net.import_state({k: opt.state[p][kv] for p, k in zip(params, keys)}, self._net.v)
            steps = int(float(opt.state[params[0]]["step"]))
        g0 = opt.param_groups[0]
        self._adam_steps, self._lr_now = steps, float(g0["lr"])
        d = dict(g0)
        d["lr"] = self._lr_now
        self._set_native_hyper(d, steps)

    def _import_optim_state(self):  # BaseAgent.load_full(): load() already imported the moments
        pass
