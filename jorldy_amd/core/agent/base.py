from abc import ABC, abstractmethod

import numpy as np
import torch


class BaseAgent(ABC):
    """core/agent/base.py:6-111 -- the drop-in boundary: act / learn / process / save / load /
    sync_in / sync_out / set_distributed / interact_callback keep the reference's signatures."""

    @abstractmethod
    def act(self, state):
        """state ndarray (N, *D_state) -> {"action": ndarray (N, *D_action), ...}"""

    @abstractmethod
    def learn(self):
        """-> dict of python floats to log"""

    @abstractmethod
    def process(self, transitions, step):
        """store + (periodically) learn; -> result dict"""

    @abstractmethod
    def save(self, path):
        pass

    @abstractmethod
    def load(self, path):
        pass

    def as_tensor(self, x):
        """base.py:61-73: everything becomes float32 on the agent's device."""
        if isinstance(x, list):
            return [torch.as_tensor(v, dtype=torch.float32, device=self.device) for v in x]
        return torch.as_tensor(x, dtype=torch.float32, device=self.device)

    def sync_in(self, weights):
        self.network.load_state_dict(weights)

    def sync_out(self, device="cpu"):
        weights = self.network.state_dict()
        for k, v in weights.items():
            weights[k] = v.to(device)
        return {"weights": weights}

    def set_distributed(self, *args, **kwargs):
        return self

    def interact_callback(self, transition):
        return transition

    # lr(step) = lr0 * schedule(step / run_step): the reference's three annealing shapes (base.py:93-111), one table for every agent
    _LR_SCHEDULES = {
        "linear": lambda frac: 1.0 - frac,
        "cosine": lambda frac: float(np.cos(0.5 * np.pi * frac)),
        "sqrt": lambda frac: max(1.0 - frac, 0.0) ** 0.5,
    }

    def _lr_weight(self, step, mode="cosine"):
        schedule = self._LR_SCHEDULES.get(mode)
        if schedule is None:
            raise Exception(f"check learning rate decay mode again! => {mode}")
        return schedule(step / self.run_step)

    def learning_rate_decay(self, step, optimizers=None, mode="cosine"):
        weight = self._lr_weight(step, mode)
        targets = optimizers if optimizers is not None else self.optimizer
        for opt in (targets if isinstance(targets, list) else [targets]):
            for group in opt.param_groups:
                group["lr"] = opt.defaults["lr"] * weight

    # ---- complete checkpoint (beyond the reference's {"network", "optimizer"} ckpt) -------------------
    _RESUME_ATTRS = ("time_t", "learn_stamp", "num_learn", "epsilon", "beta", "target_update_stamp", "learn_period_stamp",
                     "num_transitions", "_adam_steps")

    def _read_stats(self, *device_tensors):
        """The learn() statistics in ONE host synchronisation: asynchronous copies into pinned host buffers, one stream
        sync (a `.cpu()` per tensor is a blocking hipMemcpy each)."""
        pins = self.__dict__.setdefault("_stat_pins", {})
        outs = []
        for i, t in enumerate(device_tensors):
            key = (i, t.dtype, tuple(t.shape))
            if key not in pins:
                pins[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            pins[key].copy_(t, non_blocking=True)
            outs.append(pins[key])
        torch.cuda.current_stream().synchronize()
        return [o.numpy() for o in outs]

    # ---- learn() statistics in device-MAPPED pinned host memory: the loss kernels write them across PCIe themselves
    # and learn() waits for their arrival instead of for the stream (no D2H copies, no hipStreamSynchronize wake-up, and
    # the host goes on -- env steps, stores, the next rollout -- while the rest of the update is still running; everything
    # it enqueues is stream-ordered behind it).  JH_MAPPED_STATS=0: device tensors + _read_stats.
    def _mapped_stats(self, n, np_dtype=np.float32):
        """-> (CUDA tensor aliasing the buffer for the kernels, numpy view for the host) or (device tensor, None)."""
        import os

        from ... import ops

        tdt = torch.float32 if np.dtype(np_dtype) == np.float32 else torch.float64
        if os.environ.get("JH_MAPPED_STATS", "1") != "1":
            return torch.zeros(n, dtype=tdt, device=self.device), None
        pin = ops.PinnedBuffer((n,), np_dtype, self.device.index)
        pin.np[:] = 0
        return ops._wrap_device(pin.dev_ptr.value, (n,), tdt, self.device, owner=pin), pin.np

    @staticmethod
    def _await_marks(view, marks, what="learn()"):
        """Wait until view[m] != -1 for every m in marks (the elements the finishing kernel writes last; the caller set
        them to -1 before launching).  The wait is jh_host_wait_marks: a C spin entered through ctypes, i.e. without
        the GIL (a Python spin loop here cost the batched actors of an Ape-X process a third of their throughput).
        Bounded: after 2 s a stream sync, which also surfaces asynchronous errors."""
        from ... import _lib as L

        idx = np.asarray(marks, dtype=np.int32)
        flat = view.reshape(-1)
        if L.load().jh_host_wait_marks(L.ptr(flat), L.ptr(idx), int(idx.size), -1.0, 2.0) != 0:
            torch.cuda.current_stream().synchronize()
            if any(flat[m] == -1.0 for m in marks):
                raise RuntimeError(f"{what}: the statistics never arrived (failed launch?)")

    RESUME_FORMAT, RESUME_VERSION = "jorldy_amd.resume", 2

    def _resume_extra_attrs(self):
        """Subclass hook: more JSON-able state for the manifest (e.g. the native acting RNG counters)."""
        return {}

    def _resume_load_extra_attrs(self, d):
        pass

    def save_full(self, path, version=None):
        """`save(path)` (the reference's ckpt, unchanged) + everything the reference cannot resume (SURVEY.md §8f rank 4,
        core/agent/dqn.py:184-199): replay buffer / sum tree contents, step counters, epsilon / beta, numpy + torch
        RNG state.

        Format version 2 (default): directory `path/resume/` with `manifest.json` ({"format", "version", ...}) and one
        raw file per replay column, streamed from HBM in 64 MB chunks -- no host copy of the whole buffer (56 GB of
        frames at config.rainbow.atari's N = 1e6).  version=1 writes round 1's single `resume.pt` pickle (kept so that
        old checkpoints stay loadable and testable)."""
        import json
        import os

        version = self.RESUME_VERSION if version is None else version
        self.save(path)
        mem = getattr(self, "memory", None)
        if version == 1:
            extra = {"attrs": {k: getattr(self, k) for k in self._RESUME_ATTRS if hasattr(self, k)},
                     "numpy_rng": np.random.get_state(), "torch_rng": torch.get_rng_state(), "torch_cuda_rng": torch.cuda.get_rng_state(self.device)}
            if mem is not None and hasattr(mem, "state_dict"):
                extra["memory"] = mem.state_dict()
            if hasattr(self, "target_network"):
                extra["target_network"] = {k: v.cpu() for k, v in self.target_network.state_dict().items()}
            torch.save(extra, os.path.join(path, "resume.pt"))
            return
        assert version == 2, f"unknown resume format version {version}"
        d = os.path.join(path, "resume")
        os.makedirs(d, exist_ok=True)
        num = lambda v: v.item() if isinstance(v, (np.generic, torch.Tensor)) else v
        rs = np.random.get_state()
        man = {"format": self.RESUME_FORMAT, "version": 2, "agent": type(self).__name__,
               "attrs": {k: num(getattr(self, k)) for k in self._RESUME_ATTRS if hasattr(self, k)},
               "extra_attrs": self._resume_extra_attrs(),
               "numpy_rng": {"kind": rs[0], "pos": int(rs[2]), "has_gauss": int(rs[3]), "cached_gaussian": float(rs[4]), "keys": "numpy_rng_keys.u32"},
               "torch_rng": "torch_rng.u8", "torch_cuda_rng": "torch_cuda_rng.u8"}
        np.asarray(rs[1], dtype=np.uint32).tofile(os.path.join(d, man["numpy_rng"]["keys"]))
        torch.get_rng_state().numpy().tofile(os.path.join(d, man["torch_rng"]))
        torch.cuda.get_rng_state(self.device).numpy().tofile(os.path.join(d, man["torch_cuda_rng"]))
        if mem is not None and hasattr(mem, "save_stream"):
            man["memory"] = mem.save_stream(d)
        if hasattr(self, "target_network"):
            torch.save({k: v.cpu() for k, v in self.target_network.state_dict().items()}, os.path.join(d, "target_network.pt"))
            man["target_network"] = "target_network.pt"
        with open(os.path.join(d, "manifest.json"), "w") as f:
            json.dump(man, f, indent=1)

    def load_full(self, path):
        """Loads either resume format: `path/resume/manifest.json` (version 2) or round 1's `path/resume.pt` (version 1)."""
        import json
        import os

        self.load(path)
        d = os.path.join(path, "resume")
        if os.path.exists(os.path.join(d, "manifest.json")):
            man = json.load(open(os.path.join(d, "manifest.json")))
            if man.get("format") != self.RESUME_FORMAT or man.get("version") != 2:
                raise ValueError(f"unsupported resume manifest: format {man.get('format')!r} version {man.get('version')!r}")
            for k, v in man["attrs"].items():
                setattr(self, k, v)
            r = man["numpy_rng"]
            np.random.set_state((r["kind"], np.fromfile(os.path.join(d, r["keys"]), dtype=np.uint32), r["pos"], r["has_gauss"], r["cached_gaussian"]))
            torch.set_rng_state(torch.from_numpy(np.fromfile(os.path.join(d, man["torch_rng"]), dtype=np.uint8)))
            torch.cuda.set_rng_state(torch.from_numpy(np.fromfile(os.path.join(d, man["torch_cuda_rng"]), dtype=np.uint8)), self.device)
            if "memory" in man and hasattr(self.memory, "load_stream"):
                self.memory.load_stream(d, man["memory"])
            if "target_network" in man:
                self.target_network.load_state_dict(torch.load(os.path.join(d, man["target_network"]), map_location="cpu", weights_only=False))
            self._resume_load_extra_attrs(man.get("extra_attrs", {}))
        else:
            extra = torch.load(os.path.join(path, "resume.pt"), map_location="cpu", weights_only=False)  # version 1
            for k, v in extra["attrs"].items():
                setattr(self, k, v)
            np.random.set_state(extra["numpy_rng"])
            torch.set_rng_state(extra["torch_rng"])
            torch.cuda.set_rng_state(extra["torch_cuda_rng"], self.device)
            if "memory" in extra and hasattr(self.memory, "load_state_dict"):
                self.memory.load_state_dict(extra["memory"])
            if "target_network" in extra:
                self.target_network.load_state_dict(extra["target_network"])
        if getattr(self, "_net", None) is not None:
            self._import_optim_state()

    @staticmethod
    def _require_gpu(device):
        dev = torch.device(device) if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if dev.type != "cuda":
            raise RuntimeError(
                "jorldy_amd agents run their learn() path on HIP kernels and have no CPU fallback "
                f"(requested device: {dev}). Use the reference agent for CPU runs."
            )
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev
