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

    def learning_rate_decay(self, step, optimizers=None, mode="cosine"):
        """base.py:93-111."""
        if mode == "linear":
            weight = 1 - (step / self.run_step)
        elif mode == "cosine":
            weight = np.cos((np.pi / 2) * (step / self.run_step))
        elif mode == "sqrt":
            weight = (1 - (step / self.run_step)) ** (1 / 2)
        else:
            raise Exception(f"check learning rate decay mode again! => {mode}")
        if optimizers is None:
            optimizers = [self.optimizer]
        if not isinstance(optimizers, list):
            optimizers = [optimizers]
        for optimizer in optimizers:
            for g in optimizer.param_groups:
                g["lr"] = optimizer.defaults["lr"] * weight

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

    def save_full(self, path):
        """`save(path)` (reference format, unchanged) + `resume.pt`: replay buffer / sum tree contents,
        step counters, epsilon / beta, numpy + torch RNG state -- what the reference cannot resume."""
        import os

        self.save(path)
        extra = {"attrs": {k: getattr(self, k) for k in self._RESUME_ATTRS if hasattr(self, k)},
                 "numpy_rng": np.random.get_state(), "torch_rng": torch.get_rng_state(), "torch_cuda_rng": torch.cuda.get_rng_state(self.device)}
        mem = getattr(self, "memory", None)
        if mem is not None and hasattr(mem, "state_dict"):
            extra["memory"] = mem.state_dict()
        if hasattr(self, "target_network"):
            extra["target_network"] = {k: v.cpu() for k, v in self.target_network.state_dict().items()}
        torch.save(extra, os.path.join(path, "resume.pt"))

    def load_full(self, path):
        import os

        self.load(path)
        extra = torch.load(os.path.join(path, "resume.pt"), map_location="cpu", weights_only=False)
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
