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
