"""Drop-in buffers with JORLDY's interface (core/buffer/*), backed by libjorldy_hip.so:
transitions live in a GPU-resident struct-of-arrays ring, sampling is a fused gather kernel that
returns float32 device tensors (what BaseAgent.as_tensor would have produced), and the PER sum
tree is a device float64 array updated by batch kernels that keep it bit-identical to the
reference's numpy tree."""
from .base import BaseBuffer
from .replay_buffer import ReplayBuffer
from .rollout_buffer import RolloutBuffer
from .per_buffer import PERBuffer

__all__ = ["BaseBuffer", "ReplayBuffer", "RolloutBuffer", "PERBuffer"]
