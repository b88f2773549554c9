import torch

from ... import ops
from .base import BaseBuffer


class RolloutBuffer(BaseBuffer):
    """core/buffer/rollout_buffer.py:6-24 on the GPU: append in arrival (worker-major) order, `sample()`
    returns EVERYTHING as float32 device tensors and clears.  The store grows by doubling
    (device-to-device carry-over), like the reference's unbounded list."""

    def __init__(self, device=None, capacity=4096):
        super().__init__(device)
        self._capacity = int(capacity)
        self._arange = None

    def _ensure(self, cols, n):
        if self._store is None:
            while self._capacity < n:
                self._capacity *= 2
            self._make_store(cols, self._capacity)
        elif self._store.size + n > self._capacity:
            old, old_n = self._store, self._store.size
            while self._capacity < old_n + n:
                self._capacity *= 2
            self._store = ops.DeviceStore(self._capacity, old.columns, device=self.device)
            if old_n:
                self._store.push_device({name: old.column(name) for name in old.names}, old_n)
                torch.cuda.current_stream().synchronize()  # `old` is freed when it goes out of scope
            self._arange = None

    def store_soa(self, cols):
        first = next(iter(cols.values()))
        n = len(first[0]) if isinstance(first, list) else len(first)
        self._ensure(cols, n)
        self._store.push(self._flat_cols(cols))

    def store(self, transitions):
        if self.first_store:
            self.check_dim(transitions[0])
        if not transitions:
            return
        self.store_soa(self.stack_transition(transitions))

    def sample(self, as_float=True):
        n = self._store.size
        if self._arange is None or self._arange.numel() < n:
            self._arange = torch.arange(self._capacity, dtype=torch.int64, device=self.device)
        out = self._unflatten(self._store.gather(self._arange[:n], as_float=as_float))
        self._store.clear()
        return out

    @property
    def size(self):
        return 0 if self._store is None else self._store.size
