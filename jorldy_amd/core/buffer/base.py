from abc import ABC, abstractmethod

import numpy as np
import torch

from ... import _lib as L
from ... import ops


def _jh_dtype(arr, key=None):
    """Stored dtype of a column.  Everything becomes float32 at BaseAgent.as_tensor
    (core/agent/base.py:61-73), so floats are stored as float32 (same rounding), uint8 frames and
    bools as one byte.  Integer-typed arrays are kept as int64 only for the columns that are integral by
    meaning (discrete actions, the frame-slot numbers of the de-duplicated replay); any other column whose FIRST
    batch happens to be integer-typed (gym_env.py:78 computes `reward = -1 if done else 0.1`) is stored as
    float32 like as_tensor would make it -- a later fractional value must not be truncated."""
    k = arr.dtype.kind
    if arr.dtype == np.uint8 or k == "b":
        return L.JH_U8
    if k in "iu" and (key is None or key in ("action", "state", "next_state")):
        return L.JH_I64
    return L.JH_F32


class BaseBuffer(ABC):
    """Same abstract interface as core/buffer/base.py:5-56.  `stack_transition` (AoS->SoA) happens
    once, at store time on the host; samples come back as device tensors."""

    def __init__(self, device=None):
        self.first_store = True
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._store = None
        self._layout = None  # list of (key, sub_index or None, column name)

    def check_dim(self, transition):
        print("########################################")
        print("You should check dimension of transition")
        for key, val in transition.items():
            if isinstance(val, list):
                for i, v in enumerate(val):
                    print(f"{key}{i}: {v.shape}")
            else:
                print(f"{key}: {np.asarray(val).shape}")
        print("########################################")
        self.first_store = False

    @abstractmethod
    def store(self, transitions):
        """transitions: List[Dict] (the reference format, each value with a leading dim of 1)."""

    @abstractmethod
    def sample(self, batch_size):
        """returns Dict[str, Tensor] of float32 device tensors."""

    # ---- AoS -> SoA on the host (base.py:42-56), done once per store ------------------------------
    @staticmethod
    def stack_transition(batch, skip=()):
        out = {}
        for key in batch[0].keys():
            if key in skip:
                continue
            v0 = batch[0][key]
            if isinstance(v0, list):  # multimodal
                out[key] = [np.stack([b[key][i][0] for b in batch], axis=0) for i in range(len(v0))]
            else:
                out[key] = np.stack([np.asarray(b[key])[0] for b in batch], axis=0)
        return out

    def _make_store(self, cols, capacity):
        """cols: dict key -> ndarray [n, ...] or list of ndarrays (multimodal)."""
        layout, columns = [], []
        for key, v in cols.items():
            parts = v if isinstance(v, list) else [v]
            for i, a in enumerate(parts):
                a = np.asarray(a)
                name = f"{key}#{i}" if isinstance(v, list) else key
                shape = tuple(a.shape[1:])
                columns.append((name, _jh_dtype(a, key), int(np.prod(shape)) if shape else 1, shape if shape else (1,)))
                layout.append((key, i if isinstance(v, list) else None, name))
        self._layout = layout
        self._store = ops.DeviceStore(capacity, columns, device=self.device)

    def _flat_cols(self, cols):
        flat = {}
        for key, sub, name in self._layout:
            flat[name] = cols[key][sub] if sub is not None else cols[key]
        return flat

    def _unflatten(self, flat):
        out = {}
        for key, sub, name in self._layout:
            if sub is None:
                out[key] = flat[name]
            else:
                out.setdefault(key, []).append(flat[name])
        return out


_PIN_RING = {}


def h2d_small(arr, device):
    """Small host array -> device tensor through a reusable pinned buffer (async copy on the current
    stream; the ring is deep enough that a slot is never rewritten while its copy is in flight)."""
    arr = np.ascontiguousarray(arr)
    key = (str(device), arr.dtype.str)
    ring = _PIN_RING.setdefault(key, {"bufs": [None] * 16, "i": 0, "ev": [None] * 16})
    i = ring["i"]
    ring["i"] = (i + 1) % 16
    buf = ring["bufs"][i]
    if ring["ev"][i] is not None:
        ring["ev"][i].synchronize()
    if buf is None or buf.numel() < arr.size:
        buf = torch.empty(max(arr.size, 1024), dtype=torch.from_numpy(arr[:0]).dtype, pin_memory=True)
        ring["bufs"][i] = buf
    view = buf[: arr.size]
    view.numpy()[:] = arr.reshape(-1)
    out = view.to(device, non_blocking=True).reshape(arr.shape)
    ev = torch.cuda.Event()
    ev.record()
    ring["ev"][i] = ev
    return out
