"""The epoch shuffles of PPO.learn (core/agent/ppo.py:116-118) on numpy's OWN global generator, faster and earlier.

The reference draws its minibatch index lists with `np.random.shuffle(idxs)` once per epoch on a cumulatively shuffled
`np.arange(M)`; the drop-in must consume the same global stream in the same order or every later index list -- and whatever
else uses `np.random` -- diverges.  Two exact devices:

  * `epoch_shuffles`: jh_np_legacy_shuffles runs numpy's algorithm (RandomState._shuffle_raw + random_interval) in C on the
    global MT19937 state through the pointers numpy publishes in `BitGenerator.ctypes` -- bit-identical index lists and
    generator state (tests/test_abi_cpu.py), ~1.5-2 x faster than three Python-level shuffle calls.
  * `Predraw`: the NEXT learn()'s shuffles are drawn while the GPU is busy with this one -- on a COPY of the generator state.
    At the next learn() the global state is compared with the snapshot (2.5 KB memcmp): unchanged -> the copy's end state is
    installed and the pre-drawn lists are what `np.random.shuffle` would have produced now; changed (somebody drew from
    np.random in between) -> the lists are discarded and drawn afresh.  Either way the stream is the reference's.

Any other global bit generator than MT19937 (np.random.set_bit_generator) falls back to np.random.shuffle itself.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_MT_BYTES = 624 * 4 + 4  # mt19937_state {uint32 key[624]; int pos}


def _global_mt():
    """(ctypes interface, state address) of the global legacy generator when it is an MT19937, else None."""
    try:
        bg = np.random.mtrand._rand._bit_generator
        if type(bg).__name__ != "MT19937":
            return None
        ct = bg.ctypes
        return ct, int(ct.state_address)
    except Exception:
        return None


def _lock():
    """numpy serialises every draw of the global generator with this lock; taking it around our raw reads / writes of the MT19937 state
    keeps another thread's np.random call from running in the middle of them."""
    return np.random.mtrand._rand._bit_generator.lock


def _fn_ptrs(ct):
    return C.cast(ct.next_uint32, C.c_void_p), C.cast(ct.next_uint64, C.c_void_p)


def epoch_shuffles(M, n_epoch, out):
    """out: int64 numpy array [n_epoch * M] (any memory, e.g. pinned) <- the index list of every epoch, consuming the global
    np.random stream exactly like `idxs = np.arange(M); for e: np.random.shuffle(idxs)`."""
    g = _global_mt()
    flat = out.reshape(-1)
    assert flat.dtype == np.int64 and flat.size == n_epoch * M and flat.flags["C_CONTIGUOUS"]
    if g is None:
        idxs = np.arange(M)
        for e in range(n_epoch):
            np.random.shuffle(idxs)
            flat[e * M : (e + 1) * M] = idxs
        return out
    ct, addr = g
    f32, f64 = _fn_ptrs(ct)
    with _lock():
        L.check(L.load().jh_np_legacy_shuffles(C.c_void_p(addr), f32, f64, int(M), int(n_epoch), C.c_void_p(flat.ctypes.data)))
    return out


class Predraw:
    """Index lists of the next learn(), drawn ahead on a copy of the global generator (see the module docstring)."""

    def __init__(self):
        self.valid = False
        self._s0 = self._s1 = None
        self._scratch = C.create_string_buffer(_MT_BYTES + 8)
        self.key = None

    def draw(self, M, n_epoch, out):
        """Draw into `out` from a copy of the current global state; remember the state before (s0) and after (s1)."""
        self.valid = False
        g = _global_mt()
        if g is None:
            return False
        ct, addr = g
        with _lock():
            self._s0 = C.string_at(addr, _MT_BYTES)
        C.memmove(self._scratch, self._s0, _MT_BYTES)
        f32, f64 = _fn_ptrs(ct)
        flat = out.reshape(-1)
        L.check(L.load().jh_np_legacy_shuffles(C.cast(self._scratch, C.c_void_p), f32, f64, int(M), int(n_epoch), C.c_void_p(flat.ctypes.data)))
        self._s1 = self._scratch.raw[:_MT_BYTES]
        self.key, self.valid = (int(M), int(n_epoch)), True
        return True

    def commit(self, M, n_epoch):
        """True: the global generator is where draw() left it -> advanced to the post-shuffle state, the lists stand."""
        if not self.valid or self.key != (int(M), int(n_epoch)):
            self.valid = False
            return False
        self.valid = False
        g = _global_mt()
        if g is None:
            return False
        _, addr = g
        with _lock():  # compare and install atomically with respect to other threads' draws
            if C.string_at(addr, _MT_BYTES) != self._s0:
                return False
            C.memmove(addr, self._s1, _MT_BYTES)
        return True
