"""Frame de-duplication for image replay.

The reference's Atari wrapper hands over `state` / `next_state` as stacks of the last C frames
(core/env/atari.py:147-149), and the n-step assembler (rainbow.py:294-308) copies them again: consecutive
transitions share C - 1 of their C frames, and `next_state` of step t is `state` of step t + n.  Stored as is,
a transition costs 2 x C x H x W bytes of HBM and of PCIe traffic (56 KB for 4 x 84 x 84).

Here every distinct FRAME is stored once in a device frame pool and a transition keeps 2 x C int64 slot numbers:
the drop-in `store` API is unchanged -- frames are recognised by a 128-bit content hash (xxh3) on the host, only
new ones are uploaded (typically one 7 KB frame per env step: 8 x less H2D and HBM) -- and `gather` rebuilds the
stacks on the device with the ordinary row-gather kernel (frame pool rows indexed by the gathered slot numbers).
Slots are reference counted on the host and recycled when the transitions that use them are overwritten.
"""
import numpy as np
import torch
import xxhash

from ... import _lib as L
from ... import ops


class FramePool:
    KEYS = ("state", "next_state")

    def __init__(self, buffer_size, C, frame_shape, device, pool_factor=2.0):
        self.N, self.C, self.frame_shape = int(buffer_size), int(C), tuple(int(v) for v in frame_shape)
        self.elems = int(np.prod(self.frame_shape))
        self.F = int(self.N * pool_factor) + 4 * self.C + 64
        self.pool = ops.DeviceStore(self.F, [("frame", L.JH_U8, self.elems, self.frame_shape)], device=device)
        self.device = self.pool.device
        self.free = list(range(self.F - 1, -1, -1))
        self.ref = np.zeros(self.F, np.int32)
        self.table = {}  # content hash -> slot
        self.slot_key = [None] * self.F
        self.row_slots = np.full((self.N, 2 * self.C), -1, np.int64)  # what each transition row references
        self._pend_frames, self._pend_slots, self._pend_set = [], [], set()
        self._idx_buf = {}
        self.frames_uploaded = 0
        self.frames_referenced = 0

    def _release_row(self, pos):
        old = self.row_slots[pos]
        if old[0] < 0:
            return
        for s in old:
            self.ref[s] -= 1
            if self.ref[s] == 0:
                del self.table[self.slot_key[s]]
                self.slot_key[s] = None
                self.free.append(int(s))
        self.row_slots[pos] = -1

    def encode(self, cols, positions):
        """cols["state"], cols["next_state"]: uint8 [n, C, *frame_shape] -> int64 [n, C] slot numbers (new frames are
        queued for upload); `positions` = the ring rows these transitions will occupy (their old contents die)."""
        n, C = len(positions), self.C
        for p in positions:
            self._release_row(int(p))
        fidx = np.empty((n, 2 * C), np.int64)
        stacks = [np.ascontiguousarray(cols[k]).reshape(n, C, self.elems) for k in self.KEYS]
        for i in range(n):
            for j in range(2 * C):
                frame = stacks[j // C][i, j % C]
                key = xxhash.xxh3_128_digest(frame)
                s = self.table.get(key)
                if s is None:
                    if not self.free:
                        raise RuntimeError(f"frame pool exhausted ({self.F} frames for {self.N} transitions): raise frame_pool_factor "
                                           "(observations share fewer frames than the frame-stacking wrapper implies)")
                    s = self.free.pop()
                    if s in self._pend_set:
                        # the slot was released (its transitions were overwritten) while ITS upload is still queued: two
                        # rows for one slot in one scatter launch would land in undefined order -- write the queue out first
                        self.flush()
                    self._pend_set.add(s)
                    self.table[key] = s
                    self.slot_key[s] = key
                    self._pend_frames.append(frame.copy())
                    self._pend_slots.append(s)
                self.ref[s] += 1
                fidx[i, j] = s
            self.row_slots[positions[i]] = fidx[i]
        self.frames_referenced += n * 2 * C
        out = dict(cols)
        out["state"], out["next_state"] = fidx[:, :C], fidx[:, C:]
        return out

    def flush(self):
        if self._pend_slots:
            self.pool.write_rows(np.asarray(self._pend_slots, np.int64), {"frame": np.stack(self._pend_frames, 0)})
            self.frames_uploaded += len(self._pend_slots)
            self._pend_frames, self._pend_slots, self._pend_set = [], [], set()

    def idx_buffer(self, B):
        if B not in self._idx_buf:
            self._idx_buf[B] = torch.zeros(2 * B, self.C, dtype=torch.int64, device=self.device)
        return self._idx_buf[B]

    def decode(self, fidx, out, as_float):
        """fidx int64 [m, C] (device) -> out [m, C, *frame_shape] (uint8, or fp32 = as_tensor semantics)."""
        self.pool.gather(fidx.reshape(-1), names=["frame"], as_float=as_float, out={"frame": out.view((-1,) + self.frame_shape)})
        return out

    def stats(self):
        return {"frames_uploaded": self.frames_uploaded, "frames_referenced": self.frames_referenced, "pool_slots": self.F,
                "pool_in_use": self.F - len(self.free), "bytes_per_transition_plain": 2 * self.C * self.elems,
                "bytes_uploaded_per_transition": self.frames_uploaded * self.elems / max(1, self.frames_referenced // (2 * self.C))}


class LockstepFramePool:
    """The device-fed counterpart of FramePool for N lockstep actors (manager/batched_actors.py: DeviceActorFeed): every
    actor owns a private ring of `planes_per_actor` plane slots, filled by jh_feed_tick on the acting stream; transitions
    keep 2 x C slot numbers exactly like FramePool's, so `decode` (and therefore sample / gather / checkpoints) is shared.
    No host hashing, no host reference counts: a plane outlives every transition that can reference it because an actor
    allocates at most `planes_per_actor` planes in `window_ticks` ticks (checked on the device, see `check`)."""

    KEYS = ("state", "next_state")

    def __init__(self, buffer_size, n_actors, C, frame_shape, n_step, gamma, device, pool_factor=1.5, in_flight_ticks=64):
        self.N, self.C, self.frame_shape = int(buffer_size), int(C), tuple(int(v) for v in frame_shape)
        self.n_actors, self.n_step = int(n_actors), int(n_step)
        self.elems = int(np.prod(self.frame_shape))
        live = -(-self.N // self.n_actors)  # ticks until a stored row is overwritten (every tick stores n_actors rows)
        self.window = live + n_step + 1 + self.C + int(in_flight_ticks)
        # one new plane per tick unless the stack is discontinuous (reset: C planes); pool_factor is the head room
        self.R = int(np.ceil(self.window * float(pool_factor))) + 2 * self.C
        self.F = self.n_actors * self.R
        self.pool = ops.DeviceStore(self.F, [("frame", L.JH_U8, self.elems, self.frame_shape)], device=device)
        self.device = self.pool.device
        self.feed = ops.ActorFeed(self.n_actors, self.C, self.elems, self.n_step, gamma, self.R, self.window, device=self.device)
        self.planes = self.pool.column("frame")
        self._idx_buf = {}
        self.rows_stored = 0

    def encode(self, cols, positions):
        raise RuntimeError("this buffer is fed on the device (attach_actor_feed): host-side store() is not available")

    def flush(self):
        pass

    def idx_buffer(self, B):
        if B not in self._idx_buf:
            self._idx_buf[B] = torch.zeros(2 * B, self.C, dtype=torch.int64, device=self.device)
        return self._idx_buf[B]

    def decode(self, fidx, out, as_float):
        self.pool.gather(fidx.reshape(-1), names=["frame"], as_float=as_float, out={"frame": out.view((-1,) + self.frame_shape)})
        return out

    def check(self):
        """Raise if an actor overran its plane ring (blocking read of the device flag)."""
        flags, written = self.feed.state()
        if flags & 1:
            raise RuntimeError(f"plane ring overrun: an actor wrote more than {self.R} planes within {self.window} ticks (stacks are discontinuous "
                               "far more often than a frame-stacking wrapper implies): raise pool_factor")
        return written

    # ---- resume: plane pool + the feed's device state; the rows / tree are the buffer's -----------------
    def geometry(self):
        return {"n_actors": self.n_actors, "C": self.C, "frame_shape": list(self.frame_shape), "n_step": self.n_step, "planes_per_actor": self.R,
                "window_ticks": self.window, "pool_slots": self.F, "buffer_size": self.N}

    def save_stream(self, dirpath, chunk_bytes=64 << 20):
        """planes.bin (the pool, streamed from HBM in chunks) + feed_state.bin (jh_feed_save).  Raises on a plane-ring overrun."""
        import os

        self.check()
        rows_per = max(1, chunk_bytes // self.elems)
        with open(os.path.join(dirpath, "planes.bin"), "wb") as f:
            for o in range(0, self.F, rows_per):
                f.write(self.planes[o : o + rows_per].cpu().numpy().tobytes())
        self.feed.save_state().tofile(os.path.join(dirpath, "feed_state.bin"))
        return {"geometry": self.geometry(), "planes": "planes.bin", "state": "feed_state.bin", "rows_stored": int(self.rows_stored)}

    def load_stream(self, dirpath, meta, chunk_bytes=64 << 20):
        import os

        if meta["geometry"] != self.geometry():
            raise ValueError(f"the saved feed has another geometry: {meta['geometry']} vs {self.geometry()} (same buffer_size, actors, frame stack, n_step, "
                             "pool_factor and depth are needed to continue a device-fed replay)")
        rows_per = max(1, chunk_bytes // self.elems)
        with open(os.path.join(dirpath, meta["planes"]), "rb") as f:
            for o in range(0, self.F, rows_per):
                m = min(rows_per, self.F - o)
                a = np.frombuffer(f.read(m * self.elems), dtype=np.uint8).reshape((m,) + tuple(self.planes.shape[1:]))
                self.planes[o : o + m].copy_(torch.from_numpy(a.copy()))
        self.feed.load_state(np.fromfile(os.path.join(dirpath, meta["state"]), dtype=np.uint8))
        self.rows_stored = int(meta["rows_stored"])

    def stats(self):
        written = self.check()
        return {"planes_written": written, "pool_slots": self.F, "planes_per_actor": self.R, "window_ticks": self.window,
                "bytes_per_transition_plain": 2 * self.C * self.elems,
                "pool_bytes_per_buffer_row": self.F * self.elems / self.N,
                "planes_per_stored_row": written / max(1, self.rows_stored)}
