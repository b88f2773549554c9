import numpy as np

from ... import ops
from .base import BaseBuffer, h2d_small


class ReplayBuffer(BaseBuffer):
    """core/buffer/replay_buffer.py:8-35 on the GPU.

    store():   ring write (buffer_index wraps, buffer_counter saturates) through pinned staging +
               hipMemcpyAsync.
    sample():  `np.random.randint(buffer_counter, size=B)` on the host -- the SAME global-RNG draw as
               the reference, so sampled indices are bit-identical -- then one fused gather kernel.
    """

    def __init__(self, buffer_size, device=None, frame_dedup=False, frame_pool_factor=2.0):
        super().__init__(device)
        self.buffer_size = int(buffer_size)
        # frame_dedup: image observations (uint8 [C, H, W] frame stacks) are stored as single frames + slot numbers
        # (frame_dedup.py: 8 x less HBM and H2D for 4-frame Atari stacks); the API and the samples are unchanged
        self.frame_dedup, self.frame_pool_factor, self._frames = bool(frame_dedup), float(frame_pool_factor), None
        self.buffer_index = 0
        self.buffer_counter = 0
        # Coalesced stores (opt-in, set by the agents): pushes of <= defer_rows transitions are held on
        # the host and written to HBM as ONE ring append right before the next device-side read
        # (gather / sample / priority update / checkpoint).  Nothing observable changes -- buffer_index,
        # buffer_counter and size advance immediately, sampling never runs between store and flush --
        # but the per-env-step cost drops from ~8 HIP API calls to a numpy copy (Rainbow: a store every
        # step, a learn() every 4).
        self.defer_rows = 0
        self._pending, self._pending_rows, self._pend_cols = [], 0, None

    # fast path: already-SoA numpy columns [n, ...]
    def _dedup_encode(self, cols):
        st = cols.get("state")
        if self._frames is None:
            ok = (isinstance(st, np.ndarray) and st.dtype == np.uint8 and st.ndim == 4 and isinstance(cols.get("next_state"), np.ndarray)
                  and cols["next_state"].shape == st.shape and cols["next_state"].dtype == np.uint8)
            if not ok:
                raise ValueError("frame_dedup=True needs uint8 image observations 'state' / 'next_state' of shape [n, C, H, W]")
            from .frame_dedup import FramePool

            self._frames = FramePool(self.buffer_size, st.shape[1], st.shape[2:], self.device, self.frame_pool_factor)
        n = len(st)
        assert n <= self.buffer_size
        positions = (self.buffer_index + np.arange(n)) % self.buffer_size
        return self._frames.encode(cols, positions)

    def store_soa(self, cols, n=None):
        if self.frame_dedup:
            cols = self._dedup_encode(cols)
        if self._store is None:
            self._make_store(cols, self.buffer_size)
        flat = self._flat_cols(cols)
        n = len(next(iter(flat.values())))
        self._was_deferred = 0 < n <= self.defer_rows and self._pending_rows + n <= min(4 * self.defer_rows, self.buffer_size)
        if self._was_deferred:
            self._defer(flat, n)
        else:
            self.flush()
            n = self._store.push(flat)
        self.buffer_index = (self.buffer_index + n) % self.buffer_size
        self.buffer_counter = min(self.buffer_counter + n, self.buffer_size)
        return n

    def store_device(self, cols, example=None):
        """Rows that are already in HBM (a device-side producer, synthetic prefills): `cols` maps the agent-side keys
        to CUDA tensors [n, ...] in the STORED dtype (uint8 frames, float32, int64 actions, uint8 done flags);
        device-to-device ring append, same bookkeeping as store().  `example`: one host SoA batch that fixes the
        column layout when nothing has been stored yet."""
        assert not self.frame_dedup, "frame_dedup recognises frames on the host"
        if self._store is None:
            assert example is not None, "the buffer has no layout yet: pass an example batch"
            self._make_store(example, self.buffer_size)
        self.flush()
        flat = self._flat_cols(cols)
        n = int(next(iter(flat.values())).shape[0])
        self._store.push_device(flat, n)
        self.buffer_index = (self.buffer_index + n) % self.buffer_size
        self.buffer_counter = min(self.buffer_counter + n, self.buffer_size)
        return n

    # ---- device-resident actor feed (frames once + slot numbers, n-step windows built in HBM) ----
    def attach_actor_feed(self, n_actors, frame_stack_shape, n_step, gamma, pool_factor=1.5, in_flight_ticks=64):
        """Turn this (empty) buffer into the sink of a DeviceActorFeed (manager/batched_actors.py): rows hold 2 x C plane
        slot numbers + action + reward[n] + done[n]; planes live in a LockstepFramePool written by jh_feed_tick.
        frame_stack_shape = (C, H, W).  Returns the pool."""
        from .frame_dedup import LockstepFramePool

        assert self._store is None and self._frames is None, "attach the actor feed to an empty buffer"
        C = int(frame_stack_shape[0])
        self._frames = LockstepFramePool(self.buffer_size, n_actors, C, frame_stack_shape[1:], n_step, gamma, self.device, pool_factor, in_flight_ticks)
        self.frame_dedup = True
        example = {"state": np.zeros((1, C), np.int64), "action": np.zeros((1, 1), np.int64), "reward": np.zeros((1, n_step, 1), np.float32),
                   "next_state": np.zeros((1, C), np.int64), "done": np.zeros((1, n_step, 1), np.uint8)}
        self._make_store(example, self.buffer_size)
        self.first_store = False
        self._feeds = []
        return self._frames

    def store_feed_rows(self, cols, n, priorities=None):
        """Rows emitted by jh_feed_tick (device tensors, stored dtypes, slot numbers for state / next_state)."""
        self.flush()
        self._store.push_device(self._flat_cols(cols), n)
        self._frames.rows_stored += n
        self.buffer_index = (self.buffer_index + n) % self.buffer_size
        self.buffer_counter = min(self.buffer_counter + n, self.buffer_size)
        return n

    def _defer(self, flat, n, extra=None):
        # rows are copied (the caller may reuse its arrays) straight into preallocated host columns of the stored dtype:
        # no per-store allocations, no concatenate at flush time
        cap = min(4 * self.defer_rows, self.buffer_size)
        if self._pend_cols is not None and len(next(iter(self._pend_cols.values()))) < cap:
            self.flush()  # defer_rows was raised: held rows out first, then bigger host columns
            self._pend_cols = None
        if self._pend_cols is None:
            self._pend_cols = {name: np.empty((cap, elems), dtype=ops._NP_OF[dt]) for name, dt, elems, _ in self._store.columns}
        k = self._pending_rows
        for name, v in flat.items():
            self._pend_cols[name][k : k + n] = np.asarray(v).reshape(n, -1)
        self._pending.append((n, extra))
        self._pending_rows += n

    def _flush_rows(self):
        """Push the held rows; returns [(n, extra)] in store order."""
        if self._frames is not None:
            self._frames.flush()  # frames the held / just-pushed rows refer to
        if not self._pending:
            return []
        pend, rows = self._pending, self._pending_rows
        self._pending, self._pending_rows = [], 0
        self._store.push({name: buf[:rows] for name, buf in self._pend_cols.items()})
        return pend

    def flush(self):
        self._flush_rows()

    # ---- asynchronous ingestion (many actor threads -> this learner's store) ------------
    def make_ring(self, slots, example=None, with_priority=False):
        """Pinned lock-free staging ring with this buffer's column layout.  `example`: one SoA batch (dict key ->
        array [n, ...]) to fix the layout when nothing has been stored yet.  Actor threads call
        `ring.produce(self.ring_columns(cols), priorities)`; the learner calls `self.drain()`."""
        from ... import ops

        assert not self.frame_dedup, "the staging ring carries whole transitions; frame_dedup encodes on the storing thread"
        if self._store is None:
            assert example is not None, "the buffer has no layout yet: pass an example batch"
            self._make_store(example, self.buffer_size)
        self._ring = ops.StagingRing(slots, self._store.columns, with_priority=with_priority, device=self.device)
        return self._ring

    def ring_columns(self, cols):
        """agent-side keys (incl. list-valued multimodal keys) -> the ring's / store's flat column names"""
        return self._flat_cols(cols)

    def _drain_tree(self):
        return None

    def drain(self, max_rows=0):
        """Learner thread: move everything the actors have published into the device ring (async copies on the
        current stream, no intermediate host copy); returns the number of transitions taken."""
        self.flush()
        n = 0
        if getattr(self, "_ring", None) is not None:
            n = self._ring.drain(self._store, self._drain_tree(), max_rows)
            self.buffer_index = (self.buffer_index + n) % self.buffer_size
            self.buffer_counter = min(self.buffer_counter + n, self.buffer_size)
        for feed in getattr(self, "_feeds", ()):  # device-resident producers (DeviceActorFeed)
            n += feed.drain_into(self)
        return n

    def store(self, transitions):
        if self.first_store:
            self.check_dim(transitions[0])
        if not transitions:
            return
        self.store_soa(self.stack_transition(transitions))

    def sample_indices(self, batch_size):
        return np.random.randint(self.buffer_counter, size=batch_size)  # replay_buffer.py:26

    def gather(self, idx_device, idx_offset=0, as_float=True, out=None):
        """out: optional dict key -> preallocated tensor (or list of tensors for multimodal keys)."""
        self.flush()
        if self._frames is not None:
            return self._gather_dedup(idx_device, idx_offset, as_float, out)
        flat_out = None if out is None else self._flat_cols(out)
        return self._unflatten(self._store.gather(idx_device, as_float=as_float, idx_offset=idx_offset, out=flat_out))

    def _gather_dedup(self, idx_device, idx_offset, as_float, out):
        """Gather the slot numbers with the other columns, then rebuild the frame stacks from the frame pool (one
        more row-gather launch; both on static buffers: captured with the rest of learn())."""
        import torch

        fr, B = self._frames, int(idx_device.numel())
        keep = {k: (not as_float) if not isinstance(as_float, dict) else (not as_float.get(k, True)) for k in fr.KEYS}
        meta_float = dict(as_float) if isinstance(as_float, dict) else {n: as_float for n in self._store.names}
        meta_float.update({k: False for k in fr.KEYS})  # slot numbers stay int64
        ibuf = fr.idx_buffer(B)
        meta_out = None
        if out is not None:
            meta_out = dict(out)
            meta_out["state"], meta_out["next_state"] = ibuf[:B], ibuf[B:]
        got = self._store.gather(idx_device, as_float=meta_float, idx_offset=idx_offset, out=meta_out)
        if out is None:
            ibuf[:B].copy_(got["state"])
            ibuf[B:].copy_(got["next_state"])
        res = dict(got)
        shape = (B, fr.C) + fr.frame_shape
        tgt = {k: (out[k] if out is not None else torch.empty(shape, dtype=torch.uint8 if keep[k] else torch.float32, device=self.device)) for k in fr.KEYS}
        s, ns = tgt["state"], tgt["next_state"]
        one_buffer = (keep["state"] == keep["next_state"] and s.is_contiguous() and ns.is_contiguous()
                      and s.untyped_storage().data_ptr() == ns.untyped_storage().data_ptr()
                      and ns.data_ptr() == s.data_ptr() + s.numel() * s.element_size())
        if one_buffer:
            both = torch.as_strided(s, (2 * B, fr.C) + fr.frame_shape, s.stride())  # [state; next_state] is one buffer: one launch
            fr.decode(ibuf, both, as_float=not keep["state"])
        else:
            fr.decode(ibuf[:B], s, as_float=not keep["state"])
            fr.decode(ibuf[B:], ns, as_float=not keep["next_state"])
        res["state"], res["next_state"] = s, ns
        return res

    def sample(self, batch_size, as_float=True):
        idx = h2d_small(self.sample_indices(batch_size).astype(np.int64), self.device)
        return self.gather(idx, as_float=as_float)

    # ---- complete checkpoints (the reference saves network + optimizer only,
    # so a resumed run restarts with an empty buffer; core/agent/dqn.py:184-199) ---------------------
    def state_dict(self):
        """Host copy of everything needed to resume: stored rows (in slot order), ring position."""
        cols = {}
        self.flush()
        if self._store is not None:
            import torch

            torch.cuda.current_stream().synchronize()
            for name in self._store.names:
                cols[name] = self._store.column(name)[: self.buffer_counter].cpu().numpy()
            if self._frames is not None:  # portable form: the full stacks, as a plain buffer would hold them
                for k in self._frames.KEYS:
                    fidx = torch.as_tensor(cols[k], device=self.device)
                    full = torch.empty((len(fidx), self._frames.C) + self._frames.frame_shape, dtype=torch.uint8, device=self.device)
                    for o in range(0, len(fidx), 4096):
                        self._frames.decode(fidx[o : o + 4096].contiguous(), full[o : o + 4096], as_float=False)
                    cols[k] = full.cpu().numpy()
        return {"buffer_size": self.buffer_size, "buffer_index": self.buffer_index, "buffer_counter": self.buffer_counter,
                "layout": self._layout, "columns": cols}

    def _refuse_load_when_fed(self):
        if getattr(self, "_feeds", None):
            raise RuntimeError("this buffer is the sink of a DeviceActorFeed: its rows reference the feed's plane rings.  A checkpoint written FROM a fed "
                               "buffer (save_full / save_stream, resume format 2) carries the plane pool and the feed's state and continues in a fed buffer "
                               "of the same geometry; the host-copy form (state_dict) and checkpoints of plain buffers restore into a plain buffer only")

    def load_state_dict(self, sd):
        self._refuse_load_when_fed()
        assert sd["buffer_size"] == self.buffer_size
        self._pending, self._pending_rows, self._pend_cols = [], 0, None
        self._frames = None
        self._layout = None
        self._store = None
        self.buffer_index = self.buffer_counter = 0
        if sd["columns"]:
            layout = sd["layout"]
            cols = {}
            for key, sub, name in layout:
                if sub is None:
                    cols[key] = sd["columns"][name]
                else:
                    cols.setdefault(key, []).append(sd["columns"][name])
            # restored rows go STRAIGHT into the ring: a deferred store (<= defer_rows rows are held on the host) would
            # be flushed after set_position below and land behind the restored rows, leaving slots [0, n) unwritten
            keep, self.defer_rows = self.defer_rows, 0
            try:
                ReplayBuffer.store_soa(self, cols)
            finally:
                self.defer_rows = keep
            assert not self._pending
        self.buffer_index = sd["buffer_index"]
        self.buffer_counter = sd["buffer_counter"]
        if self._store is not None:  # ring position of the device store follows the host counters
            import ctypes as C

            self._store.lib.jh_store_clear(self._store.h)
            self._store.lib.jh_store_set_position(self._store.h, C.c_int64(self.buffer_index), C.c_int64(self.buffer_counter))

    # ---- streamed form (resume format version 2): one raw file per column, written / read in bounded chunks straight
    # from / into HBM -- no host copy of the whole replay (56 GB of frames at config.rainbow.atari's N = 1e6) -----------
    _STREAM_CHUNK_BYTES = 64 << 20

    def save_stream(self, dirpath):
        """Write every stored column to `dirpath`/col_<i>.bin (slot order, stored dtype) and return the JSON-able
        description load_stream needs."""
        import os

        import torch

        fed = bool(getattr(self, "_feeds", None))
        if fed:  # rows the actors have emitted but the learner has not appended yet belong to the checkpoint (actors must be paused)
            self.drain()
        self.flush()
        meta = {"buffer_size": self.buffer_size, "buffer_index": self.buffer_index, "buffer_counter": self.buffer_counter,
                "frame_dedup": bool(self._frames is not None), "layout": None, "columns": []}
        if self._store is None:
            return meta
        meta["layout"] = [[k, sub, name] for k, sub, name in self._layout]
        n = self.buffer_counter
        torch.cuda.current_stream().synchronize()
        for i, (name, dt, elems, shape) in enumerate(self._store.columns):
            col = self._store.column(name)
            dec = self._frames is not None and name in self._frames.KEYS  # portable form: full frame stacks
            row_shape = ((self._frames.C,) + tuple(self._frames.frame_shape)) if dec else tuple(shape)
            np_dt = np.dtype(np.uint8) if dec else np.dtype(ops._NP_OF[dt])
            row_bytes = int(np.prod(row_shape)) * np_dt.itemsize
            rows_per = max(1, self._STREAM_CHUNK_BYTES // row_bytes)
            fn = f"col_{i}.bin"
            with open(os.path.join(dirpath, fn), "wb") as f:
                for o in range(0, n, rows_per):
                    m = min(rows_per, n - o)
                    if dec:
                        fidx = col[o : o + m].contiguous()
                        full = torch.empty((m,) + row_shape, dtype=torch.uint8, device=self.device)
                        for q in range(0, m, 4096):
                            self._frames.decode(fidx[q : q + 4096].contiguous(), full[q : q + 4096], as_float=False)
                        chunk = full
                    else:
                        chunk = col[o : o + m]
                    f.write(chunk.cpu().numpy().tobytes())
            entry = {"name": name, "dtype": np_dt.str, "shape": list(row_shape), "file": fn, "rows": n}
            if dec and fed:  # the raw slot numbers too: a fed buffer continues from them (plane pool below), a plain one from the stacks
                sfn = f"col_{i}.slots.bin"
                with open(os.path.join(dirpath, sfn), "wb") as f:
                    f.write(col[:n].cpu().numpy().tobytes())
                entry.update({"slots_file": sfn, "slots_shape": list(shape), "slots_dtype": np.dtype(ops._NP_OF[dt]).str})
            meta["columns"].append(entry)
        if fed:
            meta["feed"] = self._frames.save_stream(dirpath, self._STREAM_CHUNK_BYTES)
            meta["feed"]["producers"] = [fd.save_stream(dirpath, k) for k, fd in enumerate(self._feeds)]
        return meta

    def _load_stream_fed(self, dirpath, meta):
        """Continue a device-fed replay: rows with their slot numbers, plane pool, feed state (the sum tree follows in PERBuffer)."""
        import ctypes as C
        import os

        if "feed" not in meta:
            self._refuse_load_when_fed()
        assert meta["buffer_size"] == self.buffer_size
        n = int(meta["buffer_counter"])
        self._frames.load_stream(dirpath, meta["feed"], self._STREAM_CHUNK_BYTES)
        by_name = {c["name"]: c for c in meta["columns"]}
        self._store.lib.jh_store_clear(self._store.h)
        rows_per = 65536
        files = {}
        for name, dt, elems, shape in self._store.columns:
            c = by_name[name]
            raw = "slots_file" in c
            files[name] = (open(os.path.join(dirpath, c["slots_file"] if raw else c["file"]), "rb"), np.dtype(c["slots_dtype"] if raw else c["dtype"]), int(elems))
        try:
            for o in range(0, n, rows_per):
                m = min(rows_per, n - o)
                chunk = {}
                for name, (f, dt, elems) in files.items():
                    chunk[name] = np.frombuffer(f.read(m * elems * dt.itemsize), dtype=dt).reshape(m, elems)
                self._store.push(chunk)
        finally:
            for f, _, _ in files.values():
                f.close()
        self.buffer_index, self.buffer_counter = int(meta["buffer_index"]), n
        self._store.lib.jh_store_clear(self._store.h)
        self._store.lib.jh_store_set_position(self._store.h, C.c_int64(self.buffer_index), C.c_int64(self.buffer_counter))
        for fd, st in zip(self._feeds, meta["feed"]["producers"]):
            fd.load_stream(dirpath, st)

    def load_stream(self, dirpath, meta):
        import ctypes as C
        import os

        if getattr(self, "_feeds", None):
            return self._load_stream_fed(dirpath, meta)
        assert meta["buffer_size"] == self.buffer_size
        self._pending, self._pending_rows, self._pend_cols = [], 0, None
        self._frames = self._layout = self._store = None
        self.buffer_index = self.buffer_counter = 0
        n = int(meta["buffer_counter"])
        if meta["columns"] and n > 0:
            layout = [tuple(x) for x in meta["layout"]]
            by_name = {c["name"]: c for c in meta["columns"]}
            row_bytes = sum(int(np.prod(c["shape"])) * np.dtype(c["dtype"]).itemsize for c in meta["columns"])
            rows_per = max(1, self._STREAM_CHUNK_BYTES // max(1, row_bytes))
            files = {name: open(os.path.join(dirpath, c["file"]), "rb") for name, c in by_name.items()}
            keep, self.defer_rows = self.defer_rows, 0  # straight into the ring, never held on the host
            try:
                for o in range(0, n, rows_per):
                    m = min(rows_per, n - o)
                    cols = {}
                    for key, sub, name in layout:
                        c = by_name[name]
                        a = np.frombuffer(files[name].read(m * int(np.prod(c["shape"])) * np.dtype(c["dtype"]).itemsize), dtype=np.dtype(c["dtype"]))
                        a = a.reshape((m,) + tuple(c["shape"]))
                        if sub is None:
                            cols[key] = a
                        else:
                            cols.setdefault(key, []).append(a)
                    ReplayBuffer.store_soa(self, cols)
            finally:
                self.defer_rows = keep
                for f in files.values():
                    f.close()
            assert not self._pending
        self.buffer_index = int(meta["buffer_index"])
        self.buffer_counter = n
        if self._store is not None:  # ring position of the device store follows the host counters
            self._store.lib.jh_store_clear(self._store.h)
            self._store.lib.jh_store_set_position(self._store.h, C.c_int64(self.buffer_index), C.c_int64(self.buffer_counter))

    @property
    def size(self):
        return self.buffer_counter
