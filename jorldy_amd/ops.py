"""Torch-facing wrappers over the C ABI (device memory, streams: plumbing only).

Every function enqueues HIP kernels from libjorldy_hip.so on torch's current stream and
returns torch tensors that alias the outputs.  Nothing here computes on the CPU.
"""
import contextlib
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_DT = {torch.uint8: L.JH_U8, torch.float32: L.JH_F32, torch.int64: L.JH_I64, torch.float64: L.JH_F64, torch.int32: L.JH_I32, torch.bool: L.JH_U8}
_NP_DT = {np.dtype("uint8"): L.JH_U8, np.dtype("float32"): L.JH_F32, np.dtype("int64"): L.JH_I64, np.dtype("float64"): L.JH_F64, np.dtype("int32"): L.JH_I32, np.dtype("bool"): L.JH_U8}
_TORCH_OF = {L.JH_U8: torch.uint8, L.JH_F32: torch.float32, L.JH_I64: torch.int64, L.JH_F64: torch.float64, L.JH_I32: torch.int32}
_NP_OF = {L.JH_U8: np.uint8, L.JH_F32: np.float32, L.JH_I64: np.int64, L.JH_F64: np.float64, L.JH_I32: np.int32}


# ----------------------------------------------------------------------------- kernel timing
# bench.py measures the hand-written kernels live with HIP events recorded on the stream they are
# launched on (torch's current stream).  Off by default: zero overhead in the product path.
_PROF = {"on": False, "ev": {}, "lib": False}


class _timed:
    def __init__(self, name, nbytes, bound="hbm"):
        self.name, self.nbytes, self.bound = name, nbytes, bound

    def __enter__(self):
        if _PROF["on"]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if _PROF["on"]:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _PROF["ev"].setdefault(self.name, []).append((self.e0, e1, self.nbytes, self.bound))


def profile_reset(enable):
    _PROF["on"] = bool(enable)
    _PROF["ev"] = {}


def profile_collect():
    """-> {kernel: (n_launches, total_ms, algorithmic_bytes_per_launch, bound)}"""
    torch.cuda.synchronize()
    out = {}
    for name, lst in _PROF["ev"].items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in lst)
        out[name] = (len(lst), ms, sum(x[2] for x in lst) / len(lst), lst[0][3])
    return out


def lib_profile(enable, repeat=1):
    """Bracket every kernel launched by libjorldy_hip with HIP events on its launch stream.  repeat > 1: the
    idempotent MFMA kernels run `repeat` times back to back inside one event pair (amortises the pair's ~4 us)."""
    _PROF["lib"] = bool(enable)
    L.check(L.load().jh_prof_enable(int(max(1, repeat)) if enable else 0))


def lib_profile_calibrate(n=64):
    """Record n empty event pairs (reported as "__event_pair_overhead")."""
    L.check(L.load().jh_prof_calibrate(int(n), L.stream_ptr()))


def lib_profile_report():
    """-> {kernel name: (launches, total_ms, total_flops)} (synchronises the device); flops only for the MFMA
    kernels that declare them, else 0."""
    buf = C.create_string_buffer(1 << 16)
    L.check(L.load().jh_prof_report(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms, work = line.split("\t")
        out[name] = (int(n), float(ms), float(work))
    return out


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, "expected a float32 CUDA tensor"
    return t.contiguous()


def _dev(t):
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


# ============================================================================= store
class DeviceStore:
    """GPU-resident SoA ring of transitions (jh_store_*)."""

    def __init__(self, capacity, columns, device=None):
        """columns: list of (name, jh_dtype, elems, shape_without_batch)."""
        self.lib = L.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ctx = L.ctx(self.device.index)
        self.capacity = int(capacity)
        self.columns = list(columns)
        self.names = [c[0] for c in columns]
        descs = (L.ColDesc * len(columns))(*[L.ColDesc(int(c[1]), int(c[2])) for c in columns])
        self.h = C.c_void_p()
        L.check(self.lib.jh_store_create(self.ctx, self.capacity, len(columns), descs, C.byref(self.h)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_store_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def size(self):
        return int(self.lib.jh_store_size(self.h))

    @property
    def index(self):
        return int(self.lib.jh_store_index(self.h))

    def clear(self):
        self.lib.jh_store_clear(self.h)

    def push(self, cols):
        """cols: dict name -> numpy array [n, ...] (any float/int dtype; converted to the stored dtype)."""
        n = None
        arrs = []
        for name, dt, elems, _ in self.columns:
            src = np.asarray(cols[name]).reshape(len(cols[name]), -1)
            a = np.ascontiguousarray(src, dtype=_NP_OF[dt])
            if dt == L.JH_I64 and src.dtype.kind == "f" and not np.array_equal(a, src):
                raise ValueError(f"column {name} is stored as int64 (its first batch was integer-typed) but this batch holds fractional values")
            assert a.shape[1] == elems, f"column {name}: expected {elems} elems, got {a.shape[1]}"
            n = a.shape[0] if n is None else n
            assert a.shape[0] == n
            arrs.append(a)
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        L.check(self.lib.jh_store_push(self.h, n, ptrs, L.stream_ptr()))
        return n

    def write_rows(self, slots, cols):
        """Positional writes: row i of cols (dict name -> numpy [n, ...]) lands in slot slots[i]; the ring position
        is not touched (frame pool of the de-duplicated image replay)."""
        slots = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        arrs = []
        for name, dt, elems, _ in self.columns:
            a = np.ascontiguousarray(np.asarray(cols[name]).reshape(len(cols[name]), -1), dtype=_NP_OF[dt])
            assert a.shape == (slots.size, elems)
            arrs.append(a)
        for o in range(0, slots.size, 32768):  # grid.y limit
            m = min(32768, slots.size - o)
            sub = (C.c_void_p * len(arrs))(*[a[o : o + m].ctypes.data for a in arrs])
            L.check(self.lib.jh_store_write_rows(self.h, m, slots[o : o + m].ctypes.data, sub, L.stream_ptr()))

    def push_device(self, cols, n):
        """cols: dict name -> CUDA tensor [n, ...] already in the stored dtype (device-to-device ring append)."""
        ts = []
        for name, dt, elems, _ in self.columns:
            t = cols[name].contiguous()
            assert t.is_cuda and t.dtype == _TORCH_OF[dt] and t.numel() >= n * elems
            ts.append(t)
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        L.check(self.lib.jh_store_push_device(self.h, int(n), ptrs, L.stream_ptr()))

    def stage(self, n):
        """Zero-copy push: returns dict name -> numpy view [n, elems] of PINNED memory; call commit() after filling."""
        ptrs = (C.c_void_p * len(self.columns))()
        L.check(self.lib.jh_store_stage_begin(self.h, n, ptrs))
        out = {}
        for (name, dt, elems, _), p in zip(self.columns, ptrs):
            buf = (C.c_char * (n * elems * np.dtype(_NP_OF[dt]).itemsize)).from_address(p)
            out[name] = np.frombuffer(buf, dtype=_NP_OF[dt]).reshape(n, elems)
        return out

    def commit(self):
        L.check(self.lib.jh_store_stage_commit(self.h, L.stream_ptr()))

    def column(self, name):
        """Zero-copy torch view of a whole device column [capacity, *shape] (the lib owns the memory)."""
        i = self.names.index(name)
        _, dt, elems, shape = self.columns[i]
        p = self.lib.jh_store_col_ptr(self.h, i)
        return _wrap_device(p, (self.capacity,) + tuple(shape), _TORCH_OF[dt], self.device, owner=self)

    def gather(self, idx, names=None, as_float=True, idx_offset=0, out=None):
        """idx: int64 CUDA tensor [B] -> dict name -> tensor [B, *shape]; float32 (as_tensor semantics)
        unless as_float=False (stored dtype, e.g. uint8 frames)."""
        names = self.names if names is None else names
        B = int(idx.numel())
        sel, outs, odt = [], [], []
        for nm in names:
            i = self.names.index(nm)
            _, dt, elems, shape = self.columns[i]
            keep = (not as_float) if not isinstance(as_float, dict) else (not as_float.get(nm, True))
            tdt = _TORCH_OF[dt] if keep else torch.float32
            if out is not None:  # static output buffers (graph capture / no allocator traffic)
                o_t = out[nm]
                assert o_t.dtype == tdt and o_t.is_contiguous() and o_t.numel() == B * elems
                outs.append(o_t)
            else:
                outs.append(torch.empty((B,) + tuple(shape), dtype=tdt, device=self.device))
            sel.append(i)
            odt.append(_DT[tdt])
        selc = (C.c_int32 * len(sel))(*sel)
        odtc = (C.c_int32 * len(sel))(*odt)
        ptrs = (C.c_void_p * len(sel))(*[o.data_ptr() for o in outs])
        assert idx.dtype == torch.int64 and idx.is_cuda
        nbytes = sum(B * self.columns[i][2] * np.dtype(_NP_OF[self.columns[i][1]]).itemsize + o.numel() * o.element_size() for i, o in zip(sel, outs)) + 8 * B
        with _timed("jh_gather_kernel", nbytes):
            L.check(self.lib.jh_store_gather(self.h, B, L.ptr(idx.contiguous()), int(idx_offset), len(sel), selc, ptrs, odtc, L.stream_ptr()))
        return dict(zip(names, outs))


class _Owner:
    pass


def _wrap_device(ptr_value, shape, dtype, device, owner=None):
    """torch tensor aliasing raw device memory owned by the library (via __cuda_array_interface__)."""
    typestr = {torch.uint8: "|u1", torch.float32: "<f4", torch.int64: "<i8", torch.float64: "<f8", torch.int32: "<i4"}[dtype]
    holder = _Owner()
    holder.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr_value), False), "version": 2}
    holder._owner = owner
    with torch.cuda.device(device):
        return torch.as_tensor(holder, device=device)


# ============================================================================= PER sum tree
class ActorFeed:
    """jh_feed_*: per-tick frame-stack de-duplication + Ape-X n-step assembly with actor-side priorities for N lockstep
    actors, entirely in HBM (core/env/atari.py:145-149 + core/agent/ape_x.py:174-199; csrc/jh_feed.hip)."""

    def __init__(self, n_actors, channels, plane_bytes, n_step, gamma, planes_per_actor, window_ticks, device=None):
        self.lib = L.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.N, self.C, self.n, self.plane = int(n_actors), int(channels), int(n_step), int(plane_bytes)
        self.h = C.c_void_p()
        L.check(self.lib.jh_feed_create(L.ctx(self.device.index), self.N, self.C, self.plane, self.n, float(gamma), int(planes_per_actor),
                                        int(window_ticks), C.byref(self.h)))
        self._rew = np.empty(self.N, np.float32)
        self._done = np.empty(self.N, np.float32)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_feed_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tick(self, obs, prev_obs, pool, action, q, reward, done, out, prio_eps=0.0):
        """obs / prev_obs uint8 [N, C, ...] (device; prev_obs None on the first tick), pool = the plane pool column
        (device), action int64 [N] / q float32 [N] (device), reward / done: host arrays [N].  out: dict of preallocated
        device tensors state int64 [N, C], next_state int64 [N, C], action int64 [N], reward float32 [N, n],
        done uint8 [N, n], priority float64 [N].  Returns the number of transitions emitted (0 or N), enqueued on the
        current stream."""
        for t in (obs, pool, action, q):
            assert t.is_cuda and t.is_contiguous()
        assert obs.dtype == torch.uint8 and action.dtype == torch.int64 and q.dtype == torch.float32
        assert out["state"].dtype == torch.int64 and out["reward"].dtype == torch.float32 and out["done"].dtype == torch.uint8 and out["priority"].dtype == torch.float64
        np.copyto(self._rew, np.asarray(reward, dtype=np.float32).reshape(-1))
        np.copyto(self._done, np.asarray(done, dtype=np.float32).reshape(-1))
        emitted = C.c_int32(0)
        L.check(self.lib.jh_feed_tick(self.h, L.ptr(obs), L.ptr(prev_obs), L.ptr(pool), L.ptr(action), L.ptr(q), L.ptr(self._rew), L.ptr(self._done),
                                      float(prio_eps), L.ptr(out["state"]), L.ptr(out["next_state"]), L.ptr(out["action"]), L.ptr(out["reward"]),
                                      L.ptr(out["done"]), L.ptr(out["priority"]), C.byref(emitted), L.stream_ptr()))
        return int(emitted.value)

    def save_state(self):
        """The feed's own state (device block + tick) as bytes (jh_feed_save; synchronises the current stream)."""
        n = int(self.lib.jh_feed_state_bytes(self.h))
        buf = np.empty(n, np.uint8)
        L.check(self.lib.jh_feed_save(self.h, L.ptr(buf), n, L.stream_ptr()))
        return buf

    def load_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        L.check(self.lib.jh_feed_load(self.h, L.ptr(buf), int(buf.size), L.stream_ptr()))

    def push_stacks(self, obs, prev_obs, pool):
        """First half of a tick, stack mode (see tick)."""
        L.check(self.lib.jh_feed_push_stacks(self.h, L.ptr(obs), L.ptr(prev_obs), L.ptr(pool), L.stream_ptr()))

    def push_frames(self, frames, reset, pool, stack_out):
        """First half of a tick, frame mode: frames uint8 [N, ...] (device) = the newest plane of every actor, reset: host
        flags [N]; stack_out uint8 [N, C, ...] (device) <- the rebuilt stacks for the acting forward."""
        assert frames.is_cuda and frames.is_contiguous() and frames.dtype == torch.uint8 and stack_out.is_cuda and stack_out.is_contiguous()
        r = np.ascontiguousarray(np.asarray(reset).reshape(-1), dtype=np.uint8)
        assert r.size == self.N
        L.check(self.lib.jh_feed_push_frames(self.h, L.ptr(frames), L.ptr(r), L.ptr(pool), L.ptr(stack_out), L.stream_ptr()))

    def emit(self, action, q, reward, done, out, prio_eps=0.0):
        """Second half of a tick (see tick for the arguments); returns the number of transitions emitted (0 or N)."""
        assert action.is_cuda and action.dtype == torch.int64 and q.is_cuda and q.dtype == torch.float32
        np.copyto(self._rew, np.asarray(reward, dtype=np.float32).reshape(-1))
        np.copyto(self._done, np.asarray(done, dtype=np.float32).reshape(-1))
        emitted = C.c_int32(0)
        L.check(self.lib.jh_feed_emit(self.h, L.ptr(action), L.ptr(q), L.ptr(self._rew), L.ptr(self._done), float(prio_eps), L.ptr(out["state"]),
                                      L.ptr(out["next_state"]), L.ptr(out["action"]), L.ptr(out["reward"]), L.ptr(out["done"]), L.ptr(out["priority"]),
                                      C.byref(emitted), L.stream_ptr()))
        return int(emitted.value)

    def state(self):
        """(flags, planes written so far); blocking."""
        fl, pw = C.c_int32(0), C.c_int64(0)
        L.check(self.lib.jh_feed_state(self.h, C.byref(fl), C.byref(pw), L.stream_ptr()))
        return int(fl.value), int(pw.value)


class SumTree:
    """Device float64 sum tree (jh_per_*), bit-identical to per_buffer.py's numpy tree."""

    def __init__(self, capacity, uniform_sample_prob=1e-3, device=None):
        self.lib = L.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ctx = L.ctx(self.device.index)
        self.capacity = int(capacity)
        self.usp = float(uniform_sample_prob)
        self.h = C.c_void_p()
        L.check(self.lib.jh_per_create(self.ctx, self.capacity, self.usp, C.byref(self.h)))
        # {sampled_p, mean_p, root, max_w} of the last sample: device-mapped pinned memory, so that a learner can read
        # them without a copy + stream sync once it knows the sampling kernel has finished (`stats_np`)
        import os

        if os.environ.get("JH_MAPPED_STATS", "1") == "1":
            self._stats_pin = PinnedBuffer((4,), np.float64, self.device.index)
            self._stats_pin.np[:] = 0
            self._stats = _wrap_device(self._stats_pin.dev_ptr.value, (4,), torch.float64, self.device, owner=self._stats_pin)
            self.stats_np = self._stats_pin.np
        else:
            self._stats, self.stats_np = torch.zeros(4, dtype=torch.float64, device=self.device), None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_per_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def tree_size(self):
        return 2 * self.capacity - 1

    def push(self, n, priorities=None):
        p = None if priorities is None else np.ascontiguousarray(priorities, dtype=np.float64).reshape(-1)
        if p is not None:
            assert p.size == n
        L.check(self.lib.jh_per_push(self.h, int(n), L.ptr(p), L.stream_ptr()))

    def push_device(self, n, priorities):
        """push() with float64 priorities that are already on the device (ActorFeed's actor-side priorities)."""
        assert priorities.is_cuda and priorities.dtype == torch.float64 and priorities.is_contiguous() and priorities.numel() >= n
        L.check(self.lib.jh_per_push_device(self.h, int(n), L.ptr(priorities), L.stream_ptr()))

    def update(self, idx, prio):
        assert idx.dtype == torch.int64 and idx.is_cuda and prio.is_cuda
        dt = L.JH_F32 if prio.dtype == torch.float32 else L.JH_F64
        assert prio.dtype in (torch.float32, torch.float64)
        L.check(self.lib.jh_per_update(self.h, int(idx.numel()), L.ptr(idx.contiguous()), L.ptr(prio.contiguous()), dt, L.stream_ptr()))

    def sample(self, beta, uniform_slot, u, want_w64=True, out_idx=None, out_w32=None):
        """uniform_slot int64[n_uni] (numpy), u float64[B-n_uni] (numpy).  Returns (idx i64[B], w64|None, w32, stats f64[4]) on device.
        out_idx / out_w32: preallocated outputs (static buffers of a captured graph)."""
        uniform_slot = np.ascontiguousarray(uniform_slot, dtype=np.int64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        B = uniform_slot.size + u.size
        idx = torch.empty(B, dtype=torch.int64, device=self.device) if out_idx is None else out_idx
        w64 = torch.empty(B, dtype=torch.float64, device=self.device) if want_w64 else None
        w32 = torch.empty(B, dtype=torch.float32, device=self.device) if out_w32 is None else out_w32
        assert idx.numel() == B and w32.numel() == B
        L.check(self.lib.jh_per_sample(self.h, B, float(beta), int(uniform_slot.size), L.ptr(uniform_slot), L.ptr(u), L.ptr(idx), L.ptr(w64), L.ptr(w32), L.ptr(self._stats), L.stream_ptr()))
        return idx, w64, w32, self._stats

    def shard_stats(self, B, out3):
        """out3 (float64 [3], device) <- {root, count, min priority of the last sample of B}."""
        L.check(self.lib.jh_per_shard_stats(self.h, int(B), L.ptr(out3), L.stream_ptr()))

    def weights_sharded(self, B, beta, all3, w32):
        """IS weights of the last sample against the logical buffer described by the gathered triples all3 [G, 3]."""
        L.check(self.lib.jh_per_weights_sharded(self.h, int(B), float(beta), L.ptr(all3), int(all3.numel() // 3), None, L.ptr(w32), L.stream_ptr()))

    def view(self):
        """Zero-copy float64 [2N-1] torch view of the device tree (the library owns the memory)."""
        return _wrap_device(self.lib.jh_per_tree_ptr(self.h), (self.tree_size,), torch.float64, self.device, owner=self)

    def state(self):
        mp, root, ti, cnt = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
        L.check(self.lib.jh_per_state(self.h, C.byref(mp), C.byref(root), C.byref(ti), C.byref(cnt), L.stream_ptr()))
        return dict(max_priority=mp.value, root=root.value, tree_index=ti.value, counter=cnt.value)

    def dump(self):
        out = np.empty(self.tree_size, dtype=np.float64)
        L.check(self.lib.jh_per_dump(self.h, L.ptr(out), L.stream_ptr()))
        return out

    def load(self, tree, max_priority, tree_index, counter):
        t = np.ascontiguousarray(tree, dtype=np.float64)
        assert t.size == self.tree_size
        L.check(self.lib.jh_per_load(self.h, L.ptr(t), float(max_priority), int(tree_index), int(counter)))


# ============================================================================= PPO math
def gae(reward, done, value, next_value, n_step, gamma, lam, standardize=True, out=None):
    """ppo.py:95-110.  Inputs float32 CUDA [M,1] (or [M]), M = W*n_step, worker-major.
    Returns (adv [M,1], ret [M,1]); `out` = preallocated (adv, ret)."""
    lib = L.load()
    r, d, v, vn = (_f32(x).reshape(-1) for x in (reward, done, value, next_value))
    M = r.numel()
    assert M % n_step == 0
    if out is None:
        adv = torch.empty(M, dtype=torch.float32, device=r.device)
        ret = torch.empty(M, dtype=torch.float32, device=r.device)
    else:
        adv, ret = (_f32(t).reshape(-1) for t in out)
    with _timed("jh_gae_kernel", 24 * M):  # 4 reads + 2 writes of fp32 per transition
      L.check(lib.jh_gae(L.ctx(_dev(r)), M // n_step, int(n_step), float(gamma), float(lam), L.ptr(r), L.ptr(d), L.ptr(v), L.ptr(vn), L.ptr(adv), L.ptr(ret), int(bool(standardize)), L.stream_ptr()))
    return adv.view(-1, 1), ret.view(-1, 1)


def mean_into(x, out):
    """out[0] = mean(x) (ppo.py:112), one deterministic workgroup."""
    x = _f32(x).reshape(-1)
    L.check(L.load().jh_mean_f32(L.ctx(_dev(x)), int(x.numel()), L.ptr(x), L.ptr(out), L.stream_ptr()))
    return out


class MinibatchRows:
    """jh_ppo_minibatch_rows: the `x[idx]` gathers of every epoch of ppo.py:118-125 done once per learn().
    srcs / dsts: lists of float32 CUDA tensors [M, e_c] / [n, e_c] (static addresses: capturable)."""

    def __init__(self, srcs, dsts):
        self.lib = L.load()
        assert len(srcs) == len(dsts) <= 8
        self.srcs, self.dsts = [_f32(t) for t in srcs], list(dsts)
        n = len(srcs)
        self.n = int(dsts[0].shape[0])
        el = [int(t.numel() // t.shape[0]) for t in self.srcs]
        assert all(int(d.numel()) == self.n * e and d.is_contiguous() for d, e in zip(self.dsts, el))
        self._elems = (C.c_int32 * n)(*el)
        self._src = (C.c_void_p * n)(*[t.data_ptr() for t in self.srcs])
        self._dst = (C.c_void_p * n)(*[t.data_ptr() for t in self.dsts])
        self.ctx = L.ctx(_dev(self.srcs[0]))

    def __call__(self, idx):
        assert idx.dtype == torch.int64 and int(idx.numel()) == self.n
        L.check(self.lib.jh_ppo_minibatch_rows(self.ctx, self.n, L.ptr(idx), len(self.srcs), self._elems, self._src, self._dst, L.stream_ptr()))
        return self.dsts


def logp_discrete(logits, action, out=None):
    lib = L.load()
    z = _f32(logits)
    M, A = z.shape
    a = _f32(action).reshape(-1)
    out = torch.empty(M, dtype=torch.float32, device=z.device) if out is None else _f32(out).reshape(-1)
    L.check(lib.jh_logp_discrete(L.ctx(_dev(z)), M, A, L.ptr(z), L.ptr(a), L.ptr(out), L.stream_ptr()))
    return out.view(-1, 1)


def logp_continuous(mu_raw, log_std_raw, action, out=None):
    lib = L.load()
    mu, ls, a = _f32(mu_raw), _f32(log_std_raw), _f32(action)
    M, A = mu.shape
    out = torch.empty(M, A, dtype=torch.float32, device=mu.device) if out is None else _f32(out)
    L.check(lib.jh_logp_continuous(L.ctx(_dev(mu)), M, A, L.ptr(mu), L.ptr(ls), L.ptr(a), L.ptr(out), L.stream_ptr()))
    return out


def ppo_loss_discrete(logits, value_pred, idx, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef, stats=None):
    """Returns (grad_logits [B,A], grad_value [B,1], stats f32[8])."""
    lib = L.load()
    z, v = _f32(logits), _f32(value_pred).reshape(-1)
    B, A = z.shape
    g_z = torch.empty_like(z)
    g_v = torch.empty(B, dtype=torch.float32, device=z.device)
    if stats is None:
        stats = torch.empty(8, dtype=torch.float32, device=z.device)
    with _timed("jh_ppo_fused_kernel" if B <= 1024 else "jh_ppo_fwd+bwd", 4 * B * (2 * A + 7) + 8 * B):  # (A+6) reads + (A+1) writes + idx
      L.check(lib.jh_ppo_loss_discrete(L.ctx(_dev(z)), B, A, L.ptr(z), L.ptr(v), L.ptr(idx), L.ptr(_f32(action).reshape(-1)), L.ptr(_f32(adv).reshape(-1)), L.ptr(_f32(ret).reshape(-1)), L.ptr(_f32(value_old).reshape(-1)), L.ptr(_f32(logp_old).reshape(-1)), float(eps_clip), float(vf_coef), float(ent_coef), L.ptr(g_z), L.ptr(g_v), L.ptr(stats), L.stream_ptr()))
    return g_z, g_v.view(-1, 1), stats


def ppo_loss_continuous(mu_raw, log_std_raw, value_pred, idx, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef, stats=None):
    lib = L.load()
    mu, ls, v = _f32(mu_raw), _f32(log_std_raw), _f32(value_pred).reshape(-1)
    B, A = mu.shape
    g_mu, g_ls = torch.empty_like(mu), torch.empty_like(ls)
    g_v = torch.empty(B, dtype=torch.float32, device=mu.device)
    if stats is None:
        stats = torch.empty(8, dtype=torch.float32, device=mu.device)
    L.check(lib.jh_ppo_loss_continuous(L.ctx(_dev(mu)), B, A, L.ptr(mu), L.ptr(ls), L.ptr(v), L.ptr(idx), L.ptr(_f32(action)), L.ptr(_f32(adv).reshape(-1)), L.ptr(_f32(ret).reshape(-1)), L.ptr(_f32(value_old).reshape(-1)), L.ptr(_f32(logp_old)), float(eps_clip), float(vf_coef), float(ent_coef), L.ptr(g_mu), L.ptr(g_ls), L.ptr(g_v), L.ptr(stats), L.stream_ptr()))
    return g_mu, g_ls, g_v.view(-1, 1), stats


def ppo_loss_dp(head0, head1, value_pred, idx, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef, stats, reduce_mean, work):
    """ppo_loss_discrete (head1 None) / ppo_loss_continuous for data-parallel learners with the global critic branch (jh_ppo_loss_deferred ->
    reduce_mean(critic sums) -> jh_ppo_critic_select_rows).  work: fp32 device tensor of >= B + 16 elements (second branch's value gradients,
    the two sums, the local statistics row).  Returns the head gradients like the plain functions."""
    lib = L.load()
    cont = head1 is not None
    h0, v = _f32(head0), _f32(value_pred).reshape(-1)
    h1 = _f32(head1) if cont else None
    B, A = h0.shape
    g0 = torch.empty_like(h0)
    g1 = torch.empty_like(h1) if cont else None
    g_v = torch.empty(B, dtype=torch.float32, device=h0.device)
    dv2, sums, local = work[:B], work[B : B + 2], work[B + 8 : B + 16]
    ctx = L.ctx(_dev(h0))
    L.check(lib.jh_ppo_loss_deferred(ctx, int(cont), B, A, L.ptr(h0), L.ptr(h1), L.ptr(v), L.ptr(idx), L.ptr(_f32(action) if cont else _f32(action).reshape(-1)),
                                     L.ptr(_f32(adv).reshape(-1)), L.ptr(_f32(ret).reshape(-1)), L.ptr(_f32(value_old).reshape(-1)),
                                     L.ptr(_f32(logp_old) if cont else _f32(logp_old).reshape(-1)), float(eps_clip), float(vf_coef), float(ent_coef),
                                     L.ptr(g0), L.ptr(g1), L.ptr(g_v), L.ptr(dv2), L.ptr(sums), L.ptr(local), L.stream_ptr()))
    reduce_mean(sums)
    L.check(lib.jh_ppo_critic_select_rows(ctx, B, L.ptr(sums), float(vf_coef), float(ent_coef), L.ptr(g_v), L.ptr(dv2), L.ptr(local), L.ptr(stats), L.stream_ptr()))
    return (g0, g1, g_v.view(-1, 1)) if cont else (g0, g_v.view(-1, 1))


def ppo_loss_packed(heads, A, idx, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef, grad_out, stats, continuous=False, reduce_mean=None, work=None):
    """The PPO losses (ppo.py:122-165) on a network whose last layer stacks the heads (jh_ppo_loss_packed): heads float32 [B, ld] = (head0 [A] | head1 [A]
    (continuous) | value | padding), grad_out the same shape -- d(loss)/d(heads), padding columns untouched (keep them zero).  idx None: the per-row
    inputs are already the minibatch's rows.  reduce_mean + work (>= B + 16 floats): data-parallel learners, the critic branch of the GLOBAL minibatch
    (ppo_loss_dp's scheme)."""
    lib = L.load()
    assert heads.dtype == torch.float32 and heads.is_contiguous() and grad_out.is_contiguous() and grad_out.shape == heads.shape
    B, ld = int(heads.shape[0]), int(heads.shape[1])
    ctx = L.ctx(_dev(heads))
    act = _f32(action) if continuous else _f32(action).reshape(-1)
    lpo = _f32(logp_old) if continuous else _f32(logp_old).reshape(-1)
    dv2 = sums = local = None
    if reduce_mean is not None:
        dv2, sums, local = work[:B], work[B : B + 2], work[B + 8 : B + 16]
    L.check(lib.jh_ppo_loss_packed(ctx, int(bool(continuous)), B, int(A), L.ptr(heads), ld, L.ptr(idx), L.ptr(act), L.ptr(_f32(adv).reshape(-1)), L.ptr(_f32(ret).reshape(-1)),
                                   L.ptr(_f32(value_old).reshape(-1)), L.ptr(lpo), float(eps_clip), float(vf_coef), float(ent_coef), L.ptr(grad_out), L.ptr(dv2), L.ptr(sums),
                                   L.ptr(local if reduce_mean is not None else stats), L.stream_ptr()))
    if reduce_mean is not None:
        reduce_mean(sums)
        nv = (2 * A if continuous else A)
        gv = grad_out.view(-1)[nv:]  # the value column, row stride ld
        L.check(lib.jh_ppo_critic_select_strided(ctx, B, L.ptr(sums), float(vf_coef), float(ent_coef), L.ptr(gv), ld, L.ptr(dv2), L.ptr(local), L.ptr(stats), L.stream_ptr()))
    return grad_out


def heads_unpack(packed, A, h0, h1, value):
    """packed [rows, ld] -> h0 [rows, A], h1 [rows, A] (or None), value [rows]: the layout jh_logp_* / jh_gae read."""
    assert packed.is_contiguous() and h0.is_contiguous() and value.is_contiguous() and (h1 is None or h1.is_contiguous())
    L.check(L.load().jh_heads_unpack(L.ctx(_dev(packed)), int(packed.shape[0]), int(A), L.ptr(packed), int(packed.shape[1]), L.ptr(h0), L.ptr(h1), L.ptr(value), L.stream_ptr()))


def policy_act_discrete(heads, A, seed, counter, training, out):
    """ppo.py:63-69 on device-resident heads [W, ld]: out int64 [W] (device / device-mapped) <- sampled (training) or greedy actions."""
    assert heads.is_contiguous() and out.dtype == torch.int64
    L.check(L.load().jh_policy_act_discrete(L.ctx(_dev(heads)), int(heads.shape[0]), int(A), L.ptr(heads), int(heads.shape[1]), int(seed) & (2**64 - 1), int(counter) & (2**64 - 1),
                                            int(bool(training)), L.ptr(out), L.stream_ptr()))
    return out


# ============================================================================= native policy-value MLP
class PinnedBuffer:
    """Pinned host memory mapped into the device address space (jh_pinned_alloc): `.np` is the host
    view, `.dev_ptr` the address kernels use."""

    def __init__(self, shape, dtype, device_index=None):
        self.lib = L.load()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        h, d = C.c_void_p(), C.c_void_p()
        L.check(self.lib.jh_pinned_alloc(L.ctx(device_index), max(n, 8), C.byref(h), C.byref(d)))
        self._h = h
        self.dev_ptr = C.c_void_p(d.value)
        buf = (C.c_char * max(n, 8)).from_address(h.value)
        self.np = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.np = None
                self.lib.jh_pinned_free(self._h)
                self._h = None
        except Exception:
            pass


class PPONet:
    """jh_pponet_*: S -> H relu -> H relu -> heads, fwd / bwd / clip + Adam on flat fp32 buckets."""

    def __init__(self, S, H, A, continuous, max_rows, device, seed=0):
        self.lib = L.load()
        self.device = torch.device(device)
        self.ctx = L.ctx(self.device.index)
        self.S, self.H, self.A, self.cont, self.max_rows = int(S), int(H), int(A), bool(continuous), int(max_rows)
        self.n_params = int(self.lib.jh_pponet_param_count(self.S, self.H, self.A, int(self.cont)))
        mk = lambda: torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.params, self.grads, self.m, self.v = mk(), mk(), mk(), mk()
        self.h = C.c_void_p()
        L.check(self.lib.jh_pponet_create(self.ctx, self.S, self.H, self.A, int(self.cont), self.max_rows, L.ptr(self.params), L.ptr(self.grads), L.ptr(self.m), L.ptr(self.v), C.c_uint64(int(seed)), C.byref(self.h)))
        self._act = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_pponet_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_hyper(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, step=-1.0):
        """step >= 0 sets Adam's step counter (0 = fresh optimizer), step < 0 keeps it."""
        L.check(self.lib.jh_pponet_set_hyper(self.h, float(lr), float(beta1), float(beta2), float(eps), float(step), L.stream_ptr()))

    def set_lr(self, lr):
        L.check(self.lib.jh_pponet_set_lr(self.h, float(lr), L.stream_ptr()))

    def hyper_ptr(self):
        return int(self.lib.jh_pponet_hyper_ptr(self.h))

    def act_rng(self, state=None):
        """(seed, counter) of the host-side sampling stream; state=(seed, counter) restores it."""
        s, c = C.c_uint64(0), C.c_uint64(0)
        if state is not None:
            s, c = C.c_uint64(int(state[0])), C.c_uint64(int(state[1]))
        L.check(self.lib.jh_pponet_act_rng(self.h, C.byref(s), C.byref(c), int(state is not None)))
        return int(s.value), int(c.value)

    def forward(self, x, idx=None, B=None, out=None):
        """x float32 [*, S]; rows gathered by idx (int64 [B]) when given.  Returns raw heads
        (logits, value) or (mu_raw, log_std_raw, value); `out` = preallocated tuple."""
        B = int(idx.numel()) if idx is not None else (int(x.shape[0]) if B is None else int(B))
        if out is None:
            h0 = torch.empty(B, self.A, dtype=torch.float32, device=self.device)
            h1 = torch.empty(B, self.A, dtype=torch.float32, device=self.device) if self.cont else None
            val = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        else:
            h0, h1, val = out
        flops = 2.0 * B * self.H * (self.S + self.H + (2 * self.A + 1 if self.cont else self.A + 1))
        with _timed("jh_pponet_forward", flops, "mfma"):
            L.check(self.lib.jh_pponet_forward(self.h, B, L.ptr(_f32(x)), L.ptr(idx), L.ptr(h0), L.ptr(h1), L.ptr(val), L.stream_ptr()))
        return (h0, h1, val) if self.cont else (h0, val)

    def backward(self, x, idx, g_head0, g_head1, g_value, B=None):
        B = int(idx.numel()) if idx is not None else (int(x.shape[0]) if B is None else int(B))
        flops = 4.0 * B * self.H * (self.S + self.H + (2 * self.A + 1 if self.cont else self.A + 1))
        with _timed("jh_pponet_backward", flops, "mfma"):
            L.check(self.lib.jh_pponet_backward(self.h, B, L.ptr(_f32(x)), L.ptr(idx), L.ptr(g_head0), L.ptr(g_head1), L.ptr(g_value), L.stream_ptr()))

    def adam_step(self, max_norm, norm_out=None):
        with _timed("jh_gradnorm+adam", 4.0 * self.n_params * 8):  # g (r twice, w) + p,m,v (r+w)
            L.check(self.lib.jh_pponet_adam_step(self.h, float(max_norm if max_norm else 0.0), L.ptr(norm_out), L.stream_ptr()))

    def fused_ok(self, B):
        """Minibatches the four- / five-launch update (jh_pponet_ppo_update) takes: < 1024 rows (from there on the
        LDS-tiled engine wins), hidden width a multiple of 32."""
        n_out = (2 * self.A + 1) if self.cont else (self.A + 1)
        return self.H % 32 == 0 and 0 < B < 1024 and n_out <= 8  # wider heads (round 5): the separate forward / loss / backward / Adam calls

    def ppo_update(self, x, idx, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef, max_norm, stats, do_adam=True):
        """One whole PPO minibatch update in 4 launches (5 beyond 256 rows or with do_adam=False) (jh_pponet_ppo_update).  B = idx.numel() <= 1024."""
        B = int(idx.numel()) if idx is not None else int(x.shape[0])
        L.check(self.lib.jh_pponet_ppo_update(self.h, B, L.ptr(_f32(x)), L.ptr(idx), L.ptr(_f32(action)), L.ptr(_f32(adv).reshape(-1)), L.ptr(_f32(ret).reshape(-1)),
                                              L.ptr(_f32(value_old).reshape(-1)), L.ptr(_f32(logp_old)), float(eps_clip), float(vf_coef), float(ent_coef),
                                              float(max_norm if max_norm else 0.0), int(bool(do_adam)), L.ptr(stats), L.stream_ptr()))

    def ppo_update_rows(self, x, idx, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef, max_norm, stats, do_adam=True):
        """The same update for any B <= max_rows in one call (jh_pponet_ppo_update_rows): forward, the loss forward + backward in ONE launch, backward,
        [clip + Adam].  What minibatches of >= 1024 rows / nets with more than 8 head outputs take; bit-identical to forward -> ppo_loss_* -> backward -> adam_step."""
        B = int(idx.numel()) if idx is not None else int(x.shape[0])
        L.check(self.lib.jh_pponet_ppo_update_rows(self.h, B, L.ptr(_f32(x)), L.ptr(idx), L.ptr(_f32(action)), L.ptr(_f32(adv).reshape(-1)), L.ptr(_f32(ret).reshape(-1)),
                                                   L.ptr(_f32(value_old).reshape(-1)), L.ptr(_f32(logp_old)), float(eps_clip), float(vf_coef), float(ent_coef),
                                                   float(max_norm if max_norm else 0.0), int(bool(do_adam)), L.ptr(stats), L.stream_ptr()))

    def ppo_update_dp(self, x, idx, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef, stats, reduce_mean, critic_sums, peer=None):
        """The minibatch update for data-parallel learners with the reference's critic exactly (jh_pponet_ppo_update_dp_begin / _end around
        an 8-byte all-reduce): reduce_mean(t) all-reduces a small fp32 device tensor to its mean over the ranks, in place, on the current
        stream.  Leaves a complete gradient bucket; the caller reduces it and calls adam_step."""
        B = int(idx.numel()) if idx is not None else int(x.shape[0])
        L.check(self.lib.jh_pponet_ppo_update_dp_begin(self.h, B, L.ptr(_f32(x)), L.ptr(idx), L.ptr(_f32(action)), L.ptr(_f32(adv).reshape(-1)), L.ptr(_f32(ret).reshape(-1)),
                                                       L.ptr(_f32(value_old).reshape(-1)), L.ptr(_f32(logp_old)), float(eps_clip), float(vf_coef), float(ent_coef),
                                                       L.ptr(critic_sums), L.stream_ptr()))
        if peer is not None and B <= 256:  # peer-pointer transport: the ranks' sums meet inside the second half's first launch
            L.check(self.lib.jh_pponet_ppo_update_dp_end_peer(self.h, peer, B, L.ptr(_f32(x)), L.ptr(idx), L.ptr(critic_sums), float(vf_coef), float(ent_coef), L.ptr(stats), L.stream_ptr()))
            return
        reduce_mean(critic_sums)
        L.check(self.lib.jh_pponet_ppo_update_dp_end(self.h, B, L.ptr(_f32(x)), L.ptr(idx), L.ptr(critic_sums), float(vf_coef), float(ent_coef), L.ptr(stats), L.stream_ptr()))

    # ---- acting (one launch; partial heads come back through device-mapped pinned memory) ---------
    def act_discrete(self, obs, training=True, want_logits=False):
        """obs: numpy float32 [W, S] -> numpy int64 [W, 1] (blocking)."""
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        W = int(obs.shape[0])
        act = np.empty((W, 1), np.int64)
        logits = np.empty((W, self.A), np.float32) if want_logits else None
        val = np.empty((W, 1), np.float32) if want_logits else None
        L.check(self.lib.jh_pponet_act_discrete(self.h, W, L.ptr(obs), L.ptr(act), L.ptr(logits), L.ptr(val), int(bool(training)), L.stream_ptr()))
        return (act, logits, val) if want_logits else act


class NormalSource:
    """jh_normal_fill: N(0,1) draws on the device from a counter-based generator whose call counter lives in device
    memory (a replayed hipGraph draws fresh noise).  fill(t) overwrites the float32 CUDA tensor t in place."""

    def __init__(self, device, seed=None):
        self.lib = L.load()
        self.device = torch.device(device)
        self.ctx = L.ctx(self.device.index)
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())  # tied to torch.manual_seed
        self.state = torch.tensor([int(seed), 0, 0, 0], dtype=torch.int64, device=self.device)

    def fill(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        L.check(self.lib.jh_normal_fill(self.ctx, int(t.numel()), L.ptr(t), L.ptr(self.state), L.stream_ptr()))
        return t


def value_act(logits, v_min=0.0, v_max=0.0, eps=None, u=None, rand_action=None, out=None, want_q_all=False):
    """jh_value_act: network outputs [N, A, K] (K = 1: Q values) -> (action int64 [N], q_taken float32 [N], q_all | None)
    on the device.  eps / u / rand_action: numpy float32 / float64 / int64 [N] (the host's epsilon-greedy draws) or
    all None for greedy."""
    lg = _f32(logits)
    N, A, K = (int(v) for v in lg.shape)
    dev = lg.device
    act, q = (torch.empty(N, dtype=torch.int64, device=dev), torch.empty(N, dtype=torch.float32, device=dev)) if out is None else out
    q_all = torch.empty(N, A, dtype=torch.float32, device=dev) if want_q_all else None
    if eps is not None:
        eps, u, rand_action = (np.ascontiguousarray(eps, dtype=np.float32), np.ascontiguousarray(u, dtype=np.float64), np.ascontiguousarray(rand_action, dtype=np.int64))
        assert eps.size == N and u.size == N and rand_action.size == N
    L.check(L.load().jh_value_act(L.ctx(_dev(lg)), N, A, K, L.ptr(lg), float(v_min), float(v_max), L.ptr(eps), L.ptr(u), L.ptr(rand_action), L.ptr(act), L.ptr(q),
                                  L.ptr(q_all), L.stream_ptr()))
    return act, q, q_all


def _pponet_act_continuous(self, obs, training=True, want_heads=False):
    """obs: numpy float32 [W, S] -> numpy float32 [W, A] in (-1, 1) (blocking); PPO.act, ppo.py:55-63."""
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    W = int(obs.shape[0])
    act = np.empty((W, self.A), np.float32)
    mu = np.empty((W, self.A), np.float32) if want_heads else None
    ls = np.empty((W, self.A), np.float32) if want_heads else None
    L.check(self.lib.jh_pponet_act_continuous(self.h, W, L.ptr(obs), L.ptr(act), L.ptr(mu), L.ptr(ls), None, int(bool(training)), L.stream_ptr()))
    return (act, mu, ls) if want_heads else act


PPONet.act_continuous = _pponet_act_continuous


# ============================================================================= TD / C51
def td_loss(q, q_next_target, action, reward, done, gamma, q_next_online=None, weights=None, alpha=0.0, n_step=0, stats=None):
    """Returns (grad_q [B,A], prio [B], stats f32[4] = {loss, max_Q, mean_td, 0})."""
    lib = L.load()
    q = _f32(q)
    B, A = q.shape
    flags = (L.JH_TD_DOUBLE if q_next_online is not None else 0) | (L.JH_TD_PER if weights is not None else 0)
    g = torch.empty_like(q)
    prio = torch.empty(B, dtype=torch.float32, device=q.device)
    if stats is None:
        stats = torch.empty(4, dtype=torch.float32, device=q.device)
    r, d = _f32(reward).reshape(B, -1), _f32(done).reshape(B, -1)
    assert r.shape[1] == max(n_step, 1) and d.shape == r.shape
    L.check(lib.jh_td_loss(L.ctx(_dev(q)), B, A, int(n_step), flags, L.ptr(q), L.ptr(None if q_next_online is None else _f32(q_next_online)), L.ptr(_f32(q_next_target)), L.ptr(_f32(action).reshape(-1)), L.ptr(r), L.ptr(d), L.ptr(None if weights is None else _f32(weights).reshape(-1)), float(gamma), float(alpha), L.ptr(g), L.ptr(prio), L.ptr(stats), L.stream_ptr()))
    return g, prio, stats


@contextlib.contextmanager
def graph_capture(g):
    """`with torch.cuda.graph(g, capture_error_mode="thread_local")` plus the two things this torch build leaves to the caller:

    * garbage collection.  torch.cuda.graph.__enter__ only collects when torch.compiler.config.force_cudagraph_gc is set (it is not): a dead
      reference cycle left by an EARLIER agent (agent <-> collector, owning device buffers, pinned memory, events) is then finalized by
      whichever Python allocation happens to trip the cyclic collector -- and if that is one inside the capture, the hipFree / hipHostFree /
      hipEventDestroy comes from the capturing thread and invalidates the capture (hipErrorStreamCaptureInvalidated at the next launch;
      seen in ~1 of 8 runs of the GPU suite, always in a test that follows many agent + collector pairs, never in that test alone).
      So: one full collection in front, the automatic collector off while capturing.
    * a failed capture.  torch.cuda.graph.__exit__ raises out of capture_end() BEFORE it restores the stream: the capture stream -- torch's
      process-wide default one -- stays current and stays capturing, and everything enqueued afterwards fails.  Here the previous stream
      is made current again, the abandoned capture is ended, and torch is given a fresh default capture stream; the caller's eager
      fallback then runs on a healthy stream."""
    import gc

    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    prev = torch.cuda.current_stream()
    cm = torch.cuda.graph(g, capture_error_mode="thread_local")  # other threads (batched actors, staging ring, collector) keep issuing HIP work on their own streams
    try:
        with cm:
            yield
    except BaseException:
        try:
            torch.cuda.set_stream(prev)
            L.load().jh_stream_abort_capture(C.c_void_p(cm.capture_stream.cuda_stream))
            if torch.cuda.graph.default_capture_stream is cm.capture_stream:
                torch.cuda.graph.default_capture_stream = None
        except Exception:
            pass
        raise
    finally:
        if was_enabled:
            gc.enable()


def c51_loss(logit, target_logit, action, reward, done, v_min, v_max, gamma, next_logit_online=None, weights=None, alpha=0.0, n_step=0, shift_max=False, stats=None):
    """logit/target_logit/next_logit_online [B,A,K].  Returns (grad_logit, prio [B], kl [B], stats f32[8])."""
    lib = L.load()
    z = _f32(logit)
    B, A, K = z.shape
    flags = (L.JH_C51_DOUBLE if next_logit_online is not None else 0) | (L.JH_C51_PER if weights is not None else 0) | (L.JH_C51_SHIFT_MAX if shift_max else 0)
    g = torch.empty_like(z)
    prio = torch.empty(B, dtype=torch.float32, device=z.device)
    kl = torch.empty(B, dtype=torch.float32, device=z.device)
    if stats is None:
        stats = torch.empty(8, dtype=torch.float32, device=z.device)
    r, d = _f32(reward).reshape(B, -1), _f32(done).reshape(B, -1)
    assert r.shape[1] == max(n_step, 1) and d.shape == r.shape
    L.check(lib.jh_c51_loss(L.ctx(_dev(z)), B, A, K, int(n_step), flags, L.ptr(z), L.ptr(None if next_logit_online is None else _f32(next_logit_online)), L.ptr(_f32(target_logit)), L.ptr(_f32(action).reshape(-1)), L.ptr(r), L.ptr(d), L.ptr(None if weights is None else _f32(weights).reshape(-1)), float(v_min), float(v_max), float(gamma), float(alpha), L.ptr(g), L.ptr(prio), L.ptr(kl), L.ptr(stats), L.stream_ptr()))
    return g, prio, kl, stats


# ============================================================================= host collector
class CartPoleVec:
    """W synthetic CartPole-v1 envs stepped in one native call (jh_cartpole_*)."""

    def __init__(self, W, seed=0):
        self.lib = L.load()
        self.W = int(W)
        self.h = C.c_void_p()
        L.check(self.lib.jh_cartpole_create(self.W, C.c_uint64(int(seed)), C.byref(self.h)))
        self.state_size, self.action_size, self.action_type = 4, 2, "discrete"

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_cartpole_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def obs(self, out=None):
        out = np.empty((self.W, 4), np.float32) if out is None else out
        L.check(self.lib.jh_cartpole_obs(self.h, L.ptr(out)))
        return out

    def step(self, action, next_obs=None, reward=None, done=None):
        a = np.ascontiguousarray(action, dtype=np.int64).reshape(-1)
        assert a.size == self.W
        next_obs = np.empty((self.W, 4), np.float32) if next_obs is None else next_obs
        reward = np.empty(self.W, np.float32) if reward is None else reward
        done = np.empty(self.W, np.uint8) if done is None else done
        L.check(self.lib.jh_cartpole_step(self.h, L.ptr(a), L.ptr(next_obs), L.ptr(reward), L.ptr(done)))
        return next_obs, reward, done


class ControlVec:
    """W synthetic continuous-control envs (jh_control_*: stand-in for MuJoCo at config.ppo.mujoco shapes, default
    Hopper-v3's S = 11, A = 3) stepped in one native call."""

    def __init__(self, W, state_size=11, action_size=3, seed=0):
        self.lib = L.load()
        self.W, self.state_size, self.action_size, self.action_type = int(W), int(state_size), int(action_size), "continuous"
        self.h = C.c_void_p()
        L.check(self.lib.jh_control_create(self.W, self.state_size, self.action_size, C.c_uint64(int(seed)), C.byref(self.h)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_control_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def obs(self, out=None):
        out = np.empty((self.W, self.state_size), np.float32) if out is None else out
        L.check(self.lib.jh_control_obs(self.h, L.ptr(out)))
        return out

    def step(self, action, next_obs=None, reward=None, done=None):
        a = np.ascontiguousarray(action, dtype=np.float32).reshape(self.W, self.action_size)
        next_obs = np.empty((self.W, self.state_size), np.float32) if next_obs is None else next_obs
        reward = np.empty(self.W, np.float32) if reward is None else reward
        done = np.empty(self.W, np.uint8) if done is None else done
        L.check(self.lib.jh_control_step(self.h, L.ptr(a), L.ptr(next_obs), L.ptr(reward), L.ptr(done)))
        return next_obs, reward, done


class RainbowNet:
    """jh_rbnet_*: the value networks of the DQN / Rainbow / Ape-X family -- kind "rainbow" (MLP or Nature-CNN
    head -> Linear -> noisy dueling categorical heads), "dueling" (head -> l1_a|l1_v -> l2_a, l2_v) and "q"
    (head -> l -> q) -- with the three learn() forwards, backward and the optimizer step as grouped MFMA GEMM
    launches.

    Parameters live in flat fp32 buckets in the library's private layout; `export_state` / `import_state`
    convert to and from the reference's `state_dict` (network/{rainbow,dueling,q_network}.py key names, shapes
    and order), so checkpoints and sync_in / sync_out payloads interchange."""

    _SEG = ("w1", "b1", "w2", "b2", "w3", "b3", "wl", "bl", "mu_av1", "sig_av1", "mub_av1", "sigb_av1", "mu_a2", "sig_a2", "mub_a2", "sigb_a2",
            "mu_v2", "sig_v2", "mub_v2", "sigb_v2")
    _KIND = {"rainbow": 0, "dueling": 1, "q": 2, "pv": 2}  # "pv": the discrete policy-value net (policy_value.py:8-22) = head -> l -> (pi | v) stacked into ONE last layer of A + 1 rows

    def __init__(self, state_size, action_size, num_support, hidden, head, max_batch, device, kind="rainbow", noise_type="factorized"):
        self.lib = L.load()
        self.device = torch.device(device)
        self.ctx = L.ctx(self.device.index)
        self.kind = kind
        self.cnn = head == "cnn"
        if self.cnn:
            self.Cin, self.Hin, self.Win = (int(v) for v in state_size)
        else:
            self.Cin, self.Hin, self.Win = int(state_size), 0, 0
        self.H, self.A, self.K, self.maxB = int(hidden), int(action_size), int(num_support), int(max_batch)
        if kind == "pv":
            self.n_actions, self.A = self.A, self.A + 1  # the library sees a q-network with one more output row: the value head
        kid = self._KIND[kind]
        self.noise_type = noise_type
        if kind == "rainbow" and noise_type == "independent":
            kid = 3
        n = int(self.lib.jh_rbnet_param_count_for(kid, int(self.cnn), self.Cin, self.Hin, self.Win, self.H, self.A, self.K))
        if n <= 0:
            L.check(-2)
        self.n_params = n
        mk = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.params, self.target, self.grads, self.m, self.v = mk(), mk(), mk(), mk(), mk()
        self.h = C.c_void_p()
        L.check(self.lib.jh_rbnet_create(self.ctx, kid, int(self.cnn), self.Cin, self.Hin, self.Win, self.H, self.A, self.K, self.maxB, L.ptr(self.params),
                                         L.ptr(self.target), L.ptr(self.grads), L.ptr(self.m), L.ptr(self.v), C.byref(self.h)))
        self.noise_len = int(self.lib.jh_rbnet_noise_len(self.h)) if kind == "rainbow" else 0
        self.seg = {}
        for i, name in enumerate(self._SEG):
            off, rows, cols = C.c_int64(), C.c_int32(), C.c_int32()
            L.check(self.lib.jh_rbnet_segment(self.h, i, C.byref(off), C.byref(rows), C.byref(cols)))
            self.seg[name] = (off.value, rows.value, cols.value)
        if self.cnn:
            d1 = ((self.Hin - 8) // 4 + 1, (self.Win - 8) // 4 + 1)
            d2 = ((d1[0] - 4) // 2 + 1, (d1[1] - 4) // 2 + 1)
            self.d3 = (d2[0] - 2, d2[1] - 2)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_rbnet_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- state_dict <-> private layout ---------------------------------------------------------
    def _v(self, bucket, name):
        off, rows, cols = self.seg[name]
        return bucket[off : off + rows * cols].view(rows, cols)

    def _feat_cols(self, w):
        """[R][F] weight that reads the head's features: the CNN head's features are (y, x, c) here and
        (c, y, x) in the reference -> a 4-D [R][c][y][x] view (export reshapes it, import views the source)."""
        return w.view(w.shape[0], self.d3[0], self.d3[1], 64).permute(0, 3, 1, 2) if self.cnn else w

    def _pairs(self, bucket):
        """[(reference key, view of the bucket shaped / strided like the reference tensor)] in the order of
        the reference module's state_dict (own parameters first, then sub-modules in registration order)."""
        H = self.H
        out, head = [], []
        if self.cnn:
            head.append(("head.conv1.weight", self._v(bucket, "w1").view(32, self.Cin, 8, 8)))
            head.append(("head.conv1.bias", self._v(bucket, "b1").view(-1)))
            head.append(("head.conv2.weight", self._v(bucket, "w2").view(64, 4, 4, 32).permute(0, 3, 1, 2)))
            head.append(("head.conv2.bias", self._v(bucket, "b2").view(-1)))
            head.append(("head.conv3.weight", self._v(bucket, "w3").view(64, 3, 3, 64).permute(0, 3, 1, 2)))
            head.append(("head.conv3.bias", self._v(bucket, "b3").view(-1)))
        else:
            head.append(("head.l.weight", self._v(bucket, "w1")))
            head.append(("head.l.bias", self._v(bucket, "b1").view(-1)))
        if self.kind == "rainbow":
            av1 = {k: self._v(bucket, f"{k}_av1") for k in ("mu", "sig")}
            bav1 = {k: self._v(bucket, f"{k}b_av1").view(-1) for k in ("mu", "sig")}
            for tag, sl in (("a1", slice(0, H)), ("v1", slice(H, 2 * H))):
                out += [(f"mu_w_{tag}", av1["mu"][sl].t()), (f"sig_w_{tag}", av1["sig"][sl].t()), (f"mu_b_{tag}", bav1["mu"][sl]), (f"sig_b_{tag}", bav1["sig"][sl])]
            for tag in ("a2", "v2"):
                out += [(f"mu_w_{tag}", self._v(bucket, f"mu_{tag}").t()), (f"sig_w_{tag}", self._v(bucket, f"sig_{tag}").t()),
                        (f"mu_b_{tag}", self._v(bucket, f"mub_{tag}").view(-1)), (f"sig_b_{tag}", self._v(bucket, f"sigb_{tag}").view(-1))]
            out += head
            out += [("l.weight", self._feat_cols(self._v(bucket, "wl"))), ("l.bias", self._v(bucket, "bl").view(-1))]
        elif self.kind == "dueling":
            av1, bav1 = self._v(bucket, "mu_av1"), self._v(bucket, "mub_av1").view(-1)
            out += head
            out += [("l1_a.weight", self._feat_cols(av1[:H])), ("l1_a.bias", bav1[:H]), ("l1_v.weight", self._feat_cols(av1[H:])), ("l1_v.bias", bav1[H:]),
                    ("l2_a.weight", self._v(bucket, "mu_a2")), ("l2_a.bias", self._v(bucket, "mub_a2").view(-1)),
                    ("l2_v.weight", self._v(bucket, "mu_v2")), ("l2_v.bias", self._v(bucket, "mub_v2").view(-1))]
        elif self.kind == "pv":
            w, b, n = self._v(bucket, "mu_a2"), self._v(bucket, "mub_a2").view(-1), self.n_actions
            out += head
            out += [("l.weight", self._feat_cols(self._v(bucket, "wl"))), ("l.bias", self._v(bucket, "bl").view(-1)),
                    ("pi.weight", w[:n]), ("pi.bias", b[:n]), ("v.weight", w[n:]), ("v.bias", b[n:])]
        else:
            out += head
            out += [("l.weight", self._feat_cols(self._v(bucket, "wl"))), ("l.bias", self._v(bucket, "bl").view(-1)),
                    ("q.weight", self._v(bucket, "mu_a2")), ("q.bias", self._v(bucket, "mub_a2").view(-1))]
        return out

    def export_state(self, bucket=None):
        from collections import OrderedDict

        bucket = self.params if bucket is None else bucket
        sd = OrderedDict()
        for k, v in self._pairs(bucket):
            sd[k] = (v.reshape(v.shape[0], -1) if (v.dim() == 4 and not k.startswith("head.")) else v).clone(memory_format=torch.contiguous_format)
        return sd

    @torch.no_grad()
    def import_state(self, sd, bucket=None):
        bucket = self.params if bucket is None else bucket
        pairs = dict(self._pairs(bucket))
        missing = [k for k in pairs if k not in sd]
        if missing:
            raise KeyError(f"state_dict is missing {missing}")
        for k, v in pairs.items():
            src = torch.as_tensor(sd[k]).to(self.device, torch.float32)
            if v.dim() == 4 and not k.startswith("head."):
                src = src.view(v.shape[0], 64, self.d3[0], self.d3[1])
            if tuple(src.shape) != tuple(v.shape):
                raise ValueError(f"{k}: expected {tuple(v.shape)}, got {tuple(src.shape)}")
            v.copy_(src)

    # ---- engine ---------------------------------------------------------------------------------
    def set_hyper(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, step=0, centered=False):
        """Adam: (lr, beta1, beta2, eps); RMSprop: (lr, alpha -> beta1, -, eps, centered)."""
        L.check(self.lib.jh_rbnet_set_hyper(self.h, float(lr), float(beta1), float(beta2), float(eps), int(step), int(bool(centered)), L.stream_ptr()))

    def set_lr(self, lr):
        L.check(self.lib.jh_rbnet_set_lr(self.h, float(lr), L.stream_ptr()))

    def sync_target(self):
        L.check(self.lib.jh_rbnet_sync_target(self.h, L.stream_ptr()))

    def _xdt(self, x):
        if x.dtype == torch.uint8:
            return L.JH_U8
        if x.dtype == torch.float32:
            return L.JH_F32
        raise TypeError(f"observations must be uint8 or float32 on the device, got {x.dtype}")

    def forward(self, x, which=0, noise=None, out=None):
        """network(x, is_train = noise is not None) -> logits [rows, A, K]; rows <= max_batch."""
        assert x.is_contiguous() and x.device == self.device
        rows = int(x.shape[0])
        if out is None:
            out = torch.empty(rows, self.A, self.K, dtype=torch.float32, device=self.device)
        L.check(self.lib.jh_rbnet_forward(self.h, int(which), L.ptr(x), self._xdt(x), rows, L.ptr(noise), L.ptr(out), L.stream_ptr()))
        return out

    def forward_keep(self, x, out):
        """network(x) of the online parameters with the activations kept for `backward` (an on-policy learner's forward: jh_rbnet_forward_keep)."""
        assert x.is_contiguous() and out.is_contiguous() and x.device == self.device
        L.check(self.lib.jh_rbnet_forward_keep(self.h, L.ptr(x), self._xdt(x), int(x.shape[0]), L.ptr(out), L.stream_ptr()))
        return out

    def learn_forward(self, x_all, B, noise, out):
        """x_all = [state; next_state] (2B rows), noise [3, noise_len] (rainbow; else None) -> out [3, B, A, K]."""
        assert x_all.is_contiguous() and (noise is None or noise.is_contiguous()) and out.is_contiguous() and int(x_all.shape[0]) == 2 * B
        L.check(self.lib.jh_rbnet_learn_forward(self.h, L.ptr(x_all), self._xdt(x_all), int(B), L.ptr(noise), L.ptr(out), L.stream_ptr()))
        return out

    def prepare_noise(self, noise):
        """The three noisy weight sets of learn()'s forwards for the draw `noise` [3, noise_len], on the CURRENT stream (may be a side stream)."""
        L.check(self.lib.jh_rbnet_prepare_noise(self.h, L.ptr(noise), L.stream_ptr()))

    def learn_trunk(self, x_all, B):
        assert x_all.is_contiguous() and int(x_all.shape[0]) == 2 * B
        L.check(self.lib.jh_rbnet_learn_trunk(self.h, L.ptr(x_all), self._xdt(x_all), int(B), L.stream_ptr()))

    def learn_heads(self, B, noise, out):
        assert (noise is None or noise.is_contiguous()) and out.is_contiguous()
        L.check(self.lib.jh_rbnet_learn_heads(self.h, int(B), L.ptr(noise), L.ptr(out), L.stream_ptr()))
        return out

    def learn_heads_raw(self, B, noise):
        """learn_heads up to the advantage / value streams; `c51_step` forms the logits inside the loss kernel."""
        assert noise is None or noise.is_contiguous()
        L.check(self.lib.jh_rbnet_learn_heads_raw(self.h, int(B), L.ptr(noise), L.stream_ptr()))

    def c51_step(self, tree, idx, action, reward, done, weights, v_min, v_max, gamma, alpha, n_step, logits, stats=None):
        """Rainbow.learn()'s loss step in three launches (jh_rbnet_c51_step): dueling combine of the three forwards (-> logits [3, B, A, K])
        + double-Q projection + KL + the gradient back through the combine (stays in the network: `backward(None)`); statistics + the
        priorities KL^alpha into the leaves `idx` (tree space, int64) of `tree` (ops.SumTree or None); the climb.  -> (prio [B], kl [B], stats f32[8])."""
        lib = self.lib
        B = int(logits.shape[1])
        assert logits.is_contiguous() and logits.dtype == torch.float32 and tuple(logits.shape) == (3, B, self.A, self.K)
        prio = torch.empty(B, dtype=torch.float32, device=self.device)
        kl = torch.empty(B, dtype=torch.float32, device=self.device)
        if stats is None:
            stats = torch.empty(8, dtype=torch.float32, device=self.device)
        r, d = _f32(reward).reshape(B, -1), _f32(done).reshape(B, -1)
        assert r.shape[1] == max(n_step, 1) and d.shape == r.shape
        flags = L.JH_C51_DOUBLE | (L.JH_C51_PER if weights is not None else 0)
        if tree is not None:
            assert idx.dtype == torch.int64 and idx.is_cuda and idx.is_contiguous() and idx.numel() == B
        L.check(lib.jh_rbnet_c51_step(self.h, None if tree is None else tree.h, B, int(n_step), flags, L.ptr(_f32(action).reshape(-1)), L.ptr(r), L.ptr(d),
                                      L.ptr(None if weights is None else _f32(weights).reshape(-1)), L.ptr(None if tree is None else idx), float(v_min), float(v_max),
                                      float(gamma), float(alpha), L.ptr(logits), L.ptr(prio), L.ptr(kl), L.ptr(stats), L.stream_ptr()))
        return prio, kl, stats

    def backward(self, g, defer=False):
        """g = d(loss)/d(logits of online(state)), or None after `c51_step`.  defer: leave d(sigma) = d(mu) * eps and the sum of conv1's
        weight-gradient partials to `optim_step` (it folds them into the optimizer pass when there is no clipping); `flush_grads`
        completes the bucket for a reader in between."""
        assert g is None or (g.is_contiguous() and g.dtype == torch.float32)
        fn = self.lib.jh_rbnet_backward_deferred if defer else self.lib.jh_rbnet_backward
        L.check(fn(self.h, L.ptr(g), L.stream_ptr()))

    def flush_grads(self):
        L.check(self.lib.jh_rbnet_flush_grads(self.h, L.stream_ptr()))

    def adam_step(self):
        L.check(self.lib.jh_rbnet_adam_step(self.h, L.stream_ptr()))

    def optim_step(self, optimizer="adam", max_norm=None):
        """[clip_grad_norm_(max_norm)] + optimizer.step(); optimizer in {"adam", "rmsprop"}."""
        L.check(self.lib.jh_rbnet_optim_step(self.h, {"adam": 0, "rmsprop": 1}[optimizer], float(max_norm or 0.0), L.stream_ptr()))


class StagingRing:
    """jh_ring_*: bounded lock-free multi-producer / single-consumer ring of transitions in pinned host memory
    (the async Ape-X transport).  `produce` may be called from any number of actor threads (the GIL is released
    inside the C call); `drain` belongs to the learner thread.

    columns: list of (name, jh_dtype, elems, shape) -- pass `store.columns` to stage for a DeviceStore.
    device=None builds a host-only ring (pageable memory; `consume_host` only)."""

    def __init__(self, slots, columns, with_priority=False, device="cuda"):
        self.lib = L.load()
        self.columns = list(columns)
        self.names = [c[0] for c in columns]
        self.slots, self.with_priority = int(slots), bool(with_priority)
        ctx = None
        if device is not None:
            self.device = torch.device(device)
            ctx = L.ctx(self.device.index if self.device.index is not None else torch.cuda.current_device())
        descs = (L.ColDesc * len(columns))(*[L.ColDesc(int(c[1]), int(c[2])) for c in columns])
        self.h = C.c_void_p()
        L.check(self.lib.jh_ring_create(ctx, self.slots, len(columns), descs, int(self.with_priority), C.byref(self.h)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jh_ring_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def produce(self, cols, priorities=None, timeout_ms=10000):
        """cols: dict name -> numpy array [n, ...] (converted to the stored dtype); thread-safe.  Blocks while the
        ring is full -- slots come back when the LEARNER thread drains / reclaims -- for at most timeout_ms
        (JhError, nothing written); timeout_ms < 0 waits forever."""
        arrs, n = [], None
        for name, dt, elems, _ in self.columns:
            src = np.asarray(cols[name]).reshape(len(cols[name]), -1)
            a = np.ascontiguousarray(src, dtype=_NP_OF[dt])
            if dt == L.JH_I64 and src.dtype.kind == "f" and not np.array_equal(a, src):
                raise ValueError(f"column {name} is stored as int64 (its first batch was integer-typed) but this batch holds fractional values")
            assert a.shape[1] == elems, f"column {name}: expected {elems} elems, got {a.shape[1]}"
            n = a.shape[0] if n is None else n
            assert a.shape[0] == n
            arrs.append(a)
        p = None
        if self.with_priority:
            p = np.ascontiguousarray(priorities, dtype=np.float64).reshape(-1)
            assert p.size == n
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        L.check(self.lib.jh_ring_produce(self.h, int(n), ptrs, None if p is None else p.ctypes.data, int(timeout_ms)))
        return n

    def drain(self, store, tree=None, max_rows=0):
        """Learner thread: everything published so far -> `store` (DeviceStore) [+ leaves into `tree` (SumTree)]."""
        n = C.c_int64()
        L.check(self.lib.jh_ring_drain(self.h, store.h, tree.h if tree is not None else None, int(max_rows), L.stream_ptr(), C.byref(n)))
        return int(n.value)

    def reclaim(self, wait=False):
        """Learner thread: give the slots of completed drains back to the producers (drain() does this too)."""
        L.check(self.lib.jh_ring_reclaim(self.h, int(bool(wait))))

    def consume_host(self, max_rows=0):
        """-> (dict name -> numpy [n, elems], priorities float64[n]) of the rows published so far."""
        cap = self.slots if max_rows <= 0 else int(max_rows)
        outs = [np.empty((cap, elems), dtype=_NP_OF[dt]) for _, dt, elems, _ in self.columns]
        prio = np.empty(cap, np.float64)
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        n = C.c_int64()
        L.check(self.lib.jh_ring_consume_host(self.h, cap, ptrs, prio.ctypes.data, C.byref(n)))
        k = int(n.value)
        return {name: o[:k] for name, o in zip(self.names, outs)}, prio[:k]

    def stats(self):
        a, b, w = C.c_int64(), C.c_int64(), C.c_double()
        L.check(self.lib.jh_ring_stats(self.h, C.byref(a), C.byref(b), C.byref(w)))
        return {"produced": a.value, "drained": b.value, "producer_wait_ms": w.value}


def tgemm_dense(a, b, a_kcont=True, b_kcont=True, epi=0, bias=None, aux=None, rowsum=False, M=None, N=None, K=None):
    """jh_tgemm_dense: C = A (.) B on the grouped LDS-tiled fp32 MFMA engine (dense operand modes).
    a: [M, K] if a_kcont else [K, M];  b: [N, K] if b_kcont else [K, N]  (row-major, any row stride)."""
    lib = L.load()
    M = (a.shape[0] if a_kcont else a.shape[1]) if M is None else M
    K = (a.shape[1] if a_kcont else a.shape[0]) if K is None else K
    N = (b.shape[0] if b_kcont else b.shape[1]) if N is None else N
    assert a.stride(-1) == 1 and b.stride(-1) == 1 and a.dtype == b.dtype == torch.float32
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    rs = torch.empty(M, dtype=torch.float32, device=a.device) if rowsum else None
    pa, pb = C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr())  # row-strided views are fine: the stride is passed as lda / ldb
    L.check(lib.jh_tgemm_dense(L.ctx(a.device.index), int(M), int(N), int(K), pa, int(a.stride(0)), int(bool(a_kcont)), pb, int(b.stride(0)), int(bool(b_kcont)),
                               L.ptr(c), int(c.stride(0)), int(epi), L.ptr(bias), L.ptr(aux), int(aux.stride(0)) if aux is not None else 0, L.ptr(rs), L.stream_ptr()))
    return (c, rs) if rowsum else c


def tgemm_set_cfg(cfg=""):
    """jh_tgemm_set_cfg: per-call-site tile / split / XCD-order overrides of the tile engine (measurement and tests); "" = defaults."""
    L.check(L.load().jh_tgemm_set_cfg((cfg or "").encode()))


def tgemm_dense_group(a_list, b_list):
    """jh_tgemm_dense_group: C_j = A_j B_j^T for up to six same-shape problems (A_j [M, K], B_j [N, K], contiguous) as ONE grouped
    launch -- the shape of the value networks' forward launches (online and target trunks side by side)."""
    lib = L.load()
    n = len(a_list)
    M, K = a_list[0].shape
    N = b_list[0].shape[0]
    for a, b in zip(a_list, b_list):
        assert a.shape == (M, K) and b.shape == (N, K) and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype == torch.float32
    cs = [torch.empty(M, N, dtype=torch.float32, device=a_list[0].device) for _ in range(n)]
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    L.check(lib.jh_tgemm_dense_group(L.ctx(a_list[0].device.index), n, int(M), int(N), int(K), arr(a_list), arr(b_list), arr(cs), L.stream_ptr()))
    return cs
