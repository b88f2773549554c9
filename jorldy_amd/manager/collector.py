"""Vectorised sync rollout collectors -- replace DistributedManager + the Ray `Actor`s
(manager/distributed_manager.py:7-95) for sync mode.

Reference: N Ray actors, each with one env and a CPU copy of the agent, run `step` B=1 forwards each;
the learner unpickles W*T dicts and re-broadcasts the full state_dict every iteration.
Here: ONE batched policy forward on the GPU per timestep for all W envs, the envs are stepped by
one native call on the host (jh_cartpole_step), transitions are written straight into SoA arrays in
the reference's worker-major order (w0 t0..tT-1, w1 ...), and `sync()` is a no-op because the
acting network IS the learner's network (weights never leave HBM; removes base.py:78-85 traffic).

VecCollector     Python loop over timesteps, works with any agent (`agent.act`).
NativeCollector  the whole T-step loop in one C call (jh_collector_run): acting kernels read the
                 observations from device-mapped pinned memory and write the actions back the same
                 way; transitions go directly into the rollout store's pinned staging slab.
"""
import ctypes as C

import numpy as np

from .. import _lib as L
from .. import ops


class VecCollector:
    def __init__(self, env_vec, agent, num_workers=None, mode="sync"):
        assert mode == "sync", "async (Ape-X) collection is a later row (SURVEY.md §8f)"
        self.env = env_vec
        self.agent = agent
        self.num_workers = env_vec.W if num_workers is None else num_workers
        assert self.num_workers == env_vec.W
        self.state = self.env.obs()  # (W, S) float32 -- or (W, C, H, W) frames in the env's own dtype (uint8 for the image envs: they stay uint8 up to the first convolution)
        from ..parallel import pin_to_gpu_node

        dev = getattr(agent, "device", None)
        self.host_cores = pin_to_gpu_node(dev.index) if dev is not None else None  # acting = PCIe round trips per step: stay next to the GPU

    def run(self, step=1):
        """-> (SoA dict in worker-major order, completed_ratio) like DistributedManager.run (:26-31)."""
        assert step > 0
        W, S, odt = self.state.shape[0], tuple(self.state.shape[1:]), self.state.dtype
        st = np.empty((W, step) + S, odt)
        ns = np.empty((W, step) + S, odt)
        rw = np.empty((W, step, 1), np.float32)
        dn = np.empty((W, step, 1), np.uint8)
        ac = None
        nxt, r, d = np.empty((W,) + S, odt), np.empty(W, np.float32), np.empty(W, np.uint8)
        for t in range(step):  # Actor.run, distributed_manager.py:76-92, for all workers at once
            action = self.agent.act(self.state, training=True)["action"]  # (W, 1) or (W, A)
            if ac is None:
                ac = np.empty((W, step) + action.shape[1:], action.dtype)
            self.env.step(action, nxt, r, d)
            st[:, t], ns[:, t], rw[:, t, 0], dn[:, t, 0], ac[:, t] = self.state, nxt, r, d, action
            self.env.obs(self.state)  # next_state, or the reset state where done (:91)
        flat = lambda a: a.reshape((W * step,) + a.shape[2:])
        return {"state": flat(st), "action": flat(ac), "reward": flat(rw), "next_state": flat(ns), "done": flat(dn)}, 1.0

    def sync(self, sync_item=None, init=False):
        """DistributedManager.sync (:55-60): nothing to do, actors read the learner's weights in HBM."""
        return None

    def terminate(self):
        return None


class JhEnvVtbl(C.Structure):
    """include/jorldy_hip.h: jh_env_vtbl."""
    OBS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float))
    STEP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8))
    FORK_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int32)
    FORK_FREE = C.CFUNCTYPE(None, C.c_void_p)
    COPY_ROW = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32)
    _fields_ = [("W", C.c_int32), ("S", C.c_int32), ("A", C.c_int32), ("continuous", C.c_int32), ("obs", OBS), ("step", STEP),
                ("fork_alloc", FORK_ALLOC), ("fork_free", FORK_FREE), ("copy_row", COPY_ROW)]


class PythonEnvTable:
    """A Python vector env behind the C collector's function table (jh_collector_create_env): the rollout loop, the persistent acting
    kernel and the staging stay native, only obs / step call back into Python -- any env with the VecCollector protocol

        env.W, env.state_size, env.action_size, env.action_type
        env.obs(out [W, S] float32)                       current observations (the reset observation where an episode just ended)
        env.step(action, next_obs, reward, done)          writes float32 [W, S], float32 [W], uint8 [W]; finished rows reset themselves

    and, optionally (both or neither): env.fork(rows) -> an env of `rows` rows with the same protocol whose obs / step also take a row
    range, and env.copy_row(di, src_env, si).  With them a two-action discrete env gets two timesteps per acting exchange, like the
    built-in CartPole.  The callbacks hold the GIL for their duration (jh_collector_run is entered through ctypes, which releases it)."""

    def __init__(self, env):
        self.env = env
        W, S, A = int(env.W), int(env.state_size), int(env.action_size)
        cont = env.action_type == "continuous"
        self._scratch = {}  # handle -> forked env (kept alive until fork_free)
        self._next_handle = 1
        # fork_alloc / fork_free / copy_row have no status in the C table: an exception inside them is recorded here and the NEXT obs / step
        # callback returns it, which stops the run (ctypes prints and swallows an exception that leaves a callback, and a
        # copy_row that failed silently left the env row stale -- rollouts that were no longer the env's)
        self._sticky = [0]
        sticky = self._sticky

        def rows_of(handle):
            return env if handle in (None, 0) or handle == 1 << 62 else self._scratch[handle]

        def obs_cb(handle, r0, r1, out):
            try:
                if sticky[0]:
                    rc, sticky[0] = sticky[0], 0
                    return rc
                e, n = rows_of(handle), r1 - r0
                buf = np.ctypeslib.as_array(out, shape=(n, S))
                if e is env:
                    e.obs(buf)
                else:
                    e.obs(buf, r0, r1)
                return 0
            except Exception:  # noqa: BLE001 -- an exception must not unwind through the C frames
                import traceback

                traceback.print_exc()
                return -2

        def step_cb(handle, r0, r1, action, nxt, rew, done):
            try:
                if sticky[0]:
                    rc, sticky[0] = sticky[0], 0
                    return rc
                e, n = rows_of(handle), r1 - r0
                a = np.ctypeslib.as_array(C.cast(action, C.POINTER(C.c_float if cont else C.c_int64)), shape=(n, A) if cont else (n,))
                args = (a, np.ctypeslib.as_array(nxt, shape=(n, S)), np.ctypeslib.as_array(rew, shape=(n,)), np.ctypeslib.as_array(done, shape=(n,)))
                if e is env:
                    e.step(*args)
                else:
                    e.step(*args, r0, r1)
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return -2

        forkable = hasattr(env, "fork") and hasattr(env, "copy_row")

        def failed():
            import traceback

            traceback.print_exc()
            sticky[0] = -2

        def fork_alloc_cb(_handle, rows):
            try:
                h = self._next_handle = self._next_handle + 1
                self._scratch[h] = env.fork(int(rows))
                return h
            except Exception:  # noqa: BLE001
                failed()
                return None  # null handle: the collector runs this env one timestep per exchange

        def fork_free_cb(handle):
            self._scratch.pop(handle, None)

        def copy_row_cb(dst, di, src, si):
            try:
                rows_of(dst).copy_row(int(di), rows_of(src), int(si))
            except Exception:  # noqa: BLE001
                failed()

        self._cbs = (JhEnvVtbl.OBS(obs_cb), JhEnvVtbl.STEP(step_cb),
                     JhEnvVtbl.FORK_ALLOC(fork_alloc_cb) if forkable else JhEnvVtbl.FORK_ALLOC(),
                     JhEnvVtbl.FORK_FREE(fork_free_cb) if forkable else JhEnvVtbl.FORK_FREE(),
                     JhEnvVtbl.COPY_ROW(copy_row_cb) if forkable else JhEnvVtbl.COPY_ROW())
        self.table = JhEnvVtbl(W, S, A, int(cont), *self._cbs)
        self.handle = C.c_void_p(1 << 62)  # the env itself (any non-null cookie: the callbacks close over the object)


class NativeCollector:
    """jh_collector_*: the whole T-step rollout loop in one C call for a native PPO agent on a vectorised env: the library's own
    ops.CartPoleVec (discrete policy) / ops.ControlVec (continuous policy, config.ppo.mujoco shapes), or ANY Python vector env with the
    VecCollector protocol, which is then put behind the collector's function table (PythonEnvTable, jh_collector_create_env).  `run(step)`
    appends W*step transitions to the agent's rollout store and returns (None, 1.0); pass None to `agent.process`."""

    def __init__(self, env_vec, agent, num_workers=None, mode="sync"):
        assert mode == "sync"
        assert getattr(agent, "_net", None) is not None, "needs the native PPO backend"
        assert agent.action_type == env_vec.action_type, "policy / env action types differ"
        self.lib = L.load()
        self.env, self.agent = env_vec, agent
        self.num_workers = env_vec.W
        from ..parallel import pin_to_gpu_node

        self.host_cores = pin_to_gpu_node(agent.device.index)  # the rollout loop runs on this thread: keep it next to the GPU
        self.h = None
        self._store_h = None
        self._net_h = None
        self._table = None  # PythonEnvTable of a non-native env (owns the ctypes callbacks: must outlive the C collector)
        import os

        # acting-time capture (jh_collector_set_capture): the raw heads and values the acting kernel computed anyway replace the two
        # no-grad passes at the start of PPO.learn (ppo.py:83-94); JH_COLLECT_CAPTURE=0 switches it off
        self.capture = os.environ.get("JH_COLLECT_CAPTURE", "1") == "1"
        self._cap_key = None
        self._rides, self._rides_prev = {}, {}

    def _bind(self, n_rows):
        mem, W = self.agent.memory, self.env.W
        S, cont = self.env.state_size, self.env.action_type == "continuous"
        action = np.zeros((n_rows, self.env.action_size), np.float32) if cont else np.zeros((n_rows, 1), np.int64)
        example = {"state": np.zeros((n_rows, S), np.float32), "action": action, "reward": np.zeros((n_rows, 1), np.float32),
                   "next_state": np.zeros((n_rows, S), np.float32), "done": np.zeros((n_rows, 1), np.uint8)}
        mem._ensure(example, n_rows)
        self.agent._grow_native(W)
        cap = self.agent._capture_targets(n_rows) if self.capture else None  # may grow the network: before its handle is read
        store, net = mem._store, self.agent._net
        cap_key = tuple(t.data_ptr() for t in cap if t is not None) if cap else None
        if self.h is not None and self._store_h == store.h.value and self._net_h == net.h.value:
            if cap_key != self._cap_key:
                self._set_capture(cap, n_rows)
            return
        if self.h is not None:
            self.lib.jh_collector_destroy(self.h)
        cols = (C.c_int32 * 5)(*[store.names.index(k) for k in ("state", "action", "reward", "next_state", "done")])
        h = C.c_void_p()
        if isinstance(self.env, (ops.CartPoleVec, ops.ControlVec)):  # the library's own envs (by type: a Python env may have an attribute `h` of its own)
            create = self.lib.jh_collector_create_control if cont else self.lib.jh_collector_create
            L.check(create(L.ctx(self.agent.device.index), net.h, self.env.h, store.h, cols, C.byref(h)))
        else:  # any other env: obs / step (/ fork) call back into Python
            if self._table is None:
                self._table = PythonEnvTable(self.env)
            L.check(self.lib.jh_collector_create_env(L.ctx(self.agent.device.index), net.h, C.byref(self._table.table), self._table.handle, store.h, cols, C.byref(h)))
        self.h, self._store_h, self._net_h = h, store.h.value, net.h.value
        self._set_capture(cap, n_rows)
        self._rides = {}
        self.agent._ride = self  # the agent may hand its small per-learn() uploads (index lists, learning rate) to the commit launch

    def _set_capture(self, cap, n_rows):
        if cap is None:
            L.check(self.lib.jh_collector_set_capture(self.h, None, None, None, None, 0))
            self._cap_key = None
            return
        h0, h1, v, nv = cap
        L.check(self.lib.jh_collector_set_capture(self.h, L.ptr(h0), L.ptr(h1), L.ptr(v), L.ptr(nv), int(n_rows)))
        self._cap_key = tuple(t.data_ptr() for t in cap if t is not None)

    def ride_along(self, slot, src_dev_ptr, dst_ptr, nbytes, keep=None):
        """jh_collector_set_ride_along: the commit launch of the NEXT run also copies nbytes from device-mapped pinned memory at
        src_dev_ptr to the device buffer at dst_ptr (read when that launch executes, i.e. at the end of the run).  One-shot: the
        launch consumes the registration.  `keep`: objects that own the two buffers -- held here until the slot is consumed or
        cleared, so that a reallocation on the agent's side cannot leave the registration dangling."""
        if self.h is None:
            return
        L.check(self.lib.jh_collector_set_ride_along(self.h, int(slot), C.c_void_p(int(src_dev_ptr)), C.c_void_p(int(dst_ptr)), int(nbytes)))
        self._rides[slot] = keep

    def clear_rides(self):
        """Drop whatever is registered (the agent reallocated its buffers, or set the learning rate directly)."""
        if self.h is not None:
            for slot in (0, 1):
                L.check(self.lib.jh_collector_set_ride_along(self.h, slot, None, None, 0))
        self._rides = {}

    def run(self, step=1):
        n_rows = self.env.W * step
        self._bind(n_rows)
        L.check(self.lib.jh_collector_run(self.h, int(step), 1, L.stream_ptr()))
        self._rides_consumed()
        self.agent._ride_done = True  # this run's commit launch carried whatever was registered with ride_along
        if self._cap_key is not None:
            self.agent._captured = n_rows  # the coming learn() takes the heads / values of these rows as delivered (no no-grad passes)
        return None, 1.0

    def _rides_consumed(self):
        # the commit launch is enqueued: the buffers must live until it has executed (stream order) -- one more generation
        self._rides_prev, self._rides = self._rides, {}

    def begin(self, step=1):
        """First half of run(step) with the commit launch enqueued AHEAD of the host loop (jh_collector_begin): after this call the
        rollout's rows count as stored, and the learner may enqueue its launches (agent.process_begin) -- they wait on the stream behind
        the acting kernel and the gated commit.  Then loop(), then agent.process_end():

            collector.begin(T); agent.process_begin(step); collector.loop(); result = agent.process_end()

        A stalled env (no observations for ~0.2 s) raises here instead of falling back to per-step launches (see the C header)."""
        n_rows = self.env.W * step
        self._bind(n_rows)
        L.check(self.lib.jh_collector_begin(self.h, int(step), L.stream_ptr()))
        self._rides_consumed()
        if self._cap_key is not None:
            self.agent._captured = n_rows
        self.agent._ride_done = True
        self._stream = L.stream_ptr()

    def loop(self):
        L.check(self.lib.jh_collector_loop(self.h, 1, self._stream))
        return None, 1.0

    def arm_prelaunch(self, step):
        """Have the coming `agent.process` enqueue the persistent acting kernel of the NEXT run(step) right behind learn()'s launches
        (jh_collector_prelaunch): the kernel's launch and start-up leave the host's critical path between learn() and the rollout.
        The kernel waits ~0.2 s for its first observations; nothing else may be enqueued on the stream before that run (a
        torch.cuda.synchronize() in between waits for the timeout, after which run() simply launches afresh)."""
        if self.h is None:
            return
        stream = L.stream_ptr()
        self.agent._post_launch_hook = lambda: L.check(self.lib.jh_collector_prelaunch(self.h, int(step), stream))

    def stats(self, reset=True):
        a, e = C.c_double(), C.c_double()
        d = (C.c_double * 6)()
        L.check(self.lib.jh_collector_stats_detail(self.h, d))
        L.check(self.lib.jh_collector_stats(self.h, C.byref(a), C.byref(e), int(reset)))
        return {"act_us_per_step": a.value, "env_us_per_step": e.value, "first_step_us_per_run": d[0], "value_query_us_per_run": d[1],
                "commit_us_per_run": d[2], "act_steady_us_per_step": d[3]}

    def sync(self, sync_item=None, init=False):
        return None

    def terminate(self):
        if self.h is not None:
            if getattr(self.agent, "_ride", None) is self:
                self.agent._ride = None
            self.lib.jh_collector_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.terminate()
        except Exception:
            pass
