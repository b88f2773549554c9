"""Batched on-GPU acting for the value-net agents (configs[3]: Ape-X's many actors).

Reference: every actor is a process with a CPU copy of the agent; each env step is one B = 1 forward
(manager/distributed_manager.py:76-92 -> core/agent/ape_x.py:64-77 / rainbow.py:140-152 / dqn.py:76-92), and the
learner ships its whole state_dict to all of them every `update_period` (DistributedManager.sync, :55-60).

Here: the N actors' observations of one tick (uint8 frames, N x 28 KB at Atari shapes) are staged in ONE pinned slab,
copied with one async H2D, pushed through ONE forward of an acting copy of the native network (jh_rbnet_forward:
implicit-GEMM convolutions reading the uint8 frames, same kernels as the learner), and turned into actions by one
kernel (jh_value_act: Q from the outputs, per-actor epsilon-greedy with the host's random draws, first-maximum
argmax, Q of the action taken for Ape-X's actor-side priorities).  The acting copy has its own activation
workspaces and runs on its own HIP stream, so actor ticks overlap with the learner's graph replays; `sync()` is one
device-to-device copy of the flat parameter bucket (the reference's state_dict broadcast).
"""
import numpy as np
import torch

from .. import ops


class BatchedValueActors:
    """act() for n_actors environments at once on an acting copy of `agent`'s native value network.

    agent: a DQN-family / Rainbow / Ape-X agent with backend="native".  epsilons: per-actor exploration rates
    (default: Ape-X's schedule eps^(1 + i/(N-1) * alpha), ape_x.py:166-172, when the agent has one; else agent.epsilon
    for all).  Rainbow acts through its noisy layers (a fresh noise draw per tick, rainbow.py:150) and epsilon = 0."""

    def __init__(self, agent, n_actors, epsilons=None, stream=None):
        src = getattr(agent, "_net", None)
        assert src is not None and hasattr(src, "target"), "needs a native value-net agent (backend='native')"
        self.agent, self.N, self.device = agent, int(n_actors), agent.device
        state_size = (src.Cin, src.Hin, src.Win) if src.cnn else src.Cin
        self.net = ops.RainbowNet(state_size, src.A, src.K, src.H, "cnn" if src.cnn else "mlp", self.N, self.device, kind=src.kind, noise_type=src.noise_type)
        self.stream = stream or torch.cuda.Stream(device=self.device)
        self.noisy = src.kind == "rainbow"
        if epsilons is None:
            if self.noisy:
                epsilons = np.zeros(self.N)
            elif hasattr(agent, "epsilon_alpha") and self.N > 1:
                epsilons = np.asarray([agent.epsilon ** (1 + (i / (self.N - 1)) * agent.epsilon_alpha) for i in range(self.N)])
            else:
                epsilons = np.full(self.N, float(getattr(agent, "epsilon", 0.0)))
        self.eps = np.ascontiguousarray(epsilons, dtype=np.float32)
        assert self.eps.size == self.N
        self.v_min, self.v_max = float(getattr(agent, "v_min", 0.0)), float(getattr(agent, "v_max", 0.0))
        shape = (self.N,) + ((src.Cin, src.Hin, src.Win) if src.cnn else (src.Cin,))
        self.x_dtype = torch.uint8 if src.cnn else torch.float32
        self._x_pin = torch.empty(shape, dtype=self.x_dtype, pin_memory=True)  # the tick's observations, staged once
        self._x_dev = torch.empty(shape, dtype=self.x_dtype, device=self.device)
        self._logits = torch.empty(self.N, src.A, src.K, dtype=torch.float32, device=self.device)
        import os

        self._mapped = os.environ.get("JH_MAPPED_STATS", "1") == "1"
        if self._mapped:
            # actions / q of a tick in device-MAPPED pinned memory: the act kernel writes them across PCIe itself and act()
            # waits for their arrival (jh_host_wait_words, no GIL) instead of two D2H copies + an event synchronise
            # Two sets, alternating by tick: the consumers of a tick's outputs (DeviceActorFeed's emit kernel) are enqueued,
            # not finished, when the next act() starts; they are finished once the NEXT tick's actions have arrived (same
            # stream), which is before the set is marked and written again two ticks later.
            self._maps = []
            for _ in range(2):
                am, qm = ops.PinnedBuffer((self.N,), np.int64, self.device.index), ops.PinnedBuffer((self.N,), np.float32, self.device.index)
                am.np[:] = 0
                self._maps.append((am, qm, ops._wrap_device(am.dev_ptr.value, (self.N,), torch.int64, self.device, owner=am),
                                   ops._wrap_device(qm.dev_ptr.value, (self.N,), torch.float32, self.device, owner=qm), am.np.view(np.uint32)))
            self._act_dev, self._q_dev = self._maps[0][2], self._maps[0][3]
            self._act_marks = np.arange(0, 2 * self.N, 2, dtype=np.int32)  # low words of the int64 actions
        else:
            self._act_dev = torch.empty(self.N, dtype=torch.int64, device=self.device)
            self._q_dev = torch.empty(self.N, dtype=torch.float32, device=self.device)
            self._act_pin = torch.empty(self.N, dtype=torch.int64, pin_memory=True)
            self._q_pin = torch.empty(self.N, dtype=torch.float32, pin_memory=True)
        self._noise = torch.empty(max(1, self.net.noise_len), dtype=torch.float32, device=self.device) if self.noisy else None
        self._normal = ops.NormalSource(self.device) if self.noisy else None
        self._done = torch.cuda.Event()
        self.ticks = 0
        self.sync()

    @property
    def obs_slab(self):
        """numpy view [N, *obs_shape] of the pinned staging slab: vectorised envs can write their observations in place."""
        return self._x_pin.numpy()

    def sync(self):
        """DistributedManager.sync + BaseAgent.sync_in (base.py:75-85) for all actors: one D2D copy of the learner's
        flat parameter bucket on the acting stream (ordered after the learner's last enqueued update)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.net.params.copy_(self.agent._net.params, non_blocking=True)

    @torch.no_grad()
    def act(self, obs=None, training=True, random_actions=False, upload=True):
        """obs: numpy [N, *obs_shape] (uint8 frames / float32 vectors) or None when the envs wrote into `obs_slab`.
        -> {"action": int64 [N, 1], "q": float32 [N, 1]} (blocking: the envs need the actions).
        random_actions: the warm-up branch of Rainbow.act / the agents' `memory.size < start_train_step` phase.
        upload=False: the observations are already in `self._x_dev` (a device-side producer on this stream)."""
        if obs is not None:
            np.copyto(self._x_pin.numpy(), obs, casting="same_kind")
        N = self.N
        greedy = not training
        eps = u = ra = None
        if not greedy and (random_actions or np.any(self.eps > 0)):
            eps = np.ones(N, np.float32) if random_actions else self.eps
            u = np.random.random(N)                                  # `np.random.random() < epsilon` per actor
            ra = np.random.randint(0, self.net.A, size=N)            # `np.random.randint(0, action_size)` per actor
        with torch.cuda.stream(self.stream):
            if upload:
                self._x_dev.copy_(self._x_pin, non_blocking=True)
            noise = None
            if self.noisy and training:
                noise = self._normal.fill(self._noise)
            self.net.forward(self._x_dev, which=0, noise=noise, out=self._logits)
            if self._mapped:
                am, qm, self._act_dev, self._q_dev, words = self._maps[self.ticks & 1]
                am.np[:] = -1  # arrival marks (actions are >= 0)
            ops.value_act(self._logits, self.v_min, self.v_max, eps, u, ra, out=(self._act_dev, self._q_dev))
            if not self._mapped:
                self._act_pin.copy_(self._act_dev, non_blocking=True)
                self._q_pin.copy_(self._q_dev, non_blocking=True)
                self._done.record(self.stream)
        self.ticks += 1
        if self._mapped:
            from .. import _lib as L

            if L.load().jh_host_wait_words(L.ptr(words), L.ptr(self._act_marks), N, 0xFFFFFFFF, 5.0) != 0:
                self.stream.synchronize()
                if (am.np < 0).any():
                    raise RuntimeError("BatchedValueActors.act(): the actions never arrived (failed launch?)")
            return {"action": am.np.reshape(N, 1).copy(), "q": qm.np.reshape(N, 1).copy()}
        self._done.synchronize()
        return {"action": self._act_pin.numpy().reshape(N, 1).copy(), "q": self._q_pin.numpy().reshape(N, 1).copy()}


class VecNStepApeX:
    """Ape-X's n-step assembler with actor-side priorities (ape_x.py:174-199) for N actors at once: the per-actor
    deque(maxlen = n + 1) of the reference becomes rolling arrays [n + 1, N, ...]; one push per tick emits N n-step
    transitions {state_t, action_t, reward[n], done[n], next_state = state_{t+n}} and priority = |G_n - q_t| where
    G_n folds r_i + (1 - d_i) gamma ... onto q_{t+n}.  Windows straddle episode ends like the reference's."""

    def __init__(self, n_actors, n_step, gamma, obs_shape, obs_dtype=np.uint8):
        self.N, self.n, self.gamma = n_actors, n_step, gamma
        L = n_step + 1
        self.state = np.zeros((L, n_actors) + tuple(obs_shape), dtype=obs_dtype)
        self.action = np.zeros((L, n_actors, 1), np.int64)
        self.reward = np.zeros((L, n_actors, 1), np.float32)
        self.done = np.zeros((L, n_actors, 1), np.float32)
        self.q = np.zeros((L, n_actors, 1), np.float32)
        self.count = 0

    def push(self, state, action, reward, done, q):
        """-> None until n + 1 ticks have been seen, then (cols dict of N rows, priorities [N])."""
        L = self.n + 1
        i = self.count % L
        self.state[i], self.action[i], self.reward[i], self.done[i], self.q[i] = state, action, reward, done, q
        self.count += 1
        if self.count < L:
            return None
        order = [(i + 1 + k) % L for k in range(L)]  # oldest .. newest
        o0, oN = order[0], order[-1]
        tq = self.q[oN].astype(np.float32)
        for k in reversed(order[:-1]):
            tq = self.reward[k] + (1.0 - self.done[k]) * self.gamma * tq
        cols = {"state": self.state[o0], "action": self.action[o0],
                "reward": np.stack([self.reward[k] for k in order[:-1]], axis=1), "next_state": self.state[oN],
                "done": np.stack([self.done[k] for k in order[:-1]], axis=1).astype(np.uint8)}
        return cols, np.abs(tq - self.q[o0]).reshape(-1).astype(np.float64)


class DeviceActorFeed:
    """BatchedValueActors -> replay, without a second trip over PCIe (SURVEY.md §8f ranks 1-3 in one path).

    The host pipeline (VecNStepApeX + StagingRing) copies every transition's two frame stacks three times on the host
    and uploads them again although the same pixels were uploaded a moment ago for the batched forward.  Here the
    tick's stacks stay where the forward read them; jh_feed_tick (csrc/jh_feed.hip) de-duplicates them into the
    buffer's plane pool and assembles Ape-X's n-step transitions with actor-side priorities (ape_x.py:174-199) on the
    acting stream; the learner's `memory.drain()` appends the emitted rows device-to-device and pushes their leaves.
    Only the env's rewards and done flags (2 N floats per tick) go up a second time.

        feed = DeviceActorFeed(actors, agent.memory, n_step, gamma)
        actor thread:    out = feed.act()            # envs wrote into feed.obs_slab
                         ... env.step(out["action"]) ...
                         feed.push(reward, done)     # blocks while `depth` emissions wait for the learner
        learner thread:  agent.process(None, step)   # -> memory.drain() -> feed.drain_into(memory)

    Emissions are never dropped: the plane rings are sized for rows that are overwritten after buffer_size / N ticks."""

    def __init__(self, actors, memory, n_step, gamma, depth=32, prio_eps=0.0, pool_factor=1.5):
        import collections
        import threading

        assert actors.x_dtype == torch.uint8 and actors._x_dev.dim() == 4, "frame-stack observations [N, C, H, W] uint8"
        self.actors, self.memory, self.N, self.n, self.depth, self.prio_eps = actors, memory, actors.N, int(n_step), int(depth), float(prio_eps)
        C = actors._x_dev.shape[1]
        self.pool = memory.attach_actor_feed(self.N, tuple(actors._x_dev.shape[1:]), n_step, gamma, pool_factor, in_flight_ticks=depth + 2)
        memory._feeds.append(self)
        dev = actors.device
        self._obs = [actors._x_dev, torch.empty_like(actors._x_dev)]  # this tick's / the previous tick's stacks
        mk = lambda shape, dt: torch.empty((self.depth + 1,) + shape, dtype=dt, device=dev)
        self._out = {"state": mk((self.N, C), torch.int64), "next_state": mk((self.N, C), torch.int64), "action": mk((self.N, 1), torch.int64),
                     "reward": mk((self.N, self.n, 1), torch.float32), "done": mk((self.N, self.n, 1), torch.uint8), "priority": mk((self.N,), torch.float64)}
        self._ready = [torch.cuda.Event() for _ in range(self.depth)]     # emission e written (acting stream)
        self._taken = [torch.cuda.Event() for _ in range(self.depth)]     # emission e stored (learner stream)
        self._taken_valid = [False] * self.depth
        self._free = threading.Semaphore(self.depth)
        self._queue = collections.deque()
        self._closed = False
        self.ticks = self.emissions = self.stored_rows = 0
        self.wait_s = 0.0
        self._mode, self._frames_pin, self._frames_dev = None, None, None

    @property
    def obs_slab(self):
        return self.actors.obs_slab

    def act(self, obs=None, training=True, random_actions=False):
        """Stack mode: the envs hand over full stacks [N, C, H, W] (or wrote them into `obs_slab`)."""
        assert self._mode in (None, "stacks"), "this feed is in frame mode"
        self._mode = "stacks"
        a = self.actors
        a._x_dev = self._obs[self.ticks & 1]
        return a.act(obs, training=training, random_actions=random_actions)

    @property
    def frame_slab(self):
        """Frame mode: numpy view [N, H, W] of a pinned slab for the envs' NEWEST frames (written in place)."""
        if self._frames_pin is None:
            shape = (self.N,) + tuple(self._obs[0].shape[2:])
            self._frames_pin = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
            self._frames_dev = torch.empty(shape, dtype=torch.uint8, device=self.actors.device)
        return self._frames_pin.numpy()

    def act_frames(self, frames=None, reset=None, training=True, random_actions=False):
        """Frame mode: the envs hand over only their newest frame [N, H, W] (or wrote it into `frame_slab`) and a reset
        flag per actor (True: the env was reset, its stack is that frame C times -- core/env/atari.py:112; otherwise the
        stack slides by one frame, :147).  The stacks for the forward are rebuilt in HBM from the plane pool: 7 KB instead
        of 28 KB per env step over PCIe at Atari shapes, and no frame-stack bookkeeping on the host at all.  A feed uses
        either act() or act_frames() for its whole life."""
        a = self.actors
        slab = self.frame_slab
        if frames is not None:
            np.copyto(slab, frames, casting="same_kind")
        if reset is None:
            reset = np.zeros(self.N, np.uint8)
        assert self._mode in (None, "frames"), "this feed is in stack mode"
        self._mode = "frames"
        a._x_dev = self._obs[self.ticks & 1]
        with torch.cuda.stream(a.stream):
            self._frames_dev.copy_(self._frames_pin, non_blocking=True)
            self.pool.feed.push_frames(self._frames_dev, reset, self.pool.planes, a._x_dev)
        return a.act(None, training=training, random_actions=random_actions, upload=False)

    def close(self):
        """Unblock a producer waiting in push() (end of run)."""
        self._closed = True
        self._free.release()

    def push(self, reward, done, timeout_s=None):
        """The env's answer to the actions of the last act(): enqueue this tick's de-duplication + n-step emission.
        Returns the number of transitions emitted (0 during the first n ticks), or -1 once close() was called."""
        import time

        a = self.actors
        e = self.emissions % self.depth
        will_emit = self.ticks >= self.n
        if will_emit:
            t0 = time.perf_counter()
            while not self._free.acquire(timeout=0.05):
                if self._closed or (timeout_s is not None and time.perf_counter() - t0 > timeout_s):
                    return -1
            self.wait_s += time.perf_counter() - t0
            if self._closed:
                return -1
        slot = e if will_emit else self.depth  # warm-up ticks write their (unused) outputs to the spare slot
        out = {k: v[slot] for k, v in self._out.items()}
        cur, prev = self._obs[self.ticks & 1], (self._obs[(self.ticks + 1) & 1] if self.ticks > 0 else None)
        got = 0
        try:
            with torch.cuda.stream(a.stream):
                if will_emit and self._taken_valid[e]:
                    a.stream.wait_event(self._taken[e])  # the learner's copies out of this slot are done
                flat = dict(out)
                flat["action"], flat["reward"], flat["done"] = out["action"].view(-1), out["reward"].view(self.N, self.n), out["done"].view(self.N, self.n)
                if self._mode != "frames":
                    self.pool.feed.push_stacks(cur, prev, self.pool.planes)
                got = self.pool.feed.emit(a._act_dev, a._q_dev, reward, done, flat, self.prio_eps)
                if got:
                    self._ready[e].record(a.stream)
        finally:
            if will_emit and not got:  # the slot was not filled (an error above, or the library emitted nothing): hand the permit back
                self._free.release()
        self.ticks += 1
        if got:
            self.emissions += 1
            self._queue.append(e)
        return got

    # ---- resume: what this object knows beyond the pool's device state (tick parity of the stack buffers, mode, last stacks) ----
    def save_stream(self, dirpath, k=0):
        """Call with the actors paused and everything drained (ReplayBuffer.save_stream does the drain)."""
        import os

        assert not self._queue, "emissions are waiting: drain first"
        torch.cuda.synchronize(self.actors.device)
        fn = None
        if self.ticks > 0:
            fn = f"feed_{k}_last_stacks.bin"
            self._obs[(self.ticks - 1) & 1].cpu().numpy().tofile(os.path.join(dirpath, fn))
        return {"ticks": int(self.ticks), "mode": self._mode, "last_stacks": fn, "stored_rows": int(self.stored_rows), "N": self.N, "n_step": self.n, "depth": self.depth}

    def load_stream(self, dirpath, st):
        import os

        assert (st["N"], st["n_step"]) == (self.N, self.n), "another feed geometry"
        assert not self._queue and self.ticks == 0, "restore into a fresh feed"
        self.ticks, self._mode, self.stored_rows, self.emissions = int(st["ticks"]), st["mode"], int(st["stored_rows"]), 0
        if st["last_stacks"]:
            a = np.fromfile(os.path.join(dirpath, st["last_stacks"]), dtype=np.uint8).reshape(tuple(self._obs[0].shape))
            self._obs[(self.ticks - 1) & 1].copy_(torch.from_numpy(a))
        self.actors._x_dev = self._obs[(self.ticks - 1) & 1] if self.ticks > 0 else self._obs[0]
        torch.cuda.synchronize(self.actors.device)

    def drain_into(self, memory):
        """Learner thread, learner stream: append every finished emission to the store (device to device) and push its
        leaves; returns the number of rows taken.  Emission slots are consecutive in memory, so everything that has
        arrived goes in as one append + one tree push per contiguous run of slots (two at the slot ring's wrap)."""
        cur = torch.cuda.current_stream(self.actors.device)
        self._drains = getattr(self, "_drains", 0) + 1
        if self._drains % 256 == 0:  # a plane-ring overrun only raises a device flag: look at it now and then (blocking, ~20 us), not only in stats()
            self.pool.check()
        slots = []
        while self._queue:
            slots.append(self._queue.popleft())
        n, i = 0, 0
        while i < len(slots):
            j = i
            while j + 1 < len(slots) and slots[j + 1] == slots[j] + 1:
                j += 1
            e0, e1, k = slots[i], slots[j], j - i + 1
            cur.wait_event(self._ready[e1])  # recorded in stream order: the earlier ones of the run are complete too
            cols = {key: self._out[key][e0 : e1 + 1].flatten(0, 1) for key in ("state", "action", "reward", "next_state", "done")}
            memory.store_feed_rows(cols, k * self.N, self._out["priority"][e0 : e1 + 1].flatten(0, 1))
            for e in range(e0, e1 + 1):
                self._taken[e].record(cur)
                self._taken_valid[e] = True
                self._free.release()
            n += k * self.N
            i = j + 1
        self.stored_rows += n
        return n

    def stats(self):
        s = self.pool.stats()
        s.update({"ticks": self.ticks, "emissions": self.emissions, "stored_rows": self.stored_rows, "producer_wait_ms": self.wait_s * 1e3})
        return s
