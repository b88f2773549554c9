"""Vectorised sync rollout collector -- replaces DistributedManager + the Ray `Actor`s
(manager/distributed_manager.py:7-95) for sync mode.

Reference: N Ray actors, each with one env and a CPU copy of the agent, run `step` B=1 forwards each;
the learner unpickles W*T dicts and re-broadcasts the full state_dict every iteration.
Here: ONE batched policy forward on the GPU per timestep for all W envs, the envs are stepped by
one native call on the host (jh_cartpole_step), transitions are written straight into SoA arrays in
the reference's worker-major order (w0 t0..tT-1, w1 ...), and `sync()` is a no-op because the
acting network IS the learner's network (weights never leave HBM; removes base.py:78-85 traffic).
"""
import numpy as np


class VecCollector:
    def __init__(self, env_vec, agent, num_workers=None, mode="sync"):
        assert mode == "sync", "async (Ape-X) collection is a later row (SURVEY.md §8f)"
        self.env = env_vec
        self.agent = agent
        self.num_workers = env_vec.W if num_workers is None else num_workers
        assert self.num_workers == env_vec.W
        self.state = self.env.obs()  # (W, S) float32

    def run(self, step=1):
        """-> (SoA dict in worker-major order, completed_ratio) like DistributedManager.run (:26-31)."""
        assert step > 0
        W, S = self.state.shape
        st = np.empty((W, step, S), np.float32)
        ns = np.empty((W, step, S), np.float32)
        rw = np.empty((W, step, 1), np.float32)
        dn = np.empty((W, step, 1), np.uint8)
        ac = None
        nxt, r, d = np.empty((W, S), np.float32), np.empty(W, np.float32), np.empty(W, np.uint8)
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
