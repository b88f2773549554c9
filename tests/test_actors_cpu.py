"""Host-side logic of the batched actors (no GPU): the vectorised Ape-X n-step assembler against the per-actor
oracle (oracle.NStepOracle = ape_x.py:174-199, itself pinned on the reference's fixture)."""
import numpy as np

from oracle.jorldy_oracle import NStepOracle


def test_vec_nstep_apex_equals_per_actor_assemblers():
    import importlib.util
    import os

    # the module only needs numpy for this class; import it without the package's GPU-side imports
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jorldy_amd", "manager", "batched_actors.py")
    src = open(path).read().replace("from .. import ops", "ops = None")
    mod = {}
    exec(compile(src, path, "exec"), mod)
    VecNStepApeX = mod["VecNStepApeX"]

    N, n, gamma, S = 5, 3, 0.99, (2, 3)
    rng = np.random.RandomState(0)
    vec = VecNStepApeX(N, n, gamma, S, np.uint8)
    refs = [NStepOracle(n, apex=True, gamma=gamma) for _ in range(N)]
    for t in range(12):
        state = rng.randint(0, 256, size=(N,) + S).astype(np.uint8)
        action = rng.randint(0, 4, size=(N, 1))
        reward = rng.randn(N, 1).astype(np.float32)
        done = (rng.rand(N, 1) < 0.3).astype(np.float32)
        q = rng.randn(N, 1).astype(np.float32)
        out = vec.push(state, action, reward, done, q)
        exp = [refs[i].push({"state": state[i : i + 1], "action": action[i : i + 1], "reward": reward[i : i + 1].astype(np.float64), "next_state": state[i : i + 1],
                             "done": done[i : i + 1].astype(bool), "q": q[i : i + 1]}) for i in range(N)]
        if t < n:
            assert out is None and not any(exp)
            continue
        cols, prio = out
        for i in range(N):
            e = exp[i]
            np.testing.assert_array_equal(cols["state"][i : i + 1], e["state"])
            np.testing.assert_array_equal(cols["next_state"][i : i + 1], e["next_state"])
            np.testing.assert_array_equal(cols["action"][i : i + 1], e["action"])
            np.testing.assert_allclose(cols["reward"][i : i + 1], e["reward"], rtol=1e-6)
            np.testing.assert_array_equal(cols["done"][i : i + 1].astype(bool), e["done"])
            np.testing.assert_allclose(prio[i], np.asarray(e["priority"]).reshape(-1)[0], rtol=1e-5, atol=1e-6)
