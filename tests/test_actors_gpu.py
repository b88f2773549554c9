"""Batched on-GPU acting of the value-net agents: jh_value_act against torch, and
BatchedValueActors (one forward per tick for all actors on an acting copy of the native network) against the agents'
own act() (ape_x.py:64-77, rainbow.py:140-152, dqn.py:76-92) row by row."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,A,K", [(64, 6, 1), (7, 4, 51), (130, 3, 21)])
def test_value_act_matches_torch(N, A, K):
    from jorldy_amd import ops

    torch.manual_seed(N)
    logits = torch.randn(N, A, K, device="cuda")
    v_min, v_max = -1.0, 10.0
    if K == 1:
        q_ref = logits[:, :, 0]
    else:
        z = torch.linspace(v_min, v_max, K, device="cuda")
        q_ref = (torch.exp(torch.log_softmax(logits, -1)) * z).sum(-1)  # rainbow.py:285-292
    act, q, q_all = ops.value_act(logits, v_min, v_max, want_q_all=True)
    torch.testing.assert_close(q_all, q_ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(act, q_ref.argmax(-1))
    torch.testing.assert_close(q, q_ref.max(-1).values, rtol=1e-5, atol=1e-5)
    # epsilon-greedy with the host's draws: rows with u < eps take the host's random action, and report ITS q
    rng = np.random.RandomState(0)
    eps, u, ra = rng.rand(N).astype(np.float32), rng.rand(N), rng.randint(0, A, size=N)
    act2, q2, _ = ops.value_act(logits, v_min, v_max, eps, u, ra)
    explore = u < eps.astype(np.float64)
    exp_act = np.where(explore, ra, q_ref.argmax(-1).cpu().numpy())
    assert np.array_equal(act2.cpu().numpy(), exp_act)
    torch.testing.assert_close(q2, q_ref[torch.arange(N), torch.from_numpy(exp_act).cuda()], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name,extra,S", [
    ("ape_x", dict(network="dueling", head="cnn", n_step=3, num_workers=8), (4, 44, 52)),
    ("rainbow", dict(head="cnn", n_step=3, num_support=21, v_min=-1, v_max=10), (4, 44, 52)),
    ("dqn", dict(), 6),
])
def test_batched_actors_equal_the_agents_own_act(name, extra, S):
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import BatchedValueActors

    torch.manual_seed(1)
    np.random.seed(1)
    A, N = 5, 9
    agent = Agent(name, state_size=S, action_size=A, hidden_size=32, batch_size=8, buffer_size=64, start_train_step=0, device="cuda", **extra)
    assert agent.backend == "native"
    with torch.no_grad():  # non-trivial weights (the policy heads start near zero)
        agent._net.params.add_(0.05 * torch.randn_like(agent._net.params))
    actors = BatchedValueActors(agent, N)
    if name == "ape_x":  # per-actor exploration schedule of ape_x.py:166-172
        np.testing.assert_allclose(actors.eps, [agent.epsilon ** (1 + i / (N - 1) * agent.epsilon_alpha) for i in range(N)], rtol=1e-6)
    rng = np.random.RandomState(2)
    obs = rng.randint(0, 256, size=(N,) + S).astype(np.uint8) if isinstance(S, tuple) else rng.randn(N, S).astype(np.float32)
    out = actors.act(obs, training=False)  # greedy, no noise: must equal the agent's own evaluation-mode act() row by row
    assert out["action"].shape == (N, 1) and out["action"].dtype == np.int64
    agent.epsilon_eval = 0.0
    for i in range(N):
        if name == "rainbow":
            agent.memory.buffer_counter = 10 ** 6  # past the warm-up branch of Rainbow.act
        ref = agent.act(obs[i : i + 1], training=False)
        assert int(ref["action"][0, 0]) == int(out["action"][i, 0]), (i, ref, out["action"][i])
        if "q" in ref:
            np.testing.assert_allclose(np.asarray(ref["q"]).reshape(-1)[0], out["q"][i, 0], rtol=1e-5, atol=1e-5)
    # the acting copy follows the learner only on sync()
    before = actors.act(obs, training=False)["q"].copy()
    with torch.no_grad():
        agent._net.params.mul_(1.5)
    np.testing.assert_array_equal(actors.act(obs, training=False)["q"], before)
    actors.sync()
    assert not np.allclose(actors.act(obs, training=False)["q"], before)
    # training mode: exploring rows follow the host RNG stream (u < eps -> np.random.randint), others stay greedy
    if name != "rainbow":
        actors.eps[:] = np.where(np.arange(N) % 2 == 0, 1.0, 0.0)
        greedy = actors.act(obs, training=False)["action"]
        np.random.seed(5)
        a = actors.act(obs, training=True)["action"]
        np.random.seed(5)
        np.random.random(N)
        ra = np.random.randint(0, A, size=N)
        assert np.array_equal(a[::2, 0], ra[::2]) and np.array_equal(a[1::2], greedy[1::2])


def test_apex_async_actors_ring_learner_end_to_end():
    """configs[3]'s structure in small: 8 actors act through ONE batched forward per tick on the acting copy (own
    stream, synced every 25 ticks), the vectorised n-step assembler emits transitions + actor-side priorities into the
    lock-free staging ring from the actor thread, the learner thread drains + learns (hipGraph captured while the actor
    thread keeps issuing HIP work).  Accounting must close and the learner must see what the actors produced."""
    import threading
    import time

    from jorldy_amd import ops
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import BatchedValueActors, VecNStepApeX

    torch.manual_seed(0)
    np.random.seed(0)
    N, n = 8, 3
    agent = Agent("ape_x", state_size=4, action_size=2, hidden_size=64, network="dueling", batch_size=32, buffer_size=2048, start_train_step=0, n_step=n,
                  num_workers=N, target_update_period=100, run_step=100000, device="cuda")
    assert agent.backend == "native"
    example = {"state": np.zeros((1, 4), np.float32), "action": np.zeros((1, 1), np.int64), "reward": np.zeros((1, n, 1), np.float32),
               "next_state": np.zeros((1, 4), np.float32), "done": np.zeros((1, n, 1), np.uint8)}
    ring = agent.memory.make_ring(64 * N, example=example, with_priority=True)
    env = ops.CartPoleVec(N, seed=1)
    actors = BatchedValueActors(agent, N)
    nstep = VecNStepApeX(N, n, agent.gamma, (4,), np.float32)
    stop, err, ticks = threading.Event(), [], [0]

    def actor_loop():
        try:
            torch.cuda.set_device(agent.device)
            arng_state = np.random.RandomState(3)
            while not stop.is_set():
                obs = env.obs()
                out = actors.act(obs, training=True)
                _, rew, done = env.step(out["action"])
                emitted = nstep.push(obs, out["action"], rew.reshape(N, 1), done.reshape(N, 1).astype(np.float32), out["q"])
                if emitted is not None:
                    cols, prio = emitted
                    ring.produce(agent.memory.ring_columns(cols), prio + 1e-3, timeout_ms=2000)
                ticks[0] += 1
                if ticks[0] % 25 == 0:
                    actors.sync()
        except Exception as e:  # surfaced by the main thread
            err.append(e)

    th = threading.Thread(target=actor_loop, daemon=True)
    th.start()
    losses, step = [], 0
    t_end = time.time() + 20
    while len(losses) < 150 and time.time() < t_end and not err:
        step += 1
        agent.learn_period_stamp = agent.learn_period
        r = agent.process(None, step)
        if r:
            losses.append(r["loss"])
    stop.set()
    th.join(timeout=10)
    assert not err, err
    assert len(losses) >= 150 and np.all(np.isfinite(losses)), (len(losses), ticks[0])
    assert agent._graph is not None, "learn() was not captured while the actor thread was running"
    st = ring.stats()
    agent.process(None, step + 1)  # take what the actors published last
    st2 = ring.stats()
    assert st2["produced"] == st2["drained"] and st["produced"] >= N * (ticks[0] - n - 1) > 0
    assert agent.num_transitions == st2["drained"] and agent.memory.size == min(agent.num_transitions, 2048)
    tree = agent.memory.sum_tree
    leaves = tree[agent.memory.first_leaf_index : agent.memory.first_leaf_index + agent.memory.size]
    assert np.all(leaves > 0) and tree[0] == pytest.approx(leaves.sum(), rel=1e-9)
