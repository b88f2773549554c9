"""Batched on-GPU acting of the value-net agents: jh_value_act against the reference's expression on the CPU (float64), and
BatchedValueActors (one forward per tick for all actors on an acting copy of the native network) against the agents'
own act() (ape_x.py:64-77, rainbow.py:140-152, dqn.py:76-92) row by row."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,A,K", [(64, 6, 1), (7, 4, 51), (130, 3, 21)])
def test_value_act_matches_the_cpu_restatement(N, A, K):
    """jh_value_act (rainbow.py:285-292 logits2Q, argmax, epsilon-greedy) against the reference's expression evaluated on the CPU in
    float64, with torch-CPU float32 -- the reference's own arithmetic -- beside it (round 5: rounds 1-4 compared with torch ON THE GPU,
    a vendor library, against DESIGN's own rule; VERDICT r4 weak #1 iv).  The greedy action must be the float64 argmax wherever the
    two best Q values are further apart than fp32 rounding."""
    import fp64_truth as T64
    from jorldy_amd import ops

    torch.manual_seed(N)
    logits_cpu = torch.randn(N, A, K)
    logits = logits_cpu.cuda()
    v_min, v_max = -1.0, 10.0

    def q_of(lg):
        if K == 1:
            return lg[:, :, 0]
        z = torch.linspace(v_min, v_max, K, dtype=lg.dtype)
        return (torch.exp(torch.log_softmax(lg, -1)) * z).sum(-1)

    q64, q32 = q_of(logits_cpu.double()), q_of(logits_cpu)
    act, q, q_all = ops.value_act(logits, v_min, v_max, want_q_all=True)
    T64.vs_exact(q_all, q64, q32, 1e-5, "Q(s, a)")
    top2 = torch.topk(q64, 2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5 * (1.0 + top2[:, 0].abs())
    assert clear.float().mean() > 0.9
    assert torch.equal(act.cpu()[clear], q64.argmax(-1)[clear])
    T64.vs_exact(q, q64.max(-1).values, q32.max(-1).values, 1e-5, "max_a Q")
    # epsilon-greedy with the host's draws: rows with u < eps take the host's random action, and report ITS q
    rng = np.random.RandomState(0)
    eps, u, ra = rng.rand(N).astype(np.float32), rng.rand(N), rng.randint(0, A, size=N)
    act2, q2, _ = ops.value_act(logits, v_min, v_max, eps, u, ra)
    explore = u < eps.astype(np.float64)
    assert np.array_equal(act2.cpu().numpy()[explore], ra[explore])
    assert np.array_equal(act2.cpu().numpy()[~explore], act.cpu().numpy()[~explore])
    taken = torch.from_numpy(act2.cpu().numpy())
    T64.vs_exact(q2, q64[torch.arange(N), taken], q32[torch.arange(N), taken], 1e-5, "Q of the action taken")


@pytest.mark.parametrize("name,extra,S", [
    ("rainbow", dict(head="cnn", n_step=3, num_support=21, v_min=-1, v_max=10), (4, 44, 52)),
    ("c51", dict(head="mlp", num_support=21, v_min=-1, v_max=10), 6),
    ("dqn", dict(), 6),
    ("double", dict(head="cnn"), (4, 44, 52)),
])
def test_native_act_branch_equals_the_generic_one(name, extra, S):
    """act()'s network branch through pinned slab -> forward -> jh_value_act -> device-mapped actions (NativeValueNetMixin._act_greedy)
    against the reference's expression evaluated with torch ops on the same forward (dqn.py:100-115, c51.py:50-66, rainbow.py:140-152):
    the same actions, row by row and for whole batches; Rainbow in training mode consumes the same torch.randn draws in the same order."""
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(3)
    np.random.seed(3)
    A = 5
    agent = Agent(name, state_size=S, action_size=A, hidden_size=32, batch_size=8, buffer_size=64, start_train_step=0, device="cuda", **extra)
    with torch.no_grad():
        agent._net.params.add_(0.05 * torch.randn_like(agent._net.params))
    agent.epsilon = agent.epsilon_eval = 0.0
    if name == "rainbow":
        agent.memory.buffer_counter = 10 ** 6  # past the warm-up branch of Rainbow.act
    rng = np.random.RandomState(4)

    def generic(x, training):
        if name == "rainbow":
            lg = agent.network(agent.as_tensor(x), training)
        else:
            lg = agent.network(agent.as_tensor(x))
        q = agent.logits2Q(lg)[1] if hasattr(agent, "logits2Q") else lg
        top2 = torch.topk(q.double(), 2, dim=-1).values
        clear = ((top2[:, 0] - top2[:, 1]) > 1e-5 * (1.0 + top2[:, 0].abs())).cpu().numpy()
        return torch.argmax(q, -1, keepdim=True).cpu().numpy(), clear

    n_clear = 0
    for N in (1, 3, 8):
        for training in (False, True):
            x = rng.randint(0, 256, size=(N,) + S).astype(np.uint8) if isinstance(S, tuple) else rng.randn(N, S).astype(np.float32)
            torch.manual_seed(100 + N)
            want, clear = generic(x, training)
            torch.manual_seed(100 + N)
            got = agent.act(x, training)["action"]
            assert got.shape == (N, 1) and got.dtype == np.int64
            assert np.array_equal(got[clear], want[clear]), (N, training, got.ravel(), want.ravel())
            n_clear += int(clear.sum())
            torch.manual_seed(100 + N)
            assert np.array_equal(agent._act_greedy(x, training), got)  # and it IS the branch act() took
    assert n_clear >= 20
    assert agent._act_greedy(np.zeros((agent._net.maxB + 1,) + (S if isinstance(S, tuple) else (S,)), np.uint8 if isinstance(S, tuple) else np.float32)) is None


def test_act_takes_what_the_references_as_tensor_takes():
    """base.py:61-73 as_tensor accepts torch tensors (CPU or CUDA) and any numeric dtype: act() gives the float32-ndarray answer for a
    float64 ndarray (native branch, down-cast into the slab), for CPU / CUDA tensors and for a list-of-arrays state (generic branch)
    instead of raising inside the native branch."""
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(5)
    np.random.seed(5)
    agent = Agent("dqn", state_size=6, action_size=5, hidden_size=32, batch_size=8, buffer_size=64, start_train_step=0, device="cuda")
    with torch.no_grad():
        agent._net.params.add_(0.05 * torch.randn_like(agent._net.params))
    agent.epsilon = agent.epsilon_eval = 0.0
    x = np.random.RandomState(6).randn(4, 6).astype(np.float32)
    want = agent.act(x, False)["action"]
    assert agent._act_greedy(x.astype(np.float64)) is not None and np.array_equal(agent.act(x.astype(np.float64), False)["action"], want)
    for other in (torch.from_numpy(x), torch.from_numpy(x).cuda(), torch.from_numpy(x).double()):
        assert agent._act_greedy(other) is None  # not the native branch ...
        assert np.array_equal(agent.act(other, False)["action"], want)  # ... and the generic one answers
    assert agent._act_greedy(x.astype(np.complex64)) is None


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


def _sliding_stack_env(rng, N, C, shape, p_reset):
    """Per tick: every actor's stack slides by one new frame (core/env/atari.py:145-149), or is refilled (reset)."""
    stacks = rng.randint(0, 256, size=(N, C) + shape).astype(np.uint8)
    while True:
        yield stacks.copy()
        fresh = rng.rand(N) < p_reset
        new = rng.randint(0, 256, size=(N,) + shape).astype(np.uint8)
        stacks[:, :-1] = stacks[:, 1:]
        stacks[:, -1] = new
        if fresh.any():
            stacks[fresh] = rng.randint(0, 256, size=(int(fresh.sum()), C) + shape).astype(np.uint8)


@pytest.mark.parametrize("shape,N,C,n", [((12, 12), 5, 4, 3), ((5, 10), 3, 4, 1), ((7, 9), 4, 1, 2), ((84, 84), 6, 4, 3)])
def test_device_feed_rows_and_priorities_equal_the_host_assembler(shape, N, C, n):
    """jh_feed_tick (plane de-duplication + n-step assembly + actor-side priorities in HBM) against the host path it
    replaces (VecNStepApeX -> PERBuffer.store_soa with full stacks): after the buffer AND the plane rings have wrapped
    several times, every live row decodes to the same stacks / action / reward / done, and the two sum trees are
    bit-identical.  Shapes include planes that are not a multiple of 16 bytes and C = 1."""
    from jorldy_amd.core.buffer import PERBuffer
    from jorldy_amd.manager import VecNStepApeX

    rng = np.random.RandomState(11)
    cap, gamma, eps, A = 8 * N, 0.99, 1e-3, 6
    dev = torch.device("cuda")
    host = PERBuffer(cap, 1e-3, device=dev)
    host.first_store = False
    fed = PERBuffer(cap, 1e-3, device=dev)
    pool = fed.attach_actor_feed(N, (C,) + shape, n, gamma, pool_factor=1.6, in_flight_ticks=2)
    nstep = VecNStepApeX(N, n, gamma, (C,) + shape, np.uint8)
    env = _sliding_stack_env(rng, N, C, shape, 0.08)
    out = {"state": torch.empty(N, C, dtype=torch.int64, device=dev), "next_state": torch.empty(N, C, dtype=torch.int64, device=dev),
           "action": torch.empty(N, dtype=torch.int64, device=dev), "reward": torch.empty(N, n, dtype=torch.float32, device=dev),
           "done": torch.empty(N, n, dtype=torch.uint8, device=dev), "priority": torch.empty(N, dtype=torch.float64, device=dev)}
    obs_dev = [torch.empty((N, C) + shape, dtype=torch.uint8, device=dev) for _ in range(2)]
    T = 12 * (cap // N) + 7
    for t in range(T):
        obs = next(env)
        action = rng.randint(0, A, size=(N, 1))
        q = rng.randn(N, 1).astype(np.float32)
        reward = rng.choice([-1.0, 0.0, 1.0], size=(N, 1)).astype(np.float32)
        done = (rng.rand(N, 1) < 0.1).astype(np.float32)
        obs_dev[t & 1].copy_(torch.from_numpy(obs))
        got = pool.feed.tick(obs_dev[t & 1], obs_dev[(t + 1) & 1] if t else None, pool.planes, torch.from_numpy(action.reshape(-1)).to(dev),
                             torch.from_numpy(q.reshape(-1)).to(dev), reward, done, out, eps)
        emitted = nstep.push(obs, action, reward, done, q)
        assert (got == N) == (emitted is not None)
        if got:
            cols, prio = emitted
            host.store_soa(cols, prio + eps)
            fed.store_feed_rows({"state": out["state"], "action": out["action"].view(N, 1), "reward": out["reward"].view(N, n, 1),
                                 "next_state": out["next_state"], "done": out["done"].view(N, n, 1)}, N, out["priority"])
            assert np.array_equal(out["priority"].cpu().numpy(), prio + eps)  # float32 fold, float64 leaf: bit for bit
    assert host.buffer_index == fed.buffer_index and host.buffer_counter == fed.buffer_counter == cap
    idx = torch.arange(cap, dtype=torch.int64, device=dev)
    a, b = host.gather(idx, as_float=False), fed.gather(idx, as_float=False)
    for k in ("state", "next_state", "action", "reward", "done"):
        assert torch.equal(a[k].reshape(cap, -1).to(torch.float64), b[k].reshape(cap, -1).to(torch.float64)), k
    assert np.array_equal(host.sum_tree, fed.sum_tree) and host.max_priority == fed.max_priority
    st = pool.stats()  # also raises on a plane-ring overrun
    # ~1 plane per actor and tick + C - 1 more per reset, against 2 C planes per transition stored plain
    assert st["planes_written"] <= N * T * (1 + 0.2 * (C - 1)) + N * C
    assert st["planes_written"] > 2 * pool.F, "the plane rings did not wrap in this test"


def test_device_feed_reports_a_plane_ring_overrun():
    """Stacks that are discontinuous on EVERY tick write C planes per tick: more than the ring was sized for -> the
    device flag is raised and surfaces as an error instead of rows silently decoding to newer frames."""
    from jorldy_amd.core.buffer import ReplayBuffer

    N, C, n, shape, dev = 2, 4, 2, (4, 4), torch.device("cuda")
    buf = ReplayBuffer(16, device=dev)
    pool = buf.attach_actor_feed(N, (C,) + shape, n, 0.99, pool_factor=1.0, in_flight_ticks=0)
    rng = np.random.RandomState(0)
    out = {"state": torch.empty(N, C, dtype=torch.int64, device=dev), "next_state": torch.empty(N, C, dtype=torch.int64, device=dev),
           "action": torch.empty(N, dtype=torch.int64, device=dev), "reward": torch.empty(N, n, dtype=torch.float32, device=dev),
           "done": torch.empty(N, n, dtype=torch.uint8, device=dev), "priority": torch.empty(N, dtype=torch.float64, device=dev)}
    obs = [torch.empty((N, C) + shape, dtype=torch.uint8, device=dev) for _ in range(2)]
    z, zq = torch.zeros(N, dtype=torch.int64, device=dev), torch.zeros(N, dtype=torch.float32, device=dev)
    for t in range(3 * pool.window):
        obs[t & 1].copy_(torch.from_numpy(rng.randint(0, 256, size=(N, C) + shape).astype(np.uint8)))
        pool.feed.tick(obs[t & 1], obs[(t + 1) & 1] if t else None, pool.planes, z, zq, np.zeros(N, np.float32), np.zeros(N, np.float32), out)
    with pytest.raises(RuntimeError, match="plane ring overrun"):
        pool.check()


def test_apex_device_feed_learner_end_to_end():
    """configs[3]'s structure with the device-resident feed: 8 actors on synthetic frame-stack envs act through one
    batched forward per tick; their stacks never leave HBM again (DeviceActorFeed: plane pool + n-step rows + actor-side
    priorities on the acting stream); the learner thread drains device-to-device and learns (captured graph).  The
    accounting closes, nothing is dropped, and what the learner stored decodes to what the actors saw."""
    import threading
    import time

    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import BatchedValueActors, DeviceActorFeed, VecNStepApeX

    torch.manual_seed(0)
    np.random.seed(0)
    N, n, C, shape, cap = 8, 3, 4, (44, 52), 1024
    agent = Agent("ape_x", state_size=[C, 44, 52], action_size=4, hidden_size=64, network="dueling", head="cnn", batch_size=32, buffer_size=cap,
                  start_train_step=0, n_step=n, num_workers=N, target_update_period=100, run_step=100000, device="cuda")
    assert agent.backend == "native"
    actors = BatchedValueActors(agent, N)
    feed = DeviceActorFeed(actors, agent.memory, n, agent.gamma, depth=8, prio_eps=1e-3)
    shadow = VecNStepApeX(N, n, agent.gamma, (C,) + shape, np.uint8)  # the host assembler, fed the same ticks
    rng = np.random.RandomState(5)
    env = _sliding_stack_env(rng, N, C, shape, 0.02)
    err, ticks, log = [], [0], []
    stop = threading.Event()

    def actor_loop():
        try:
            torch.cuda.set_device(agent.device)
            while not stop.is_set():
                obs = next(env)
                out = feed.act(obs, training=True)
                reward = rng.choice([-1.0, 0.0, 1.0], size=(N, 1)).astype(np.float32)
                done = (rng.rand(N, 1) < 0.01).astype(np.float32)
                emitted = shadow.push(obs, out["action"], reward, done, out["q"])
                got = feed.push(reward, done)
                if got < 0:
                    return
                if emitted is not None:
                    log.append(({k: np.array(v) for k, v in emitted[0].items()}, emitted[1] + 1e-3))  # copies: the assembler's outputs are views
                    if len(log) > 4:
                        log[len(log) - 5] = None
                ticks[0] += 1
                if ticks[0] % 25 == 0:
                    actors.sync()
        except Exception as e:
            err.append(e)

    th = threading.Thread(target=actor_loop, daemon=True)
    th.start()
    losses, step = [], 0
    t_end = time.time() + 30
    while len(losses) < 100 and time.time() < t_end and not err:
        step += 1
        agent.learn_period_stamp = agent.learn_period
        r = agent.process(None, step)
        if r:
            losses.append(r["loss"])
    stop.set()
    feed.close()
    th.join(timeout=10)
    assert not err, err
    assert len(losses) >= 100 and np.all(np.isfinite(losses)), (len(losses), ticks[0])
    assert agent._graph is not None
    agent.memory.drain()
    st = feed.stats()
    assert st["stored_rows"] == st["emissions"] * N == agent.memory._frames.rows_stored and st["emissions"] >= ticks[0] - n > 0
    assert st["planes_per_stored_row"] < 2.0, st  # ~1 plane per env step (+ resets) against 8 stored plain
    # the last rows the learner stored are the last transitions the host assembler produced from the same ticks
    m = agent.memory
    want_cols, want_prio = log[st["emissions"] - 1]
    last = (m.buffer_index - N) % m.buffer_size
    idx = (last + torch.arange(N, device=agent.device)) % m.buffer_size
    got = m.gather(idx, as_float=False)
    for key in ("state", "next_state", "action", "reward", "done"):
        assert np.array_equal(got[key].cpu().numpy().reshape(N, -1).astype(np.float64), np.asarray(want_cols[key]).reshape(N, -1).astype(np.float64)), key


@pytest.mark.parametrize("shape,N,C,n", [((12, 12), 5, 4, 3), ((5, 10), 3, 4, 1), ((7, 9), 4, 1, 2), ((84, 84), 6, 4, 3)])
def test_device_feed_frame_mode_equals_host_stacking_and_assembler(shape, N, C, n):
    """Frame mode (jh_feed_push_frames): the env hands over only its newest frame + a reset flag.  The stacks rebuilt in HBM
    for the acting forward equal the wrapper's host stacking tick by tick (core/env/atari.py:112 np.tile on reset, :147
    slide otherwise), and the stored rows / sum tree equal VecNStepApeX -> PERBuffer.store_soa on those host stacks."""
    from jorldy_amd.core.buffer import PERBuffer
    from jorldy_amd.manager import VecNStepApeX

    rng = np.random.RandomState(23)
    cap, gamma, eps, A = 8 * N, 0.99, 1e-3, 6
    dev = torch.device("cuda")
    host = PERBuffer(cap, 1e-3, device=dev)
    host.first_store = False
    fed = PERBuffer(cap, 1e-3, device=dev)
    pool = fed.attach_actor_feed(N, (C,) + shape, n, gamma, pool_factor=1.1, in_flight_ticks=2)
    nstep = VecNStepApeX(N, n, gamma, (C,) + shape, np.uint8)
    out = {"state": torch.empty(N, C, dtype=torch.int64, device=dev), "next_state": torch.empty(N, C, dtype=torch.int64, device=dev),
           "action": torch.empty(N, dtype=torch.int64, device=dev), "reward": torch.empty(N, n, dtype=torch.float32, device=dev),
           "done": torch.empty(N, n, dtype=torch.uint8, device=dev), "priority": torch.empty(N, dtype=torch.float64, device=dev)}
    stack_dev = torch.empty((N, C) + shape, dtype=torch.uint8, device=dev)
    stacks = np.zeros((N, C) + shape, np.uint8)
    T = 12 * (cap // N) + 7
    for t in range(T):
        frames = rng.randint(0, 256, size=(N,) + shape).astype(np.uint8)
        reset = (rng.rand(N) < 0.1) | (t == 0)
        for a in range(N):  # the env wrapper on the host
            if reset[a]:
                stacks[a] = np.tile(frames[a], (C, 1, 1))
            else:
                stacks[a] = np.concatenate((stacks[a][1:], frames[a][None]), axis=0)
        pool.feed.push_frames(torch.from_numpy(frames).to(dev), reset if t else np.zeros(N, np.uint8), pool.planes, stack_dev)  # tick 0 is a reset by itself
        assert torch.equal(stack_dev.cpu(), torch.from_numpy(stacks)), t
        action = rng.randint(0, A, size=(N, 1))
        q = rng.randn(N, 1).astype(np.float32)
        reward = rng.choice([-1.0, 0.0, 1.0], size=(N, 1)).astype(np.float32)
        done = (rng.rand(N, 1) < 0.1).astype(np.float32)
        got = pool.feed.emit(torch.from_numpy(action.reshape(-1)).to(dev), torch.from_numpy(q.reshape(-1)).to(dev), reward, done, out, eps)
        emitted = nstep.push(stacks, action, reward, done, q)
        assert (got == N) == (emitted is not None)
        if got:
            cols, prio = emitted
            host.store_soa(cols, prio + eps)
            fed.store_feed_rows({"state": out["state"], "action": out["action"].view(N, 1), "reward": out["reward"].view(N, n, 1),
                                 "next_state": out["next_state"], "done": out["done"].view(N, n, 1)}, N, out["priority"])
    idx = torch.arange(cap, dtype=torch.int64, device=dev)
    a, b = host.gather(idx, as_float=False), fed.gather(idx, as_float=False)
    for k in ("state", "next_state", "action", "reward", "done"):
        assert torch.equal(a[k].reshape(cap, -1).to(torch.float64), b[k].reshape(cap, -1).to(torch.float64)), k
    assert np.array_equal(host.sum_tree, fed.sum_tree) and host.max_priority == fed.max_priority
    st = pool.stats()
    assert st["planes_written"] == N * T and st["planes_written"] > 2 * pool.F  # exactly one plane per env step; the rings wrapped
    if shape == (12, 12):  # checkpoints of a fed buffer are the portable full-stack form: they restore into a plain buffer
        import tempfile

        with tempfile.TemporaryDirectory() as tmp:
            meta = fed.save_stream(tmp)
            plain = PERBuffer(cap, 1e-3, device=dev)
            plain.load_stream(tmp, meta)
            with pytest.raises(RuntimeError, match="sink of a DeviceActorFeed"):
                fed._feeds.append(object())
                fed.load_stream(tmp, meta)
        c = plain.gather(idx, as_float=False)
        for k in ("state", "next_state", "action", "reward", "done"):
            assert torch.equal(a[k].reshape(cap, -1).to(torch.float64), c[k].reshape(cap, -1).to(torch.float64)), k
        assert np.array_equal(host.sum_tree, plain.sum_tree) and plain.buffer_index == host.buffer_index


def test_device_actor_feed_frame_mode_through_the_agent():
    """DeviceActorFeed.act_frames / push / agent.process in one thread: the envs hand over their newest frame only; the
    actions come from the stacks rebuilt in HBM (equal to the agent's own act() on the host-built stacks), and what the
    learner's buffer holds afterwards is what the host wrapper + assembler produce from the same frames."""
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import BatchedValueActors, DeviceActorFeed, VecNStepApeX

    torch.manual_seed(0)
    np.random.seed(0)
    N, n, C, shape, cap = 4, 3, 4, (44, 52), 64
    agent = Agent("ape_x", state_size=[C, 44, 52], action_size=4, hidden_size=64, network="dueling", head="cnn", batch_size=16, buffer_size=cap,
                  start_train_step=10**9, n_step=n, num_workers=N, target_update_period=100, run_step=100000, device="cuda")
    actors = BatchedValueActors(agent, N, epsilons=np.zeros(N))  # greedy: comparable with the agent's own act
    feed = DeviceActorFeed(actors, agent.memory, n, agent.gamma, depth=4, prio_eps=1e-3)
    shadow = VecNStepApeX(N, n, agent.gamma, (C,) + shape, np.uint8)
    rng = np.random.RandomState(9)
    stacks = np.zeros((N, C) + shape, np.uint8)
    last = None
    for t in range(24):
        frames = rng.randint(0, 256, size=(N,) + shape).astype(np.uint8)
        reset = (rng.rand(N) < 0.15) | (t == 0)
        for a in range(N):
            stacks[a] = np.tile(frames[a], (C, 1, 1)) if reset[a] else np.concatenate((stacks[a][1:], frames[a][None]), axis=0)
        out = feed.act_frames(frames, reset if t else None, training=True)
        if t % 8 == 0:  # the stacks rebuilt on the device give the actions the agent computes from the host-built stacks
            agent.epsilon = 0.0
            for a in range(N):
                assert int(agent.act(stacks[a][None], training=False)["action"][0, 0]) == int(out["action"][a, 0]), (t, a)
        reward = rng.choice([-1.0, 0.0, 1.0], size=(N, 1)).astype(np.float32)
        done = (rng.rand(N, 1) < 0.05).astype(np.float32)
        emitted = shadow.push(stacks, out["action"], reward, done, out["q"])
        assert feed.push(reward, done) == (N if emitted is not None else 0)
        if emitted is not None:
            last = ({k: np.array(v) for k, v in emitted[0].items()}, emitted[1] + 1e-3)
        agent.process(None, t + 1)  # drains (start_train_step keeps learn() out of this test)
    m = agent.memory
    assert feed.stats()["stored_rows"] == (24 - n) * N == agent.num_transitions
    idx = ((m.buffer_index - N) % m.buffer_size + torch.arange(N, device=agent.device)) % m.buffer_size
    got = m.gather(idx, as_float=False)
    for key in ("state", "next_state", "action", "reward", "done"):
        assert np.array_equal(got[key].cpu().numpy().reshape(N, -1).astype(np.float64), np.asarray(last[0][key]).reshape(N, -1).astype(np.float64)), key
    leaves = m.sum_tree[m.first_leaf_index + idx.cpu().numpy()]
    assert np.array_equal(leaves, last[1])


def test_device_feed_full_size_chain_property():
    """configs[3] shapes (64 actors, (4,84,84) frames, n = 3) at a size the host assembler is too slow to shadow in a
    test: size-independent properties of what the feed stored.  (a) next_state of the row (tick k, actor a) is the state
    of the row (tick k + n, actor a); (b) a state is its predecessor's shifted by one frame unless the env was reset,
    where it is one frame C times; (c) the newest plane of every state is the frame the env handed over at that tick
    (checked through a per-frame checksum); (d) exactly one plane per env step was written and nothing overran."""
    from jorldy_amd.core.buffer import PERBuffer

    N, C, n, shape, dev = 64, 4, 3, (84, 84), torch.device("cuda")
    ticks_kept = 96
    cap = N * ticks_kept
    buf = PERBuffer(cap, 1e-3, device=dev)
    pool = buf.attach_actor_feed(N, (C,) + shape, n, 0.99, pool_factor=1.1, in_flight_ticks=2)
    g = torch.Generator(device=dev).manual_seed(5)
    out = {"state": torch.empty(N, C, dtype=torch.int64, device=dev), "next_state": torch.empty(N, C, dtype=torch.int64, device=dev),
           "action": torch.empty(N, dtype=torch.int64, device=dev), "reward": torch.empty(N, n, dtype=torch.float32, device=dev),
           "done": torch.empty(N, n, dtype=torch.uint8, device=dev), "priority": torch.empty(N, dtype=torch.float64, device=dev)}
    stack = torch.empty((N, C) + shape, dtype=torch.uint8, device=dev)
    rng = np.random.RandomState(3)
    T = 3 * ticks_kept + 11
    sums, resets = [], []
    zq = torch.zeros(N, dtype=torch.float32, device=dev)
    for t in range(T):
        frames = torch.randint(0, 256, (N,) + shape, dtype=torch.uint8, device=dev, generator=g)
        reset = (rng.rand(N) < 0.02) | (t == 0)
        sums.append(frames.view(N, -1).to(torch.int64).sum(1))  # checksum of the frame handed over at tick t
        resets.append(reset.copy())
        pool.feed.push_frames(frames, reset if t else np.zeros(N, np.uint8), pool.planes, stack)
        act = torch.full((N,), t % 7, dtype=torch.int64, device=dev)
        got = pool.feed.emit(act, zq, np.zeros(N, np.float32), np.zeros(N, np.float32), out, 1e-3)
        if got:
            buf.store_feed_rows({"state": out["state"], "action": out["action"].view(N, 1), "reward": out["reward"].view(N, n, 1),
                                 "next_state": out["next_state"], "done": out["done"].view(N, n, 1)}, N, out["priority"])
    st = pool.stats()
    assert st["planes_written"] == N * T
    # the buffer holds the emissions of ticks T - ticks_kept .. T - 1; emission e (made at tick e + n) sits in ring rows
    # (e * N + a) % cap and describes (state of tick e, next_state of tick e + n)
    first_e, last_e = T - n - ticks_kept, T - n - 1

    def rows_of(e):
        return ((e * N + torch.arange(N, device=dev)) % cap).to(torch.int64)

    for e in (first_e, first_e + 1, (first_e + last_e) // 2, last_e - n, last_e - n - 1):
        a = buf.gather(rows_of(e), as_float=False)
        b = buf.gather(rows_of(e + n), as_float=False)
        assert torch.equal(a["next_state"], b["state"]), e                                   # (a)
        assert torch.equal(a["action"].view(-1), torch.full((N,), e % 7, device=dev, dtype=a["action"].dtype))
        s0, s1 = buf.gather(rows_of(e), as_float=False)["state"], buf.gather(rows_of(e + 1), as_float=False)["state"]
        r1 = torch.from_numpy(resets[e + 1]).to(dev)
        slid = (s1[:, :-1] == s0[:, 1:]).flatten(1).all(1)
        tiled = (s1 == s1[:, -1:].expand_as(s1)).flatten(1).all(1)
        assert bool((slid | r1).all()) and bool((tiled | ~r1).all()), e                       # (b)
        assert torch.equal(s1[:, -1].reshape(N, -1).to(torch.int64).sum(1), sums[e + 1]), e  # (c)


@pytest.mark.parametrize("frames", [False, True])
def test_full_checkpoint_resumes_a_device_fed_replay_bit_identically(tmp_path, frames):
    """SURVEY.md 8f rank 4 / VERDICT r2 "missing" #4: save_full of an Ape-X learner whose replay is fed on the device (plane rings,
    per-actor cursors, rolling n-step windows, rows with slot numbers, sum tree) -> load_full into a FRESH agent + feed of the same
    geometry -> the continuation (acting, feeding, draining, learning) is bit-identical to the uninterrupted run: same stored rows,
    same decoded stacks, same float64 tree, same losses, same weights.  The buffer (96 slots) and the plane rings wrap before the
    checkpoint."""
    from jorldy_amd.core.agent import Agent
    from jorldy_amd.manager import BatchedValueActors, DeviceActorFeed

    N, n, C, shape, cap, T1, T2 = 6, 3, 4, (44, 52), 96, 40, 30
    rs = np.random.RandomState(77)
    script = []
    stacks = rs.randint(0, 256, size=(N, C) + shape).astype(np.uint8)
    for t in range(T1 + T2):
        reset = (rs.rand(N) < 0.04) if t else np.ones(N, bool)
        new = rs.randint(0, 256, size=(N,) + shape).astype(np.uint8)
        if t:
            stacks[:, :-1] = stacks[:, 1:]
            stacks[:, -1] = new
        stacks[reset] = np.repeat(new[reset][:, None], C, 1)  # a reset stack is the new frame C times (core/env/atari.py:112)
        script.append((stacks.copy(), new.copy(), reset.astype(np.uint8), rs.choice([-1.0, 0.0, 1.0], size=(N, 1)).astype(np.float32),
                       (rs.rand(N, 1) < 0.03).astype(np.float32)))

    def mk():
        torch.manual_seed(0)
        np.random.seed(0)
        agent = Agent("ape_x", state_size=[C, 44, 52], action_size=4, hidden_size=64, network="dueling", head="cnn", batch_size=16, buffer_size=cap,
                      start_train_step=0, n_step=n, num_workers=N, target_update_period=7, run_step=100000, device="cuda", use_graph=False)
        actors = BatchedValueActors(agent, N)
        return agent, actors, DeviceActorFeed(actors, agent.memory, n, agent.gamma, depth=8, prio_eps=1e-3)

    def run(agent, actors, feed, t0, t1, out):
        for t in range(t0, t1):
            st, new, reset, reward, done = script[t]
            act = feed.act_frames(new, reset, training=True) if frames else feed.act(st, training=True)
            assert feed.push(reward, done) >= 0
            if t >= 8 and t % 2 == 0:
                agent.learn_period_stamp = agent.learn_period
                r = agent.process(None, t)
                if r:
                    out.append((r["loss"], r["max_Q"], act["action"].copy()))
            if t % 10 == 9:
                actors.sync()

    def snapshot(agent):
        agent.memory.drain()
        torch.cuda.synchronize()
        m = agent.memory
        idx = torch.arange(m.buffer_counter, device=agent.device)
        rows = {k: v.cpu().numpy().copy() for k, v in m.gather(idx, as_float=False).items()}
        return rows, m.sum_tree.copy(), m.buffer_index, m.buffer_counter, agent._net.params.cpu().numpy().copy(), agent._net.target.cpu().numpy().copy()

    a, a_act, a_feed = mk()
    log_a = []
    run(a, a_act, a_feed, 0, T1, log_a)
    a_act.sync()  # the acting copy at the checkpoint = the learner's weights (a resumed run starts from a sync)
    a.save_full(str(tmp_path))
    tail_a = []
    run(a, a_act, a_feed, T1, T1 + T2, tail_a)
    want = snapshot(a)
    assert a.memory._frames.rows_stored > 2 * cap  # wrapped

    b, b_act, b_feed = mk()
    b.load_full(str(tmp_path))
    b_act.sync()
    tail_b = []
    run(b, b_act, b_feed, T1, T1 + T2, tail_b)
    got = snapshot(b)
    assert len(tail_a) == len(tail_b) > 5
    for (la, qa, aa), (lb, qb, ab) in zip(tail_a, tail_b):
        assert la == lb and qa == qb and np.array_equal(aa, ab)
    assert want[2:4] == got[2:4]
    for k in want[0]:
        assert np.array_equal(want[0][k], got[0][k]), k
    assert np.array_equal(want[1], got[1]), "sum tree"
    assert np.array_equal(want[4], got[4]) and np.array_equal(want[5], got[5]), "weights"
    # a checkpoint of a plain buffer still cannot be loaded into a fed one (its rows have no planes behind them) -- and says so
    import json

    mp = os.path.join(str(tmp_path), "resume", "manifest.json")
    man = json.load(open(mp))
    man["memory"].pop("feed")
    json.dump(man, open(mp, "w"))
    c, _, _ = mk()
    with pytest.raises(RuntimeError, match="sink of a DeviceActorFeed"):
        c.load_full(str(tmp_path))
