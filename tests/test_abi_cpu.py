"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol the
header declares, fails loudly without a GPU, and its host-side collector matches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    from jorldy_amd import _lib

    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "jorldy_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol(lib):
    from jorldy_amd import _lib

    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/jorldy_hip.h but not exported"
    assert set(syms) == set(_lib.exported_names()), "ctypes binding table out of sync with the header"
    assert lib.jh_abi_version() == 2


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "jorldy_hip.h")).read()
    for ref in ("per_buffer.py:", "replay_buffer.py:", "rollout_buffer.py:", "ppo.py:", "rainbow.py:", "dqn.py:", "distributed_manager.py:"):
        assert ref in src


def test_fails_loudly_without_gpu(lib):
    import torch

    from jorldy_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.jh_device_count() == 0
    h = C.c_void_p()
    rc = lib.jh_ctx_create(0, C.byref(h))
    assert rc == -5  # JH_ERR_NODEVICE
    assert b"no HIP device" in lib.jh_last_error()
    with pytest.raises(_lib.JhError):
        _lib.ctx(0)


def test_native_cartpole_matches_oracle_bit_exact(lib):
    from jorldy_amd.ops import CartPoleVec
    from oracle.jorldy_oracle import CartPoleOracle

    W = 5
    nat, orc = CartPoleVec(W, seed=11), CartPoleOracle(W, seed=11)
    rng = np.random.RandomState(2)
    np.testing.assert_array_equal(nat.obs(), orc.obs())
    dones = 0
    for t in range(700):
        a = rng.randint(0, 2, size=W)
        n1, r1, d1 = nat.step(a)
        n2, r2, d2 = orc.step(a)
        np.testing.assert_array_equal(n1, n2)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(d1.astype(bool), d2)
        np.testing.assert_array_equal(nat.obs(), orc.obs())  # includes post-reset states
        dones += int(d2.sum())
    assert dones > 20


def test_staging_ring_many_producers_one_consumer_host_only():
    """jh_ring_* (the async Ape-X transport) without a GPU: 8 producer threads x 400 rows through a 64-slot ring
    (so producers block on a full ring and slots recycle many times); every row arrives exactly once, rows of one
    produce() call stay contiguous and in order, priorities travel with their rows."""
    import threading

    import numpy as np

    from jorldy_amd import _lib as L
    from jorldy_amd import ops

    cols = [("state", L.JH_F32, 3, (3,)), ("action", L.JH_I64, 1, (1,)), ("frame", L.JH_U8, 16, (16,))]
    ring = ops.StagingRing(64, cols, with_priority=True, device=None)
    P, CHUNKS, ROWS = 8, 80, 5
    errors = []

    def actor(pid):
        try:
            for c in range(CHUNKS):
                base = (pid * CHUNKS + c) * ROWS
                ids = np.arange(base, base + ROWS)
                ring.produce({"state": np.stack([ids, ids * 2, ids * 3], 1).astype(np.float32), "action": ids.reshape(-1, 1),
                              "frame": np.repeat((ids % 251).astype(np.uint8)[:, None], 16, 1)}, priorities=ids + 0.5, timeout_ms=20000)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=actor, args=(p,)) for p in range(P)]
    for t in threads:
        t.start()
    got_ids, got = [], 0
    total = P * CHUNKS * ROWS
    import time

    t0 = time.time()
    while got < total and time.time() - t0 < 60:
        out, prio = ring.consume_host()
        n = len(prio)
        if n == 0:
            time.sleep(0.0005)
            continue
        ids = out["action"][:, 0]
        np.testing.assert_array_equal(out["state"], np.stack([ids, ids * 2, ids * 3], 1).astype(np.float32))
        np.testing.assert_array_equal(out["frame"], np.repeat((ids % 251).astype(np.uint8)[:, None], 16, 1))
        np.testing.assert_array_equal(prio, ids + 0.5)
        got_ids.append(ids.copy())
        got += n
    for t in threads:
        t.join(timeout=30)
    assert not errors and got == total
    ids = np.concatenate(got_ids)
    assert sorted(ids.tolist()) == list(range(total))  # exactly once
    # rows of one produce() call are contiguous and ordered
    assert np.all(np.diff(ids.reshape(-1, ROWS), axis=1) == 1) and np.all(ids.reshape(-1, ROWS)[:, 0] % ROWS == 0)
    st = ring.stats()
    assert st["produced"] == total and st["drained"] == total
    # a full ring with a bounded wait reports JH_ERR_STATE and writes nothing
    small = ops.StagingRing(4, cols, with_priority=False, device=None)
    z = {"state": np.zeros((4, 3), np.float32), "action": np.zeros((4, 1), np.int64), "frame": np.zeros((4, 16), np.uint8)}
    small.produce(z)
    import pytest

    with pytest.raises(L.JhError):
        small.produce({k: v[:1] for k, v in z.items()}, timeout_ms=50)
    assert small.stats()["produced"] == 4


def test_control_env_matches_oracle_bit_for_bit(lib):
    """jh_control_* (synthetic continuous control at config.ppo.mujoco shapes, host code) vs oracle ControlOracle."""
    from jorldy_amd import ops
    from oracle.jorldy_oracle import ControlOracle

    env, orc = ops.ControlVec(4, 11, 3, seed=3), ControlOracle(4, 11, 3, seed=3)
    rng = np.random.RandomState(1)
    n_done = 0
    for t in range(1200):
        assert np.array_equal(env.obs(), orc.obs())
        a = np.tanh(rng.randn(4, 3) * 3).astype(np.float32)
        n1, r1, d1 = env.step(a)
        n2, r2, d2 = orc.step(a)
        assert np.array_equal(n1, n2) and np.array_equal(r1, r2) and np.array_equal(d1.astype(bool), d2), t
        n_done += int(d2.sum())
    assert n_done >= 4  # the 1000-step cap at least


def test_epoch_shuffles_are_numpys_own_draws_bit_for_bit():
    """jh_np_legacy_shuffles / np_rng.epoch_shuffles vs `idxs = np.arange(M); for e: np.random.shuffle(idxs)` (core/agent/ppo.py:116-118):
    identical index lists AND an identical generator afterwards; np_rng.Predraw: lists drawn ahead on a copy of the state are
    installed only when nobody touched np.random in between, and leave the stream exactly where the reference's calls would."""
    from jorldy_amd import np_rng

    for seed in range(6):
        for M in (1, 2, 7, 64, 1000, 1024, 4096):
            np.random.seed(seed)
            idx, ref = np.arange(M), []
            for _ in range(3):
                np.random.shuffle(idx)
                ref.append(idx.copy())
            ref = np.concatenate(ref)
            tail_ref = np.random.randint(1 << 30, size=5)
            np.random.seed(seed)
            out = np.empty(3 * M, np.int64)
            np_rng.epoch_shuffles(M, 3, out)
            assert np.array_equal(out, ref) and np.array_equal(np.random.randint(1 << 30, size=5), tail_ref), (seed, M)
            # drawn ahead, nobody in between: same lists, same stream
            np.random.seed(seed)
            p, out2 = np_rng.Predraw(), np.empty(3 * M, np.int64)
            assert p.draw(M, 3, out2)
            assert p.commit(M, 3) and np.array_equal(out2, ref) and np.array_equal(np.random.randint(1 << 30, size=5), tail_ref)
            assert not p.commit(M, 3)  # one-shot
            # somebody drew from np.random in between: the lists are refused and the global stream is untouched by the attempt
            np.random.seed(seed)
            p.draw(M, 3, out2)
            a = np.random.rand()
            assert not p.commit(M, 3)
            b = np.random.rand()
            np.random.seed(seed)
            assert (np.random.rand(), np.random.rand()) == (a, b)
            # another shape than the one drawn for: refused
            np.random.seed(seed)
            p.draw(M, 3, out2)
            assert not p.commit(M + 1, 3)


def test_profile_name_mapping_follows_the_engine_tags():
    """profiles/ are only reproducible if a launch name finds its kernel symbol: bench.py's matcher and tools/rocprof_tgemm_names.py
    both derive the call-site <-> template-tag map; it has to be the one in csrc/jh_tgemm.h (JH_TGEMM_TAGS)."""
    import importlib.util

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    names = load("tools/rocprof_tgemm_names.py", "rocprof_tgemm_names")
    tags = names.tags()
    src = open(os.path.join(ROOT, "bench.py")).read()
    listed = eval(src[src.index("TGEMM_TAGS = [") + len("TGEMM_TAGS = "):src.index("]", src.index("TGEMM_TAGS = [")) + 1])
    assert listed == [tags[i] for i in range(len(tags))], "bench.py TGEMM_TAGS out of step with jh_tgemm.h"
    assert names.rename("void (anonymous namespace)::jh_tgemm_kernel<2, 1, 9>(TGemmBatch)", tags) == "jh_tgemm_stream1_bwd[64x32]"
    assert names.rename("void (anonymous namespace)::jh_tgemm_dma_kernel<16>(TGemmBatch)", tags) == "jh_tgemm_ppo_bwd[dma]"
    assert names.rename("jh_gae_kernel(int, int)", tags) == "jh_gae_kernel(int, int)"
