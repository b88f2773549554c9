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
    assert lib.jh_abi_version() == 1


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
