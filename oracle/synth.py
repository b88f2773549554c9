"""Deterministic recipes for BIG fixture inputs  --  TEST INFRASTRUCTURE, NOT PRODUCT.

The fixtures at BASELINE widths (hidden 512, (4,84,84) frames, B=32; PPO minibatch 2048) would
need tens of MB of initial weights and frames if they were stored.  Instead both sides regenerate
them from a seed with numpy's legacy `RandomState` (bit-stable across machines and numpy
versions): `oracle/gen_golden.py` writes the recipe values INTO the unmodified reference agent
before it runs `learn()`, and the `-m gpu` tests write the same values into the drop-in agent.
Only the reference's OUTPUTS (losses, indices, trees, head gradients, strided samples of the
parameter gradients / updated weights) are stored under tests/golden/.

Only `oracle/gen_golden.py` and `tests/` import this module.
"""
import zlib

import numpy as np


def _fan_in(name, shape):
    leaf = name.split(".")[-1]
    if leaf.startswith(("mu_w", "sig_w")):  # NoisyNet weights are stored [in][out] (network/utils.py:84-107)
        return int(shape[0])
    if len(shape) >= 2:
        return int(np.prod(shape[1:]))
    return 1


def recipe_tensor(name, shape, seed):
    """Value of parameter `name` (a state_dict key) with `shape`: order-independent (the stream is keyed
    by the name), magnitudes of a sane initialisation so that activations stay alive."""
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    fan = _fan_in(name, shape)
    if leaf.startswith("sig_"):
        fan_w = fan if leaf.startswith("sig_w") else max(shape[0], 1)
        v = rs.uniform(0.5, 1.5, size=shape) * (0.5 / np.sqrt(max(fan_w, 1)))
    elif len(shape) >= 2:
        v = rs.uniform(-1.0, 1.0, size=shape) * np.sqrt(6.0 / fan)
    else:
        v = rs.uniform(-0.1, 0.1, size=shape)
    return np.ascontiguousarray(v, dtype=np.float32)


def recipe_state_dict(named_shapes, seed):
    """{name: shape} (or iterable of pairs) -> {name: float32 array}."""
    items = named_shapes.items() if hasattr(named_shapes, "items") else named_shapes
    return {k: recipe_tensor(k, tuple(s), seed) for k, s in items}


def ppo_recipe(named_shapes, seed):
    """PPO policy-value net: the head matrices scaled down (like the reference's small policy gain) so the
    probability ratios of a perturbed policy stay around the clip range."""
    sd = recipe_state_dict(named_shapes, seed)
    for k in sd:
        if k.split(".")[0] in ("pi", "mu", "log_std", "v") and sd[k].ndim == 2:
            sd[k] = np.ascontiguousarray(sd[k] * np.float32(0.3), dtype=np.float32)
    return sd


def thin(a, limit=8192, stride=61):
    """Strided sample of a big array (flat[::stride]); small arrays are kept whole.  `stride` is prime so
    the sample walks through every row/column phase of the usual power-of-two shapes."""
    a = np.asarray(a)
    return a if a.size <= limit else np.ascontiguousarray(a.reshape(-1)[::stride])


def raw_transition(rng, S, A, with_q=False):
    """One synthetic env step in the draw order gen_golden._fill has always used.  S int -> float32 vector
    observation, tuple -> uint8 frames as the Atari wrapper hands them over (core/env/atari.py:147-149)."""
    if isinstance(S, (int, np.integer)):
        draw = lambda: rng.randn(1, S).astype(np.float32)
    else:
        draw = lambda: rng.randint(0, 256, size=(1,) + tuple(S)).astype(np.uint8)
    t = {
        "state": draw(),
        "action": rng.randint(0, A, size=(1, 1)),
        "reward": rng.choice([-1.0, 0.0, 1.0, 0.5], size=(1, 1)),
        "next_state": draw(),
        "done": np.asarray([[rng.rand() < 0.1]]),
    }
    if with_q:
        t["q"] = rng.randn(1, 1).astype(np.float32)
    return t


def ppo_rollout(rng, M, S, A, cont, clamp_every=17):
    """M synthetic PPO transitions in the draw order gen_golden.gen_ppo has always used."""
    trs = []
    for i in range(M):
        t = {
            "state": rng.randn(1, S).astype(np.float32),
            "next_state": rng.randn(1, S).astype(np.float32),
            "reward": rng.randn(1, 1) * 0.5,
            "done": np.asarray([[rng.rand() < 0.05]]),
        }
        if cont:
            t["action"] = np.tanh(rng.randn(1, A)).astype(np.float32)
            if clamp_every and i % clamp_every == 0:
                t["action"][0, 0] = 1.0  # hits the atanh clamp
        else:
            t["action"] = rng.randint(0, A, size=(1, 1))
        trs.append(t)
    return trs


def ppo_image_rollout(rng, M, S, A):
    """M synthetic PPO transitions on uint8 frames of shape S = (C, H, W) (core/env/atari.py:147-149 hands frames over as uint8), discrete actions."""
    trs = []
    for _ in range(M):
        trs.append({
            "state": rng.randint(0, 256, size=(1,) + tuple(S)).astype(np.uint8),
            "next_state": rng.randint(0, 256, size=(1,) + tuple(S)).astype(np.uint8),
            "reward": rng.randn(1, 1) * 0.5,
            "done": np.asarray([[rng.rand() < 0.05]]),
            "action": rng.randint(0, A, size=(1, 1)),
        })
    return trs


def row_checksum(a):
    """int64 byte sum + a position-weighted sum per leading row: pins regenerated frames to the fixture."""
    b = np.ascontiguousarray(a).reshape(a.shape[0], -1).view(np.uint8).astype(np.int64)
    w = (np.arange(b.shape[1], dtype=np.int64) % 251) + 1
    return np.stack([b.sum(1), (b * w).sum(1)], 1)
