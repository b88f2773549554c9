"""Margin ledger of the GPU parity suite: every tolerance assert records achieved / allowed.

`leq(achieved, allowed, what)` asserts achieved <= allowed AND remembers the ratio under the running test's node id;
`np.testing.assert_allclose` is wrapped (tests/conftest.py) so that its calls land here too, with
achieved / allowed = max |a - b| / (atol + rtol |b|).  At session end the ledger is written to
`$JH_MARGINS_OUT` (default gpurun_out/margins_<host>_<pid>.json): per test the worst ratio and what it was, so that a
tolerance which passes by a hair on one box is visible BEFORE it fails on another (round 3's red driver run was a 3e-5
assert passing at 0.9 on the builder's boxes).  tools/margins_merge.py folds several boxes' files into profiles/.
"""
import json
import os
import socket

import numpy as np

_LEDGER = {}        # node id -> list of (ratio, what, achieved, allowed)
_CURRENT = [None]


def _node():
    return _CURRENT[0] or os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def record(achieved, allowed, what=""):
    achieved, allowed = float(achieved), float(allowed)
    ratio = achieved / allowed if allowed > 0 else (0.0 if achieved == 0 else float("inf"))
    _LEDGER.setdefault(_node(), []).append((ratio, str(what), achieved, allowed))
    return ratio


def leq(achieved, allowed, what=""):
    """assert achieved <= allowed, recorded."""
    record(achieved, allowed, what)
    assert float(achieved) <= float(allowed), f"{what}: {float(achieved):.4e} > allowed {float(allowed):.4e}"


def lt(achieved, allowed, what=""):
    record(achieved, allowed, what)
    assert float(achieved) < float(allowed), f"{what}: {float(achieved):.4e} >= allowed {float(allowed):.4e}"


def close(actual, desired, rtol=1e-7, atol=0.0, what=""):
    """np.testing.assert_allclose with the margin recorded under `what`."""
    return _wrapped_allclose(actual, desired, rtol=rtol, atol=atol, err_msg=what)


_np_assert_allclose = np.testing.assert_allclose


def _wrapped_allclose(actual, desired, rtol=1e-7, atol=0, equal_nan=True, err_msg="", verbose=True, **kw):
    a = np.asarray(actual, dtype=np.float64) if not isinstance(actual, np.ndarray) or actual.dtype != object else None
    if a is not None and (rtol > 0 or atol > 0):
        try:
            d = np.asarray(desired, dtype=np.float64)
            a2, d2 = np.broadcast_arrays(a, d)
            fin = np.isfinite(a2) & np.isfinite(d2)
            if fin.any():
                lim = atol + rtol * np.abs(d2[fin])
                diff = np.abs(a2[fin] - d2[fin])
                ok = lim > 0
                if ok.any():
                    r = diff[ok] / lim[ok]
                    i = int(np.argmax(r))
                    record(diff[ok][i], lim[ok][i], err_msg or f"allclose rtol={rtol:g} atol={atol:g}")
        except (ValueError, TypeError):
            pass
    return _np_assert_allclose(actual, desired, rtol=rtol, atol=atol, equal_nan=equal_nan, err_msg=err_msg, verbose=verbose, **kw)


def _wrap_torch_assert_close():
    import torch

    orig = torch.testing.assert_close

    def wrapped(actual, expected, *args, rtol=None, atol=None, **kw):
        try:
            if torch.is_tensor(actual) and torch.is_tensor(expected) and actual.shape == expected.shape and actual.numel() > 0 and actual.is_floating_point():
                r, a = rtol, atol
                if r is None and a is None:  # torch's defaults for float32 / float64
                    r, a = (1.3e-6, 1e-5) if actual.dtype == torch.float32 else (1e-7, 1e-7)
                lim = a + r * expected.detach().double().abs()
                diff = (actual.detach().double() - expected.detach().double()).abs()
                ok = lim > 0
                if bool(ok.any()):
                    ratio = torch.where(ok, diff / torch.where(ok, lim, torch.ones_like(lim)), torch.zeros_like(diff)).reshape(-1)
                    i = int(torch.argmax(ratio))
                    record(float(diff.reshape(-1)[i]), float(lim.reshape(-1)[i]), kw.get("msg") if isinstance(kw.get("msg"), str) else f"assert_close rtol={r:g} atol={a:g}")
        except Exception:
            pass
        return orig(actual, expected, *args, rtol=rtol, atol=atol, **kw)

    torch.testing.assert_close = wrapped


def install():
    np.testing.assert_allclose = _wrapped_allclose
    try:
        _wrap_torch_assert_close()
    except ImportError:
        pass


def set_current(nodeid):
    _CURRENT[0] = nodeid


def summary():
    out = {}
    for node, rows in _LEDGER.items():
        worst = max(rows, key=lambda r: r[0])
        out[node] = {"n_asserts": len(rows), "worst_ratio": round(worst[0], 4), "what": worst[1][:160], "achieved": worst[2], "allowed": worst[3]}
    return out


def dump(path=None):
    if not _LEDGER:
        return None
    s = summary()
    gpu = None
    try:
        import torch

        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(0)
            gpu = {"name": p.name, "cus": p.multi_processor_count, "uuid": str(getattr(p, "uuid", ""))}
    except Exception:
        pass
    doc = {"host": socket.gethostname(), "gpu": gpu, "n_tests": len(s), "max_ratio": max(v["worst_ratio"] for v in s.values()),
           "over_half": sorted(k for k, v in s.items() if v["worst_ratio"] > 0.5), "tests": s}
    if path is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.environ.get("JH_MARGINS_OUT") or os.path.join(d, f"margins_{socket.gethostname()}_{os.getpid()}.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    return path
