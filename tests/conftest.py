import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)

import margins  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    margins.install()  # np.testing.assert_allclose records achieved / allowed (tests/margins.py)


@pytest.fixture(autouse=True)
def _margin_scope(request):
    margins.set_current(request.node.nodeid)
    yield
    margins.set_current(None)


def pytest_sessionfinish(session, exitstatus):
    """GPU runs leave the margin ledger under gpurun_out/ (scratch; tools/margins_merge.py folds boxes into profiles/)."""
    try:
        import torch

        if torch.cuda.is_available():
            margins.dump()
    except Exception as e:  # the ledger must never turn a green run red
        print(f"[margins] not written: {e}")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
