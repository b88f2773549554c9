"""Two RCCL ranks, rank r on GPU r: the library's own communicator (jh_comm_create from a torch-broadcast id, the
all-reduces of the gradient bucket and of the critic sums captured into the learn() graph) before the N-GPU bench meets it.  Needs two
GPUs: skipped on the one-GPU build boxes, so these tests have NEVER run on hardware when the driver's node sees them first -- which is why
they sit in the file pytest collects last: whatever a first contact with RCCL brings, `-x` has nothing behind it to hide.
The checks themselves are the gloo two-ranks-on-one-GPU tests' (tests/test_dp_two_ranks_gpu.py), which run everywhere."""
import numpy as np
import pytest
import torch

from tests.test_dp_two_ranks_gpu import _check_ppo_two_ranks, _run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two RCCL ranks need two GPUs (the driver's 8-GPU node; the builder's box has one)")
@pytest.mark.parametrize("inject", ["", "create"])
def test_ppo_native_two_ranks_over_rccl_equal_one_learner(tmp_path, monkeypatch, inject):
    """The same equality with rank r on GPU r over RCCL: the library's own communicator (jh_comm_create from a torch-broadcast id,
    the all-reduces of the gradient bucket and of the critic sums captured into the learn() graph) before the bench meets it.
    inject = create: jh_comm_create fails on every rank -> all ranks fall back to torch.distributed's collectives TOGETHER."""
    _check_ppo_two_ranks(_run_ranks("ppo", tmp_path, backend="nccl", extra_env={"JH_COMM_INJECT": inject} if inject else None), monkeypatch)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two RCCL ranks need two GPUs")
def test_rainbow_native_two_ranks_over_rccl_identical_weights(tmp_path):
    r0, r1 = _run_ranks("rainbow", tmp_path, backend="nccl")
    assert np.array_equal(r0["params"], r1["params"]) and np.array_equal(r0["target"], r1["target"])
    assert np.all(np.isfinite(r0["losses"])) and np.all(np.isfinite(r1["losses"]))
