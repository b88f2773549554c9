"""The suite's own instruments, checked on the CPU: the margin ledger (tests/margins.py), its merge tool (tools/margins_merge.py), the
float64 criterion (tests/fp64_truth.py) -- an optimizer step with a wrong bias correction must FAIL it, a correct one pass -- and the
Hopper CPU reference bench.py reports beside its `hopper` leg."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import fp64_truth as T
import margins

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_margin_ledger_records_ratios_and_the_merge_flags_what_matters(tmp_path):
    margins.leq(2.0, 8.0, "a quarter")
    np.testing.assert_allclose(np.array([1.0, 2.0]), np.array([1.0, 2.0 + 3e-6]), rtol=0, atol=1e-5)  # wrapped by conftest: 0.3
    with pytest.raises(AssertionError):
        margins.leq(3.0, 2.0, "over")
    node = margins._node()
    ratios = sorted(round(r[0], 3) for r in margins._LEDGER[node])
    assert ratios == [0.25, 0.3, 1.5]
    margins._LEDGER.pop(node, None)  # a deliberate 1.5 must not show up in a dumped ledger
    # two boxes' ledgers: one test identical on both and below 0.5, one above 0.5 and different between the boxes
    boxes = []
    for i, hair in enumerate((0.62, 0.91)):
        p = tmp_path / f"box{i}.json"
        json.dump({"host": f"h{i}", "gpu": {"uuid": str(i)}, "n_tests": 2, "max_ratio": hair,
                   "tests": {"t::calm": {"worst_ratio": 0.2, "what": "x"}, "t::hair": {"worst_ratio": hair, "what": "y"}}}, open(p, "w"))
        boxes.append(str(p))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "margins_merge.py")] + boxes, capture_output=True, text=True, check=True)
    m = json.loads(out.stdout)
    assert m["max_ratio"] == 0.91 and [r["test"] for r in m["over_half"]] == ["t::hair"] and m["over_half_and_box_dependent"] == ["t::hair"]
    calm = [r for r in m["tests"] if r["test"] == "t::calm"][0]
    assert calm["same_on_every_box"] and calm["worst_ratio"] == 0.2


def _tiny():
    torch.manual_seed(0)
    m64 = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3)).double()
    T.round_to_fp32_(m64)
    return m64, T.as32(m64)


def test_float64_criterion_passes_fp32_arithmetic_and_rejects_a_wrong_bias_correction():
    """check_first_step_from_our_gradient: float64 Adam fed "our" gradient must land on "our" weights.  "Ours" here is an fp32 Adam
    step done by hand -- once correctly (passes), once with the bias corrections of step 2 instead of step 1 (a quarter of a step), and
    once with the update simply 5 % too long (what VERDICT r3 weak #12 said no test would notice) -- both must fail."""
    m64, m32 = _tiny()
    x = torch.randn(16, 6)
    loss = (m32(x) ** 2).mean()
    loss.backward()
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    w0 = {k: v.detach().clone() for k, v in m32.state_dict().items()}
    g = {k: p.grad.detach().clone() for k, p in m32.named_parameters()}

    def adam_fp32(step_for_bc):
        out = {}
        for k in w0:
            m = (1 - b1) * g[k]
            v = (1 - b2) * g[k] * g[k]
            bc1, bc2 = 1 - b1 ** step_for_bc, 1 - b2 ** step_for_bc
            out[k] = w0[k] - (lr / bc1) * (m / (v.sqrt() / (bc2 ** 0.5) + eps))
        return out

    make = lambda ps: torch.optim.Adam(ps, lr=lr, betas=(b1, b2), eps=eps)
    T.check_first_step_from_our_gradient(w0, g, adam_fp32(1), make, lr, "correct step")
    with pytest.raises(AssertionError):
        T.check_first_step_from_our_gradient(w0, g, adam_fp32(2), make, lr, "bias correction of the wrong step")
    good = adam_fp32(1)
    with pytest.raises(AssertionError):
        T.check_first_step_from_our_gradient(w0, g, {k: w0[k] + 1.05 * (good[k] - w0[k]) for k in w0}, make, lr, "a step 5 % too long")


def test_vs_exact_allows_what_fp32_itself_cannot_do_better_and_nothing_more():
    exact = torch.linspace(-1, 1, 101, dtype=torch.float64)
    ref32 = exact + 3e-5          # the fp32 comparator is itself 3e-5 away: 6e-5 allowed
    T.vs_exact(exact + 5e-5, exact, ref32, 1e-5, "within twice the comparator's own error")
    with pytest.raises(AssertionError):
        T.vs_exact(exact + 7e-5, exact, ref32, 1e-5, "beyond it")
    with pytest.raises(AssertionError):
        T.vs_exact(exact + 2e-5, exact, exact.clone(), 1e-5, "comparator exact: only the tolerance is left")


def test_hopper_cpu_reference_runs_the_ports_learner_at_the_legs_shapes():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    threads = torch.get_num_threads()
    try:
        r = b.hopper_cpu_reference(W=2, T=64, B=32, epochs=2)
    finally:
        torch.set_num_threads(threads)
    assert r["kind"] == "port" and r["unit"] == "transitions/s" and r["value"] > 0 and abs(r["value"] * r["s_per_learn"] - 128) < 1e-6 * 128
    assert "2 epochs x 4 minibatches of 32" in r["sample"]
