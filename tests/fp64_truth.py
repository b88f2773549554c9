"""Float64 ground truth for the network / optimizer parity tests.

Criterion (VERDICT r3 "fix the criterion, not the number"): the comparator is NOT another GPU library (round 3's
MIOpen-vs-ours assert measured vendor algorithm choices that differ per box) but the exact result -- the same torch modules
evaluated on the CPU in float64 -- with torch-CPU float32 (the oracle's arithmetic) evaluated beside it:

  forward values / gradients / optimizer moments (well conditioned):
        |ours - exact| <= max(tol, 2 x |torch_cpu_fp32 - exact|)            per tensor, relative to max |exact|
  optimizer step (weights, moments): checked as ARITHMETIC, not through the gradient.  A weight after a step is ill conditioned in
  the gradient (centered RMSprop's first step moves a weight by lr g / (0.218 |g| + eps): a gradient off by 2e-8 moves it by
  0.13 lr -- that, compared against MIOpen, was round 3's red assert), so the step is decomposed: (1) OUR raw gradient is checked
  against float64 as above; (2) float64 torch.optim is fed OUR raw gradient (clip_grad_norm_ + step) and our stepped weights / moments
  must equal ITS result up to fp32 arithmetic:
        |w_ours - w_f64(g_ours)|_i <= 2^-22 |w_i| + 1e-4 |dw_i| + 1e-6 lr       per ELEMENT (one rounding of w + the update's own rounding)
        |m_ours - m_f64(g_ours)|   <= 1e-5 max |m|                              per tensor (the clip coefficient carries the fp32 norm's error)
  Steps are teacher-forced (every implementation starts each step from the SAME fp32-representable weights and optimizer state,
  taken from the float64 trajectory), so one step's check is not polluted by the previous step's differences.

Every comparison lands in the margin ledger (tests/margins.py)."""
import copy

import torch

import margins


def round_to_fp32_(module64):
    with torch.no_grad():
        for p in module64.parameters():
            p.copy_(p.float().double())
    return module64


def as32(module64):
    m = copy.deepcopy(module64).float()
    return m


def vs_exact(ours, exact, ref32, tol, what):
    """Well-conditioned quantity: per-tensor criterion above.  -> (e_ours, e_ref32), both relative to max |exact|."""
    ours = torch.as_tensor(ours).detach().double().cpu()
    exact = exact.detach().double().cpu()
    scale = float(exact.abs().max()) + 1e-30
    e_ours = float((ours.reshape(exact.shape) - exact).abs().max()) / scale
    e_ref = float((ref32.detach().double().cpu().reshape(exact.shape) - exact).abs().max()) / scale
    margins.leq(e_ours, max(tol, 2.0 * e_ref), f"{what} |ours - fp64| / max|fp64| (torch-cpu-fp32: {e_ref:.2e})")
    return e_ours, e_ref


class OptimTruth:
    """One optimizer configuration run three ways: float64 truth (torch.optim on the CPU), torch-CPU float32, and whatever `native`
    callbacks drive.  `named64` / `named32`: OrderedDict name -> parameter (same names as the native state_dict)."""

    def __init__(self, mod64, mod32, make_opt, lr, state_keys):
        self.mod64, self.mod32 = mod64, mod32
        self.opt64, self.opt32 = make_opt(mod64.parameters()), make_opt(mod32.parameters())
        self.lr, self.state_keys = lr, state_keys  # state_keys = (first moment name | None, second moment name)

    def teacher_force(self):
        """Round the float64 trajectory to float32 and give that state to the float32 run.  -> (params, m, v) dicts of fp32 tensors
        for the native implementation (m / v None before the first step)."""
        round_to_fp32_(self.mod64)
        with torch.no_grad():
            for p32, p64 in zip(self.mod32.parameters(), self.mod64.parameters()):
                p32.copy_(p64.float())
        names = [k for k, _ in self.mod64.named_parameters()]
        st_m, st_v = {}, {}
        for name, p64, p32 in zip(names, self.mod64.parameters(), self.mod32.parameters()):
            s64 = self.opt64.state.get(p64)
            if not s64:
                continue
            s32 = self.opt32.state[p32]
            for k, v in s64.items():
                if torch.is_tensor(v) and v.dim() > 0:
                    v.copy_(v.float().double())
                    s32[k].copy_(v.float())
                elif torch.is_tensor(v):
                    s32[k].copy_(v)
            if self.state_keys[0] is not None:
                st_m[name] = s64[self.state_keys[0]].float()
            st_v[name] = s64[self.state_keys[1]].float()
        params = {k: v.detach().float() for k, v in self.mod64.state_dict().items()}
        return params, (st_m or None), (st_v or None)

    def _clip(self, mod, max_norm):
        if max_norm:
            torch.nn.utils.clip_grad_norm_(mod.parameters(), max_norm)

    def step(self, max_norm, ours_raw_grads, ours_params, ours_m, ours_v, tag):
        """p.grad of mod64 / mod32 hold this step's RAW exact gradients; ours_raw_grads {name: tensor} is OUR raw gradient (already
        checked against them by the caller).  Steps the float64 / float32 trajectories with their own gradients, and a float64 copy
        with OUR gradient -- what `ours_*` (dicts name -> tensor AFTER the native [clip +] step) must reproduce.  -> worst ratio."""
        mod, opt = copy.deepcopy((self.mod64, self.opt64))
        with torch.no_grad():
            for (name, p) in mod.named_parameters():
                p.grad = ours_raw_grads[name].detach().double().cpu().reshape(p.shape).clone()
        w_before = [p.detach().clone() for p in mod.parameters()]
        self._clip(mod, max_norm)
        opt.step()
        for m_, o_ in ((self.mod64, self.opt64), (self.mod32, self.opt32)):
            self._clip(m_, max_norm)
            o_.step()
        worst = 0.0
        for i, (name, p) in enumerate(mod.named_parameters()):
            w = p.detach()
            allowed = 2.0 ** -22 * w.abs() + 1e-4 * (w - w_before[i]).abs() + 1e-6 * self.lr
            err = (ours_params[name].detach().double().cpu().reshape(w.shape) - w).abs()
            ratio = (err / allowed).reshape(-1)
            j = int(torch.argmax(ratio))
            margins.leq(float(err.reshape(-1)[j]), float(allowed.reshape(-1)[j]), f"{tag} weight {name}[{j}] vs float64 step of OUR gradient (max err {float(err.max()):.2e})")
            worst = max(worst, float(ratio[j]))
            st = opt.state[p]
            for key, ours in ((self.state_keys[0], ours_m), (self.state_keys[1], ours_v)):
                if key is not None and ours is not None:
                    exact = st[key].detach()
                    scale = float(exact.abs().max()) + 1e-30
                    e = float((ours[name].detach().double().cpu().reshape(exact.shape) - exact).abs().max()) / scale
                    margins.leq(e, 1e-5, f"{tag} {key} {name} vs float64 step of OUR gradient, / max |{key}|")
        return worst


def check_first_step_from_our_gradient(w0, g_clipped, w1, make_opt, lr, tag):
    """ONE optimizer step as arithmetic (the statistical "99.5 % within 5 % of a step" criteria of the fixture tests cannot see a bias
    correction or eps that is off by a few per cent): float64 torch.optim, started from the fixture's initial weights w0 and zero state, is
    fed OUR gradient as it sits in the bucket after the step (i.e. already clipped, like p.grad in the reference) and must land where our
    weights landed, per element: |w_ours - w_f64| <= 2^-22 |w| + 1e-4 |dw| + 1e-6 lr.  Dicts name -> tensor / array (any device)."""
    t64 = lambda a: torch.as_tensor(a).detach().double().cpu()
    params = {k: torch.nn.Parameter(t64(v).clone()) for k, v in w0.items()}
    opt = make_opt(list(params.values()))
    before = {k: p.detach().clone() for k, p in params.items()}
    for k, p in params.items():
        p.grad = t64(g_clipped[k]).reshape(p.shape).clone()
    opt.step()
    worst = 0.0
    for k, p in params.items():
        w = p.detach()
        allowed = 2.0 ** -22 * w.abs() + 1e-4 * (w - before[k]).abs() + 1e-6 * lr
        err = (t64(w1[k]).reshape(w.shape) - w).abs()
        ratio = (err / allowed).reshape(-1)
        j = int(torch.argmax(ratio))
        margins.leq(float(err.reshape(-1)[j]), float(allowed.reshape(-1)[j]), f"{tag} weight {k}[{j}] vs float64 step of OUR gradient from the fixture's w0 (max err {float(err.max()):.2e})")
        worst = max(worst, float(ratio[j]))
    return worst


def ppo_head_grads_float64(cont, heads, action, adv, ret, v_old, lp_old, eps, vf, ent):
    """d(loss)/d(raw heads) of ONE PPO minibatch in float64: core/agent/ppo.py:125-165 (+ the policy modules of
    core/network/policy_value.py:38-57) restated with torch autograd on the CPU -- the comparator of the loss-kernel tests (test
    infrastructure, not the product).  heads: {"logits" | "mu_raw", "log_std_raw", "v"} arrays of the minibatch's rows; the other
    arguments are the rows' upstream quantities.  -> {name: float64 gradient array}."""
    import numpy as np

    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    adv, ret, v_old, lp_old, action = t(adv), t(ret), t(v_old), t(lp_old), t(action)
    v = t(heads["v"]).reshape(-1, 1).requires_grad_(True)
    if cont:
        mu_raw, ls_raw = t(heads["mu_raw"]).requires_grad_(True), t(heads["log_std_raw"]).requires_grad_(True)
        m = torch.distributions.Normal(torch.clamp(mu_raw, -5.0, 5.0), torch.tanh(ls_raw).exp())
        # the action clamp is part of the INPUT preparation and happens in float32 in the reference (reinforce.py / ppo.py:85-88 on fp32
        # tensors: the bound 1 - 1e-7 rounds to 1 - 2^-23 there); everything from the clamped action on is float64
        a_cl = torch.clamp(action.float(), -1 + 1e-7, 1 - 1e-7).double()
        log_prob = m.log_prob(torch.atanh(a_cl))
        leaves = {"mu_raw": mu_raw, "log_std_raw": ls_raw, "v": v}
    else:
        logits = t(heads["logits"]).requires_grad_(True)
        m = torch.distributions.Categorical(torch.exp(torch.log_softmax(logits, dim=-1)))
        log_prob = m.log_prob(action.reshape(-1).long()).unsqueeze(-1)
        leaves = {"logits": logits, "v": v}
    ratio = (log_prob - lp_old).sum(1, keepdim=True).exp()
    actor = -torch.min(ratio * adv, torch.clamp(ratio, 1 - eps, 1 + eps) * adv).mean()
    v_clip = v_old + torch.clamp(v - v_old, -eps, eps)
    critic = torch.max(torch.nn.functional.mse_loss(v, ret), torch.nn.functional.mse_loss(v_clip, ret))
    loss = actor + vf * critic + ent * (-m.entropy().mean())
    loss.backward()
    out = {k: x.grad.numpy() for k, x in leaves.items()}
    out["ratio"] = ratio.detach().numpy()
    return out


def grad_vs_exact(ours, exact, ref32, tol, what, rows=None):
    """|ours - exact| <= max(tol, 2 x |reference fp32 - exact|), relative to max |exact| (ref32 None: tol alone); rows: boolean row mask."""
    import numpy as np

    exact = np.asarray(exact, dtype=np.float64)
    scale = float(np.abs(exact).max()) + 1e-30
    pick = (lambda a: np.asarray(a, dtype=np.float64).reshape(exact.shape)[rows]) if rows is not None else (lambda a: np.asarray(a, dtype=np.float64).reshape(exact.shape))
    ex = pick(exact)
    e_ours = float(np.abs(pick(ours) - ex).max()) / scale
    e_ref = float(np.abs(pick(ref32) - ex).max()) / scale if ref32 is not None else 0.0
    margins.leq(e_ours, max(tol, 2.0 * e_ref), f"{what} |ours - fp64| / max|fp64| (reference fp32: {e_ref:.2e})")
    return e_ours, e_ref
