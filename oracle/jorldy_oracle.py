"""CPU oracle for the JORLDY RL hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain numpy restatement of the reference algorithms that `jorldy_amd`
replaces with HIP kernels.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product path
(`jorldy_amd/`) never does and fails loudly when its HIP library is missing.

Pinned against the reference itself: every function here is checked in
`tests/test_oracle_golden.py` against fixtures produced by running the
unmodified reference (`oracle/gen_golden.py` -> `tests/golden/*.npz`), because
the reference's own tests hold no numeric golden vectors.

Each function cites the reference file:line (relative to
/root/reference/jorldy/) whose behaviour it restates, quirks included.
All "fp32" math is done in np.float32 in the same operation order as the
torch expressions of the reference.
"""
import math
from collections import deque

import numpy as np

F32 = np.float32


# =============================================================================
# buffers  (core/buffer/base.py, replay_buffer.py, rollout_buffer.py, per_buffer.py)
# =============================================================================
def stack_transition(batch):
    """core/buffer/base.py:42-56 -- AoS list of per-transition dicts -> SoA dict.
    Each value has a leading dim of 1 that is dropped; list-valued (multimodal)
    keys are stacked per element."""
    out = {}
    for key in batch[0].keys():
        v0 = batch[0][key]
        if len(v0) > 1:  # multimodal: list of arrays (or any len>1 first dim)
            out[key] = [np.stack([b[key][i][0] for b in batch], axis=0) for i in range(len(v0))]
        else:
            out[key] = np.stack([b[key][0] for b in batch], axis=0)
    return out


class ReplayOracle:
    """core/buffer/replay_buffer.py:8-35 -- ring of transitions, uniform sampling
    with numpy's GLOBAL RNG (`np.random.randint(counter, size=B)`)."""

    def __init__(self, buffer_size):
        self.buffer = [None] * buffer_size
        self.buffer_index = 0
        self.buffer_size = buffer_size
        self.buffer_counter = 0

    def store(self, transitions):
        for t in transitions:
            self.buffer[self.buffer_index] = t
            self.buffer_index = (self.buffer_index + 1) % self.buffer_size
            self.buffer_counter = min(self.buffer_counter + 1, self.buffer_size)

    def sample_indices(self, batch_size):
        return np.random.randint(self.buffer_counter, size=batch_size)

    def sample(self, batch_size):
        idx = self.sample_indices(batch_size)
        return stack_transition([self.buffer[i] for i in idx])

    @property
    def size(self):
        return self.buffer_counter


class RolloutOracle:
    """core/buffer/rollout_buffer.py:6-24 -- append, stack everything, clear."""

    def __init__(self):
        self.buffer = []

    def store(self, transitions):
        self.buffer += transitions

    def sample(self):
        out = stack_transition(self.buffer)
        self.buffer.clear()
        return out

    @property
    def size(self):
        return len(self.buffer)


class PEROracle(ReplayOracle):
    """core/buffer/per_buffer.py:7-105 -- array-heap sum tree in float64.

    tree has 2N-1 nodes, leaves at [N-1, 2N-2]; every priority change is an
    incremental `+= delta` climb (per_buffer.py:42-54), so the tree's float64
    rounding history is part of the state and must be reproduced op by op."""

    def __init__(self, buffer_size, uniform_sample_prob=1e-3):
        super().__init__(buffer_size)
        self.tree_size = buffer_size * 2 - 1
        self.first_leaf_index = buffer_size - 1
        self.sum_tree = np.zeros(self.tree_size)
        self.tree_index = self.first_leaf_index
        self.max_priority = 1.0
        self.uniform_sample_prob = uniform_sample_prob

    # per_buffer.py:19-33
    def store(self, transitions):
        for t in transitions:
            self.buffer[self.buffer_index] = t
            p = t["priority"] if "priority" in t else self.max_priority
            self.add_tree_data(float(np.asarray(p).reshape(-1)[0]))
            self.buffer_counter = min(self.buffer_counter + 1, self.buffer_size)
            self.buffer_index = (self.buffer_index + 1) % self.buffer_size

    # per_buffer.py:35-40
    def add_tree_data(self, new_priority):
        self.update_priority(new_priority, self.tree_index)
        self.tree_index += 1
        if self.tree_index == self.tree_size:
            self.tree_index = self.first_leaf_index

    # per_buffer.py:42-54
    def update_priority(self, new_priority, index):
        delta = new_priority - self.sum_tree[index]
        self.sum_tree[index] = new_priority
        while index > 0:
            index = (index - 1) // 2
            self.sum_tree[index] += delta
        self.max_priority = max(self.max_priority, new_priority)

    # per_buffer.py:56-68  (tie-break: `num <= left` goes LEFT)
    def search_tree(self, num):
        index = 0
        while index < self.first_leaf_index:
            left = 2 * index + 1
            if num <= self.sum_tree[left]:
                index = left
            else:
                num -= self.sum_tree[left]
                index = left + 1
        return index

    def draw(self, batch_size):
        """The three global-RNG draws of per_buffer.py:72-81, in reference order.
        Returns (n_uniform, uniform_leaf_offsets int64[n_uniform], u float64[B-n_uniform])."""
        mask = np.random.uniform(size=batch_size) < self.uniform_sample_prob
        n_uni = int(np.sum(mask))
        uni = np.random.randint(self.buffer_counter, size=n_uni)
        u = np.random.uniform(size=batch_size - n_uni)
        return n_uni, uni, u

    # per_buffer.py:70-101
    def sample_indices(self, beta, batch_size):
        assert self.sum_tree[0] > 0.0
        n_uni, uni, u = self.draw(batch_size)
        targets = u * self.sum_tree[0]
        idx = [int(i) + self.first_leaf_index for i in uni] + [self.search_tree(t) for t in targets]
        indices = np.asarray(idx, dtype=np.int64)
        priorities = self.sum_tree[indices]
        uniform_probs = np.asarray(1.0 / self.buffer_counter)
        prioritized_probs = priorities / self.sum_tree[0]
        usp = self.uniform_sample_prob
        sample_probs = (1.0 - usp) * prioritized_probs + usp * uniform_probs
        weights = (uniform_probs / sample_probs) ** beta
        weights /= np.max(weights)
        sampled_p = np.mean(priorities)
        mean_p = self.sum_tree[0] / self.buffer_counter
        return weights, indices, sampled_p, mean_p

    def sample(self, beta, batch_size):
        weights, indices, sampled_p, mean_p = self.sample_indices(beta, batch_size)
        batch = [self.buffer[i] for i in indices - self.first_leaf_index]
        return stack_transition(batch), weights, indices, sampled_p, mean_p


# =============================================================================
# n-step assemblers (rainbow.py:294-308, multistep.py:90-104, ape_x.py:174-199)
# =============================================================================
class NStepOracle:
    """Sliding window that is NOT reset at episode ends (windows straddle resets;
    the (1-done_i) mask in the fold handles it)."""

    def __init__(self, n_step, apex=False, gamma=0.99):
        self.n_step = n_step
        self.apex = apex
        self.gamma = gamma
        self.buf = deque(maxlen=n_step + 1 if apex else n_step)

    def push(self, tr):
        self.buf.append(tr)
        if len(self.buf) < self.buf.maxlen:
            return {}
        out = {"state": self.buf[0]["state"], "action": self.buf[0]["action"]}
        if not self.apex:
            out["next_state"] = self.buf[-1]["next_state"]
            win = list(self.buf)
        else:
            out["next_state"] = self.buf[-1]["state"]  # ape_x.py:180
            win = list(self.buf)[:-1]
        for key in self.buf[0].keys():
            if key not in ("state", "action", "next_state", "q"):
                out[key] = np.stack([t[key] for t in win], axis=1)
        if self.apex:  # actor-side initial priority |G_n - q_t| (no ^alpha), ape_x.py:188-196
            target = self.buf[-1]["q"]
            for i in reversed(range(self.n_step)):
                target = self.buf[i]["reward"] + (1 - self.buf[i]["done"]) * self.gamma * target
            out["priority"] = abs(target - self.buf[0]["q"])
        return out


# =============================================================================
# PPO math  (core/agent/ppo.py)
# =============================================================================
def gae(reward, done, value, next_value, gamma, lam, n_step):
    """ppo.py:95-103.  All inputs (M,1) fp32, M = W*n_step, worker-major.
    delta = r + (1-d)*gamma*V' - V ; adv viewed (W,T) ; reverse scan that does NOT
    bootstrap across a row's last step ; ret = adv + V.  Returns (adv (W,T), ret (M,1))."""
    reward, done, value, next_value = (np.asarray(x, F32) for x in (reward, done, value, next_value))
    g, l = F32(gamma), F32(lam)
    delta = reward + (F32(1) - done) * g * next_value - value
    adv = delta.copy().reshape(-1, n_step)
    d = done.reshape(-1, n_step)
    for t in reversed(range(n_step - 1)):
        adv[:, t] += (F32(1) - d[:, t]) * g * l * adv[:, t + 1]
    ret = adv.reshape(-1, 1) + value
    return adv, ret


def standardize_rows(adv):
    """ppo.py:105-108: per ROW (worker): (adv - mean) / (std_unbiased + 1e-7)."""
    adv = np.asarray(adv, F32)
    mean = adv.mean(axis=1, keepdims=True, dtype=F32)
    std = adv.std(axis=1, keepdims=True, ddof=1, dtype=F32)
    return ((adv - mean) / (std + F32(1e-7))).astype(F32)


def _log_softmax(z):
    z = np.asarray(z, F32)
    m = z.max(axis=-1, keepdims=True)
    s = z - m
    return (s - np.log(np.exp(s).sum(axis=-1, keepdims=True, dtype=F32))).astype(F32)


_EPS32 = F32(np.finfo(np.float32).eps)


def ppo_loss_discrete(logits, value_pred, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef):
    """ppo.py:131-165 for the discrete head (policy_value.py:19-22 returns
    pi = exp(log_softmax(logits)); Categorical(probs=pi) re-normalises and clamps
    to [eps, 1-eps] before the log).  Inputs (B,A),(B,1)...  Returns dict with the
    scalar losses, the reported extrema and d(loss)/d(logits), d(loss)/d(value_pred)."""
    logits = np.asarray(logits, F32)
    B, A = logits.shape
    v = np.asarray(value_pred, F32).reshape(B)
    a = np.asarray(action).reshape(B).astype(np.int64)
    adv = np.asarray(adv, F32).reshape(B)
    ret = np.asarray(ret, F32).reshape(B)
    v_old = np.asarray(value_old, F32).reshape(B)
    lp_old = np.asarray(logp_old, F32).reshape(B)
    e = F32(eps_clip)

    lsm = _log_softmax(logits)
    p = np.exp(lsm)
    s = p.sum(axis=1, keepdims=True, dtype=F32)
    pn = p / s
    inr = (pn >= _EPS32) & (pn <= F32(1) - _EPS32)
    c = np.clip(pn, _EPS32, F32(1) - _EPS32)
    lg = np.log(c)
    rows = np.arange(B)
    logp = lg[rows, a]
    ratio = np.exp(logp - lp_old)
    surr1 = ratio * adv
    rc = np.clip(ratio, F32(1) - e, F32(1) + e)
    surr2 = rc * adv
    actor = -np.minimum(surr1, surr2).mean(dtype=F32)
    v_clip = v_old + np.clip(v - v_old, -e, e)
    c1 = ((v - ret) ** 2).mean(dtype=F32)
    c2 = ((v_clip - ret) ** 2).mean(dtype=F32)
    critic = max(c1, c2)
    ent = -(pn * lg).sum(axis=1, dtype=F32)
    entropy_loss = -ent.mean(dtype=F32)
    loss = actor + F32(vf_coef) * critic + F32(ent_coef) * entropy_loss

    # ---- backward (closed form of what autograd does) -------------------------------
    invB = F32(1.0 / B)
    # d actor / d ratio : torch.min ties split 1/2-1/2; clamp passes grad inside [1-e,1+e]
    in_clip = (ratio >= F32(1) - e) & (ratio <= F32(1) + e)
    g1 = np.where(surr1 < surr2, F32(1), np.where(surr1 == surr2, F32(0.5), F32(0)))
    g2 = np.where(surr2 < surr1, F32(1), np.where(surr1 == surr2, F32(0.5), F32(0)))
    d_ratio = -invB * (g1 * adv + g2 * adv * in_clip)
    d_logp = d_ratio * ratio
    # d loss / d lg_k  and direct d loss / d pn_k (entropy = -sum pn*lg)
    g_lg = np.zeros((B, A), F32)
    g_lg[rows, a] += d_logp
    ce = F32(ent_coef) * invB  # loss += ent_coef * (-mean(ent)) = ent_coef*invB * sum pn*lg
    g_lg += ce * pn
    g_pn = ce * lg
    g_pn = g_pn + g_lg / c * inr
    g_p = g_pn / s - (g_pn * p).sum(axis=1, keepdims=True, dtype=F32) / (s * s)
    g_lsm = g_p * p
    d_logits = g_lsm - p * g_lsm.sum(axis=1, keepdims=True, dtype=F32)
    # critic = max(c1,c2) (ties split)
    w1 = F32(1) if c1 > c2 else (F32(0.5) if c1 == c2 else F32(0))
    w2 = F32(1) - w1
    in_v = ((v - v_old) >= -e) & ((v - v_old) <= e)
    d_v = F32(vf_coef) * (w1 * F32(2) * (v - ret) * invB + w2 * F32(2) * (v_clip - ret) * invB * in_v)
    return dict(
        loss=loss, actor_loss=actor, critic_loss=critic, critic_loss1=c1, critic_loss2=c2,
        entropy_loss=entropy_loss, ratio=ratio.reshape(B, 1), log_prob=logp.reshape(B, 1),
        max_ratio=ratio.max(), min_prob=np.exp(logp).min(),
        d_logits=d_logits.astype(F32), d_value=d_v.reshape(B, 1).astype(F32),
    )


_HALF_LOG_2PI = F32(0.5 * math.log(2 * math.pi))
_ATANH_HI = F32(1 - 1e-7)
_ATANH_LO = F32(-1 + 1e-7)


def normal_head(mu_raw, log_std_raw):
    """policy_value.py:52-56: mu = clamp(mu_raw, -5, 5); std = exp(tanh(log_std_raw))."""
    mu = np.clip(np.asarray(mu_raw, F32), F32(-5), F32(5))
    std = np.exp(np.tanh(np.asarray(log_std_raw, F32)))
    return mu, std


def normal_logp_of_action(mu, std, action):
    """ppo.py:85-88: z = atanh(clamp(a, -1+1e-7, 1-1e-7)); Normal(mu,std).log_prob(z)."""
    a = np.clip(np.asarray(action, F32), _ATANH_LO, _ATANH_HI)
    z = np.arctanh(a).astype(F32)
    var = std * std
    return (-((z - mu) ** 2) / (F32(2) * var) - np.log(std) - _HALF_LOG_2PI).astype(F32), z


def ppo_loss_continuous(mu_raw, log_std_raw, value_pred, action, adv, ret, value_old, logp_old, eps_clip, vf_coef, ent_coef):
    """ppo.py:125-165 continuous branch, taking the RAW head outputs (pre clamp/tanh)
    so the backward reaches the Linear layers.  logp_old is (B,A) per-dimension."""
    mu_raw = np.asarray(mu_raw, F32)
    ls_raw = np.asarray(log_std_raw, F32)
    B, A = mu_raw.shape
    v = np.asarray(value_pred, F32).reshape(B)
    adv = np.asarray(adv, F32).reshape(B)
    ret = np.asarray(ret, F32).reshape(B)
    v_old = np.asarray(value_old, F32).reshape(B)
    lp_old = np.asarray(logp_old, F32).reshape(B, A)
    e = F32(eps_clip)
    mu, std = normal_head(mu_raw, ls_raw)
    logp, z = normal_logp_of_action(mu, std, action)
    ratio = np.exp((logp - lp_old).sum(axis=1, dtype=F32))
    surr1 = ratio * adv
    surr2 = np.clip(ratio, F32(1) - e, F32(1) + e) * adv
    actor = -np.minimum(surr1, surr2).mean(dtype=F32)
    v_clip = v_old + np.clip(v - v_old, -e, e)
    c1 = ((v - ret) ** 2).mean(dtype=F32)
    c2 = ((v_clip - ret) ** 2).mean(dtype=F32)
    critic = max(c1, c2)
    ent = F32(0.5) + _HALF_LOG_2PI + np.log(std)  # (B,A); mean over ALL elements
    entropy_loss = -ent.mean(dtype=F32)
    loss = actor + F32(vf_coef) * critic + F32(ent_coef) * entropy_loss

    invB = F32(1.0 / B)
    in_clip = (ratio >= F32(1) - e) & (ratio <= F32(1) + e)
    g1 = np.where(surr1 < surr2, F32(1), np.where(surr1 == surr2, F32(0.5), F32(0)))
    g2 = np.where(surr2 < surr1, F32(1), np.where(surr1 == surr2, F32(0.5), F32(0)))
    d_ratio = -invB * (g1 * adv + g2 * adv * in_clip)
    d_logp = (d_ratio * ratio)[:, None]  # same for every action dim
    var = std * std
    d_mu = d_logp * (z - mu) / var
    d_std = d_logp * (((z - mu) ** 2) / (var * std) - F32(1) / std)
    d_std = d_std + F32(ent_coef) * (-F32(1.0 / (B * A))) / std
    d_mu_raw = d_mu * ((mu_raw >= F32(-5)) & (mu_raw <= F32(5)))
    th = np.tanh(ls_raw)
    d_ls_raw = d_std * std * (F32(1) - th * th)
    w1 = F32(1) if c1 > c2 else (F32(0.5) if c1 == c2 else F32(0))
    w2 = F32(1) - w1
    in_v = ((v - v_old) >= -e) & ((v - v_old) <= e)
    d_v = F32(vf_coef) * (w1 * F32(2) * (v - ret) * invB + w2 * F32(2) * (v_clip - ret) * invB * in_v)
    return dict(
        loss=loss, actor_loss=actor, critic_loss=critic, critic_loss1=c1, critic_loss2=c2,
        entropy_loss=entropy_loss, ratio=ratio.reshape(B, 1), log_prob=logp,
        max_ratio=ratio.max(), min_prob=np.exp(logp).min(),
        d_mu_raw=d_mu_raw.astype(F32), d_log_std_raw=d_ls_raw.astype(F32), d_value=d_v.reshape(B, 1).astype(F32),
    )


# =============================================================================
# DQN family math  (dqn.py, double.py, per.py, multistep.py, ape_x.py)
# =============================================================================
def nstep_fold(target, reward, done, gamma):
    """multistep.py:47-48 / ape_x.py:105-106: for i=n-1..0: y = r_i + (1-d_i)*gamma*y.
    reward/done (B,n,1) fp32, target (B,1)."""
    y = np.asarray(target, F32)
    reward = np.asarray(reward, F32)
    done = np.asarray(done, F32)
    for i in reversed(range(reward.shape[1])):
        y = reward[:, i] + (F32(1) - done[:, i]) * F32(gamma) * y
    return y


def dqn_loss(q_all, action, reward, done, next_q_target, gamma, next_q_online=None, weights=None, alpha=None, n_step=0):
    """One entry point for the five TD losses:
       dqn.py:128-141      y = r + (1-d) g max Qt(s')            , Huber
       double.py:28-39     a*=argmax Q(s'); y = r + Qt(s')[a*] (g(1-d)), Huber
       multistep.py:41-50  y = fold_n(max Qt(s'))                 , Huber
       per.py:54-74        double target; td=|y-q|; p=td^alpha; loss=mean(w td^2)
       ape_x.py:96-116     double target folded n times; same PER loss
    Returns loss, q (B,1), target (B,1), td, prio, d_q_all (B,A), max_Q."""
    q_all = np.asarray(q_all, F32)
    B, A = q_all.shape
    a = np.asarray(action).reshape(B).astype(np.int64)
    rows = np.arange(B)
    q = q_all[rows, a].reshape(B, 1)
    nqt = np.asarray(next_q_target, F32)
    if next_q_online is not None:
        astar = np.argmax(np.asarray(next_q_online, F32), axis=1)
        boot = nqt[rows, astar].reshape(B, 1)
    else:
        boot = nqt.max(axis=1, keepdims=True)
    reward = np.asarray(reward, F32)
    done = np.asarray(done, F32)
    if n_step:
        y = nstep_fold(boot, reward, done, gamma)
    elif next_q_online is not None:  # double.py:35-37 multiplies (gamma*(1-d)) as one factor
        y = reward + boot * (F32(gamma) * (F32(1) - done))
    else:
        y = reward + (F32(1) - done) * F32(gamma) * boot
    diff = q - y
    invB = F32(1.0 / B)
    if weights is None:  # smooth_l1_loss, beta=1, mean
        ad = np.abs(diff)
        per = np.where(ad < F32(1), F32(0.5) * diff * diff, ad - F32(0.5))
        loss = per.mean(dtype=F32)
        dq = np.where(ad < F32(1), diff, np.sign(diff)) * invB
        td = np.abs(y - q)
        prio = None
    else:
        w = np.asarray(weights, F32).reshape(B, 1)
        td = np.abs(y - q)
        prio = np.power(td, F32(alpha))
        loss = (w * td * td).mean(dtype=F32)
        dq = F32(2) * w * diff * invB
    d_q_all = np.zeros((B, A), F32)
    d_q_all[rows, a] = dq.reshape(B)
    return dict(loss=loss, q=q, target_q=y.astype(F32), td_error=td, p_j=prio, d_q=dq.astype(F32), d_q_all=d_q_all, max_Q=q.max())


# =============================================================================
# C51 / Rainbow  (c51.py:49-135, rainbow.py:154-253,285-292)
# =============================================================================
def logits2Q(logits, z, shift_max=False):
    """rainbow.py:285-292 (shift_max=False) / c51.py:125-135 (shift_max=True).
    logits (B,A,K) -> p (B,A,K), q (B,A)."""
    l = np.asarray(logits, F32)
    if shift_max:
        l = l - l.max(axis=-1, keepdims=True)
    p = np.exp(_log_softmax(l))
    q = (np.asarray(z, F32).reshape(1, 1, -1) * p).sum(axis=-1, dtype=F32)
    return p, q


def c51_project_kl(logit, action, reward, done, target_logit, v_min, v_max, K, gamma,
                   next_logit_online=None, weights=None, alpha=None, shift_max=False):
    """C51 n-step categorical projection + cross-entropy ("KL") loss.

    rainbow.py:167-239 when next_logit_online/weights are given (double-Q action
    selection, PER weights, priority = KL^alpha); c51.py:68-109 otherwise
    (target net selects its own action, reward/done are (B,1), plain mean).
    Quirks reproduced, not fixed: the PER weights enter only through their batch
    mean (shape broadcast, see below); the terminal branch is keyed on done[:,0]
    and averages the one-hot row sums (`mean` over source atoms); integral b
    drops its mass in the non-terminal branch (l==u -> both weights 0)."""
    logit = np.asarray(logit, F32)
    B, A, _ = logit.shape
    z = np.linspace(v_min, v_max, K, dtype=F32).reshape(1, K)  # torch.linspace fp32
    dz = F32((v_max - v_min) / (K - 1))
    p_logit, q_action = logits2Q(logit, z, shift_max)
    a = np.asarray(action).reshape(B).astype(np.int64)
    rows = np.arange(B)
    p_action = p_logit[rows, a]
    tp, tq = logits2Q(target_logit, z, shift_max)
    if next_logit_online is not None:
        _, nq = logits2Q(next_logit_online, z, shift_max)
        astar = np.argmax(nq, axis=-1)
    else:
        astar = np.argmax(tq, axis=-1)
    tpa = tp[rows, astar]  # (B,K)
    reward = np.asarray(reward, F32)
    done = np.asarray(done, F32)
    if reward.ndim == 3:  # (B,n,1)
        Tz = np.broadcast_to(z, (B, K)).astype(F32)
        for i in reversed(range(reward.shape[1])):
            Tz = reward[:, i] + (F32(1) - done[:, i]) * F32(gamma) * Tz
        done0 = done[:, 0, :]
    else:
        Tz = reward + (F32(1) - done) * F32(gamma) * z
        done0 = done
    b = np.clip(Tz - F32(v_min), F32(0), F32(v_max - v_min)) / dz
    l = np.floor(b).astype(np.int64)
    u = np.ceil(b).astype(np.int64)
    wl = (u.astype(F32) - b).astype(F32)
    wu = (b - l.astype(F32)).astype(F32)
    m_term = np.zeros((B, K), F32)
    m_non = np.zeros((B, K), F32)
    for j in range(K):  # sum over source atoms j in ascending order
        both = (l[:, j] == u[:, j]).astype(F32)
        np.add.at(m_term, (rows, l[:, j]), both + wl[:, j])
        np.add.at(m_term, (rows, u[:, j]), wu[:, j])
        np.add.at(m_non, (rows, l[:, j]), tpa[:, j] * wl[:, j])
        np.add.at(m_non, (rows, u[:, j]), tpa[:, j] * wu[:, j])
    m_term = m_term / F32(K)
    m = done0 * m_term + (F32(1) - done0) * m_non
    m = m / np.maximum(m.sum(axis=1, keepdims=True, dtype=F32), F32(1e-8))
    pc = np.maximum(p_action, F32(1e-8))
    KL = -(m * np.log(pc)).sum(axis=-1, dtype=F32)
    invB = F32(1.0 / B)
    if weights is not None:
        # rainbow.py:233-235: weights is (B,1) but KL is (B,), so `weights * KL` BROADCASTS to
        # (B,B) and the loss is mean(w)*mean(KL): every sample gets the batch-mean IS weight.
        w = np.asarray(weights, F32).reshape(B)
        loss = (w[:, None] * KL[None, :]).mean(dtype=F32)
        prio = np.power(KL, F32(alpha))
        w = np.full(B, w.mean(dtype=F32), F32)
    else:
        w = np.ones(B, F32)
        loss = KL.mean(dtype=F32)
        prio = None
    mt = m * (p_action >= F32(1e-8))
    dz_a = (-mt + p_action * mt.sum(axis=1, keepdims=True, dtype=F32)) * (w * invB)[:, None]
    d_logit = np.zeros_like(logit)
    d_logit[rows, a] = dz_a
    return dict(loss=loss, KL=KL, p_j=prio, target_dist=m, Tz=Tz, b=b, l=l, u=u, p_action=p_action,
                target_action=astar.reshape(B, 1), target_p_action=tpa, q_action=q_action,
                d_logit=d_logit, max_Q=q_action.max(), max_logit=logit.max(), min_logit=logit.min())


# =============================================================================
# synthetic CartPole-v1 (stands in for gym, which is not installable here;
# dynamics are the standard cart-pole ODE, reward shaping per core/env/gym_env.py:78)
# =============================================================================
class CartPoleOracle:
    """W independent CartPole-v1 envs, float64 dynamics, float32 observations.
    Euler tau=0.02, force 10 N, theta limit 12 deg, x limit 2.4, 500-step cap.
    reward = -1 if done else 0.1 (gym_env.py:78).  Reset draws U(-0.05,0.05)^4 from
    a per-env splitmix64 stream so the C++ collector can reproduce it bit-exactly."""

    GRAV, MC, MP, LEN, FMAG, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    THETA_LIM = 12 * 2 * math.pi / 360
    X_LIM = 2.4
    MAX_STEPS = 500

    def __init__(self, W, seed=0):
        self.W = W
        self.rng = np.asarray([(seed * 0x9E3779B97F4A7C15 + (w + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF for w in range(W)], dtype=np.uint64)
        self.s = np.zeros((W, 4), np.float64)
        self.t = np.zeros(W, np.int64)
        for w in range(W):
            self._reset(w)

    def _next_u01(self, w):
        # splitmix64
        with np.errstate(over="ignore"):
            self.rng[w] = self.rng[w] + np.uint64(0x9E3779B97F4A7C15)
            x = self.rng[w]
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
        return float(x >> np.uint64(11)) * (1.0 / 9007199254740992.0)

    def _reset(self, w):
        for k in range(4):
            self.s[w, k] = -0.05 + 0.1 * self._next_u01(w)
        self.t[w] = 0

    def obs(self):
        return self.s.astype(np.float32)

    def step(self, action):
        """action int[W] in {0,1}.  Returns (next_obs f32 (W,4), reward f32 (W,), done bool (W,));
        envs that finished are auto-reset AFTER next_obs is taken (distributed_manager.py:91)."""
        W = self.W
        nxt = np.zeros((W, 4), np.float32)
        rew = np.zeros(W, np.float32)
        done = np.zeros(W, bool)
        total_mass = self.MC + self.MP
        pml = self.MP * self.LEN
        for w in range(W):
            x, xd, th, thd = self.s[w]
            force = self.FMAG if int(action[w]) == 1 else -self.FMAG
            ct, st = math.cos(th), math.sin(th)
            temp = (force + pml * thd * thd * st) / total_mass
            thacc = (self.GRAV * st - ct * temp) / (self.LEN * (4.0 / 3.0 - self.MP * ct * ct / total_mass))
            xacc = temp - pml * thacc * ct / total_mass
            x = x + self.TAU * xd
            xd = xd + self.TAU * xacc
            th = th + self.TAU * thd
            thd = thd + self.TAU * thacc
            self.s[w] = (x, xd, th, thd)
            self.t[w] += 1
            d = x < -self.X_LIM or x > self.X_LIM or th < -self.THETA_LIM or th > self.THETA_LIM or self.t[w] >= self.MAX_STEPS
            nxt[w] = self.s[w].astype(np.float32)
            done[w] = d
            rew[w] = -1.0 if d else 0.1
            if d:
                self._reset(w)
        return nxt, rew, done


# =============================================================================
# synthetic continuous control (stand-in for MuJoCo at config.ppo.mujoco shapes; mirrors csrc/jh_env.hip bit for bit)
# =============================================================================
class ControlOracle:
    """W envs, float64 dynamics  s'_i = 0.95 s_i + 0.05 sum_j P[i][j] a_j + 0.02 sin(s_{(i+1) mod S}),
    P[i][j] = 0.5 sin(1.7 (i+1) + 2.3 (j+1));  reward = s'_0 + 0.1 - 0.001 |a|^2;  done when |s'_0| > 2 or after 1000
    steps;  reset U(-0.05, 0.05)^S from a per-env splitmix64 stream;  auto-reset AFTER next_obs is taken."""

    MAX_STEPS = 1000

    def __init__(self, W, S=11, A=3, seed=0):
        self.W, self.S, self.A = W, S, A
        self.rng = np.asarray([(seed * 0x9E3779B97F4A7C15 + (w + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF for w in range(W)], dtype=np.uint64)
        self.s = np.zeros((W, S), np.float64)
        self.t = np.zeros(W, np.int64)
        for w in range(W):
            self._reset(w)

    _next_u01 = CartPoleOracle._next_u01

    def _reset(self, w):
        for k in range(self.S):
            self.s[w, k] = -0.05 + 0.1 * self._next_u01(w)
        self.t[w] = 0

    def obs(self):
        return self.s.astype(np.float32)

    def step(self, action):
        W, S, A = self.W, self.S, self.A
        action = np.asarray(action, dtype=np.float32).reshape(W, A)
        nxt = np.zeros((W, S), np.float32)
        rew = np.zeros(W, np.float32)
        done = np.zeros(W, bool)
        for w in range(W):
            a = [float(v) for v in action[w]]
            a2 = 0.0
            for j in range(A):
                a2 += a[j] * a[j]
            new = [0.0] * S
            for i in range(S):
                drive = 0.0
                for j in range(A):
                    drive += 0.5 * math.sin(1.7 * (i + 1) + 2.3 * (j + 1)) * a[j]
                new[i] = 0.95 * self.s[w, i] + 0.05 * drive + 0.02 * math.sin(self.s[w, (i + 1) % S])
            self.s[w] = new
            self.t[w] += 1
            d = bool(new[0] > 2.0 or new[0] < -2.0 or self.t[w] >= self.MAX_STEPS)
            nxt[w] = self.s[w].astype(np.float32)
            done[w] = d
            rew[w] = np.float32(new[0] + 0.1 - 0.001 * a2)
            if d:
                self._reset(w)
        return nxt, rew, done
