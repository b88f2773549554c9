// Value-based losses for gfx950:
//   jh_td_loss   target-Q (+double-Q, +n-step fold) with Huber or PER-weighted MSE, forward and
//                backward to Q(s), priorities td^alpha   (dqn.py:128-141, double.py:28-39,
//                multistep.py:41-50, per.py:54-74, ape_x.py:96-116)
//   jh_c51_loss  categorical n-step projection + cross-entropy, forward and backward to the
//                online logits, priorities KL^alpha      (rainbow.py:167-239, c51.py:68-109)
// Both are latency/HBM-bound (B = 32..512 rows): one lane per row (TD) / one wave per row with
// atoms across lanes and LDS staging of the per-sample projection operands (C51).
#include "jh_common.h"
#include "jh_fused.h"

// ============================================================================ TD losses
struct TdArgs {
  int B, A, n, n_step, flags;
  const float *q, *qno, *qnt, *action, *reward, *done, *weights;
  float gamma, alpha;
  float *grad_q, *prio, *stats, *partial;
};

// partial per block: {sum_loss_terms, max_q, sum_td}
__global__ void __launch_bounds__(256) jh_td_loss_kernel(TdArgs a) {
  __shared__ float s_red[16];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < a.B;
  float lterm = 0.f, qa = -3.4e38f, td = 0.f;
  if (on) {
    int act = (int)a.action[i];
    act = act < 0 ? 0 : (act >= a.A ? a.A - 1 : act);
    const float* q = a.q + (size_t)i * a.A;
    const float* qnt = a.qnt + (size_t)i * a.A;
    qa = q[act];
    float boot;
    if (a.flags & JH_TD_DOUBLE) {
      const float* qno = a.qno + (size_t)i * a.A;
      int best = 0;
      float bv = qno[0];
      for (int k = 1; k < a.A; ++k)
        if (qno[k] > bv) { bv = qno[k]; best = k; }  // first maximum, like torch.argmax
      boot = qnt[best];
    } else {
      boot = qnt[0];
      for (int k = 1; k < a.A; ++k) boot = fmaxf(boot, qnt[k]);
    }
    float y;
    if (a.n_step > 0) {  // multistep.py:47-48 / ape_x.py:105-106
      y = boot;
      for (int j = a.n - 1; j >= 0; --j) {
        const float r = a.reward[(size_t)i * a.n + j], d = a.done[(size_t)i * a.n + j];
        y = r + (1.f - d) * a.gamma * y;
      }
    } else {
      const float r = a.reward[i], d = a.done[i];
      if (a.flags & JH_TD_DOUBLE) y = r + boot * (a.gamma * (1.f - d));  // double.py:35-37, per.py:62-64
      else y = r + (1.f - d) * a.gamma * boot;                            // dqn.py:135-137
    }
    const float diff = qa - y;
    const float invB = 1.f / (float)a.B;
    float g;
    td = fabsf(y - qa);
    if (a.flags & JH_TD_PER) {
      const float w = a.weights[i];
      lterm = w * td * td;  // per.py:74
      g = 2.f * w * diff * invB;
      if (a.prio) a.prio[i] = powf(td, a.alpha);  // per.py:68
    } else {
      const float ad = fabsf(diff);
      lterm = ad < 1.f ? 0.5f * diff * diff : ad - 0.5f;  // smooth_l1, beta = 1
      g = (ad < 1.f ? diff : (diff > 0.f ? 1.f : -1.f)) * invB;
      if (a.prio) a.prio[i] = td;
    }
    float* gq = a.grad_q + (size_t)i * a.A;
    for (int k = 0; k < a.A; ++k) gq[k] = (k == act) ? g : 0.f;
  }
  const float s_l = jh_block_reduce(lterm, s_red, JhAdd(), 0.f);
  const float m_q = jh_block_reduce(qa, s_red, JhMax(), -3.4e38f);
  const float s_t = jh_block_reduce(td, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) {
      if (a.stats) {
        a.stats[0] = s_l / (float)a.B;
        a.stats[1] = m_q;
        a.stats[2] = s_t / (float)a.B;
        __threadfence_system();  // stats may be device-mapped host memory and [3] is the arrival mark the host spins on: payload first
        a.stats[3] = 0.f;
      }
    } else {
      float* p = a.partial + 3 * (size_t)blockIdx.x;
      p[0] = s_l; p[1] = m_q; p[2] = s_t;
    }
  }
}

__global__ void __launch_bounds__(256) jh_td_finish_kernel(int nb, int B, const float* __restrict__ partial,
                                                           float* __restrict__ stats) {
  __shared__ float s_red[16];
  float l = 0.f, m = -3.4e38f, t = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256) {
    l += partial[3 * b];
    m = fmaxf(m, partial[3 * b + 1]);
    t += partial[3 * b + 2];
  }
  l = jh_block_reduce(l, s_red, JhAdd(), 0.f);
  m = jh_block_reduce(m, s_red, JhMax(), -3.4e38f);
  t = jh_block_reduce(t, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0) {
    stats[0] = l / (float)B;
    stats[1] = m;
    stats[2] = t / (float)B;
    __threadfence_system();  // payload before the arrival mark (mapped host memory, jh_host_wait_marks)
    stats[3] = 0.f;
  }
}

JH_EXPORT int jh_td_loss(jh_ctx* ctx, int32_t B, int32_t A, int32_t n_step, int32_t flags, const float* d_q,
                         const float* d_q_next_online, const float* d_q_next_target, const float* d_action,
                         const float* d_reward, const float* d_done, const float* d_weights, float gamma, float alpha,
                         float* d_grad_q, float* d_prio, float* d_stats, jh_stream stream) {
  JH_ARG(ctx && d_q && d_q_next_target && d_action && d_reward && d_done && d_grad_q);
  JH_ARG(B > 0 && A > 0 && n_step >= 0);
  JH_ARG(!(flags & JH_TD_DOUBLE) || d_q_next_online);
  JH_ARG(!(flags & JH_TD_PER) || d_weights);
  TdArgs a{};
  a.B = B; a.A = A; a.n_step = n_step; a.n = n_step > 0 ? n_step : 1; a.flags = flags;
  a.q = d_q; a.qno = d_q_next_online; a.qnt = d_q_next_target; a.action = d_action; a.reward = d_reward;
  a.done = d_done; a.weights = d_weights; a.gamma = gamma; a.alpha = alpha; a.grad_q = d_grad_q; a.prio = d_prio;
  a.stats = d_stats;
  const int nb = (B + 255) / 256;
  if (nb > 1) {
    void* scratch = nullptr;
    int rc = jh_ctx_scratch(ctx, sizeof(float) * 3 * (size_t)nb, &scratch);
    if (rc) return rc;
    a.partial = (float*)scratch;
  }
  JH_LAUNCH(jh_td_loss_kernel, dim3(nb), dim3(256), 0, jh_s(stream), a);
  JH_LAUNCH_CHECK();
  if (nb > 1 && d_stats) {
    JH_LAUNCH(jh_td_finish_kernel, dim3(1), dim3(256), 0, jh_s(stream), nb, B, a.partial, d_stats);
    JH_LAUNCH_CHECK();
  }
  return JH_OK;
}

// ============================================================================ C51 / Rainbow
// torch.linspace(v_min, v_max, K) in float32 (symmetric form used by ATen)
__device__ __forceinline__ float support_z(int k, int K, float v_min, float v_max) {
  const float step = (v_max - v_min) / (float)(K - 1);
  return (k < K / 2) ? v_min + step * (float)k : v_max - step * (float)(K - 1 - k);
}

// softmax over the K atoms of one action row: lane j owns atoms j, j+64, ...  (K <= 256: <= 4 per lane)
// returns this lane's probabilities in p[0..3] and the expectation sum_k z_k p_k (all lanes).
__device__ __forceinline__ float atom_softmax_z(const float z[4], int K, int lane, float v_min, float v_max, float p[4], float& row_max, float& row_min) {
  float m = -3.4e38f, mn = 3.4e38f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    m = fmaxf(m, z[s]);
    if (k < K) mn = fminf(mn, z[s]);
  }
  m = jh_wave_max(m);
  mn = jh_wave_min(mn);
  float se = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) se += (lane + 64 * s < K) ? expf(z[s] - m) : 0.f;
  se = jh_wave_sum(se);
  const float lse = logf(se);
  float q = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    p[s] = k < K ? expf((z[s] - m) - lse) : 0.f;  // exp(log_softmax)  rainbow.py:287
    q += k < K ? support_z(k, K, v_min, v_max) * p[s] : 0.f;
  }
  row_max = m;
  row_min = mn;
  return jh_wave_sum(q);
}
__device__ __forceinline__ float atom_softmax(const float* __restrict__ row, int K, int lane, float v_min, float v_max,
                                              float p[4], float& row_max, float& row_min) {
  float z[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    z[s] = k < K ? row[k] : -3.4e38f;
  }
  return atom_softmax_z(z, K, lane, v_min, v_max, p, row_max, row_min);
}

// Dueling combine of one (set, sample): this lane's share of mean_a xa[b][a][k] (the sum runs over a in order, like
// jh_rb_duel_fwd_kernel), and one action's logits (xa - mean) + xv, written to the set's logits as a side effect.
// All three sets at once, four actions per round: 12 fetches in flight (round 5; one fetch per wait before: 3 sets x A dependent
// round trips in front of the first softmax).  Addresses are clamped (atom K - 1, action A - 1) and the surplus dropped; the sums
// run over a in ascending order as before.
__device__ __forceinline__ void duel_mean3(const C51Duel& d, int b, int A, int K, int lane, float (&mean0)[4], float (&mean1)[4], float (&mean2)[4]) {
  const float* xa0 = d.xa[0] + (size_t)b * d.ld_a;
  const float* xa1 = d.xa[1] + (size_t)b * d.ld_a;
  const float* xa2 = d.xa[2] + (size_t)b * d.ld_a;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (64 * s < K) {  // wave-uniform
      const int kc = k < K ? k : K - 1;
      for (int a0 = 0; a0 < A; a0 += 4) {
        float v0[4], v1[4], v2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int o = (a0 + u < A ? a0 + u : A - 1) * K + kc;
          v0[u] = xa0[o]; v1[u] = xa1[o]; v2[u] = xa2[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (a0 + u < A) { s0 += v0[u]; s1 += v1[u]; s2 += v2[u]; }
      }
    }
    mean0[s] = (k < K ? s0 : 0.f) / (float)A;
    mean1[s] = (k < K ? s1 : 0.f) / (float)A;
    mean2[s] = (k < K ? s2 : 0.f) / (float)A;
  }
}
// one action's row: the loads (issued with the mean's loads, in front of any store), then combine + store
__device__ __forceinline__ void duel_row_load(const C51Duel& d, int set, int b, int aa, int K, int lane, float ra[4], float rv[4]) {
  const float* __restrict__ xa = d.xa[set] + (size_t)b * d.ld_a + (size_t)aa * K;
  const float* __restrict__ xv = d.xv[set] + (size_t)b * d.ld_v;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    ra[s] = k < K ? xa[k] : 0.f;
    rv[s] = k < K ? xv[k] : 0.f;
  }
}
__device__ __forceinline__ void duel_row_put(const C51Duel& d, int set, int b, int aa, int A, int K, int lane, const float ra[4], const float rv[4],
                                             const float mean[4], float z[4]) {
  float* o = d.out[set] + ((size_t)b * A + aa) * K;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    z[s] = -3.4e38f;
    if (k < K) {
      z[s] = (ra[s] - mean[s]) + rv[s];
      o[k] = z[s];
    }
  }
}

// The projection's sum over source atoms (rainbow.py:212-217) for the atoms this lane owns: m_k = d0 * mean_j [l_j == k == u_j] + ... in
// ASCENDING source atom j like the reference's sum over dim 1.  Source atom j = 64 s + i lives in lane i's registers (slot s); it reaches all
// lanes through v_readlane (j is wave-uniform: a scalar broadcast, a few cycles) instead of five LDS reads per j -- at K = 51 that loop
// was 5.9 of the block kernel's 14.4 us (in-kernel clock).  Same operations in the same order as the LDS form: bit-identical results.
__device__ __forceinline__ float bcast_lane(float v, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i)); }
template <int NS>  // NS = ceil(K / 64) slots in use: no per-slot branches inside the loop over source atoms
__device__ __forceinline__ void c51_project_ns(const float l[4], const float u[4], const float wl[4], const float wu[4], const float tp[4], int K, int lane, float d0,
                                               float m[4], float& msum) {
  float term[NS], non[NS], kf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { term[s] = 0.f; non[s] = 0.f; kf[s] = (float)(lane + 64 * s); }
#pragma unroll
  for (int sj = 0; sj < NS; ++sj) {
    const int nj = K - 64 * sj < 64 ? K - 64 * sj : 64;
    for (int i = 0; i < nj; ++i) {
      const float lj = bcast_lane(l[sj], i), uj = bcast_lane(u[sj], i), wlj = bcast_lane(wl[sj], i), wuj = bcast_lane(wu[sj], i), tpj = bcast_lane(tp[sj], i);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float val = (lj == kf[s] ? wlj : 0.f) + (uj == kf[s] ? wuj : 0.f);
        term[s] += ((lj == kf[s] && uj == kf[s]) ? 1.f : 0.f) + val;  // rainbow.py:212-214
        non[s] += tpj * val;                                          // rainbow.py:215-217
      }
    }
  }
  msum = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    m[s] = 0.f;
    if (s < NS) {
      const bool own = lane + 64 * s < K;
      m[s] = own ? d0 * (term[s] / (float)K) + (1.f - d0) * non[s] : 0.f;  // torch.mean over the source atoms
    }
    msum += m[s];
  }
}
__device__ __forceinline__ void c51_project(const float l[4], const float u[4], const float wl[4], const float wu[4], const float tp[4], int K, int lane, float d0,
                                            float m[4], float& msum) {
  if (K <= 64) c51_project_ns<1>(l, u, wl, wu, tp, K, lane, d0, m, msum);
  else if (K <= 128) c51_project_ns<2>(l, u, wl, wu, tp, K, lane, d0, m, msum);
  else if (K <= 192) c51_project_ns<3>(l, u, wl, wu, tp, K, lane, d0, m, msum);
  else c51_project_ns<4>(l, u, wl, wu, tp, K, lane, d0, m, msum);
}

// 4 waves per block, one wave per sample.
__global__ void __launch_bounds__(256) jh_c51_kernel(C51Args a) {
  __shared__ float s_part[4][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wid;
  const int K = a.K;
  float kl = 0.f, maxq = -3.4e38f, maxl = -3.4e38f, minl = 3.4e38f;
  if (b < a.B) {
    int act = (int)a.action[b];
    act = act < 0 ? 0 : (act >= a.A ? a.A - 1 : act);
    // ---- online distribution of the taken action + stats over all actions
    float p_act[4] = {0.f, 0.f, 0.f, 0.f};
    for (int aa = 0; aa < a.A; ++aa) {
      float p[4], rmx, rmn;
      const float q = atom_softmax(a.logit + ((size_t)b * a.A + aa) * K, K, lane, a.v_min, a.v_max, p, rmx, rmn);
      maxq = fmaxf(maxq, q);
      maxl = fmaxf(maxl, rmx);
      minl = fminf(minl, rmn);
      if (aa == act) {
#pragma unroll
        for (int s = 0; s < 4; ++s) p_act[s] = p[s];
      }
    }
    // ---- greedy next action: online net (rainbow.py:177-181) or the target net itself (c51.py:76-80)
    const float* sel = (a.flags & JH_C51_DOUBLE) ? a.next_logit : a.target_logit;
    int best = 0;
    float bq = -3.4e38f;
    for (int aa = 0; aa < a.A; ++aa) {
      float p[4], rmx, rmn;
      const float q = atom_softmax(sel + ((size_t)b * a.A + aa) * K, K, lane, a.v_min, a.v_max, p, rmx, rmn);
      if (q > bq) { bq = q; best = aa; }  // first maximum
    }
    float tp[4], rmx, rmn;
    (void)atom_softmax(a.target_logit + ((size_t)b * a.A + best) * K, K, lane, a.v_min, a.v_max, tp, rmx, rmn);
    // ---- n-step Bellman image of every atom and its two neighbours on the support
    const float range = a.v_max - a.v_min;
    const float dz = (float)(((double)a.v_max - (double)a.v_min) / (double)(K - 1));
    float pl[4], pu[4], pwl[4], pwu[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = lane + 64 * s;
      pl[s] = pu[s] = -1.f; pwl[s] = pwu[s] = 0.f;
      if (j < K) {
        float Tz = support_z(j, K, a.v_min, a.v_max);
        for (int i = a.n - 1; i >= 0; --i) {  // rainbow.py:188-193
          const float r = a.reward[(size_t)b * a.n + i], d = a.done[(size_t)b * a.n + i];
          Tz = r + (1.f - d) * a.gamma * Tz;
        }
        const float bb = fminf(fmaxf(Tz - a.v_min, 0.f), range) / dz;  // rainbow.py:195
        pl[s] = floorf(bb);
        pu[s] = ceilf(bb);
        pwl[s] = pu[s] - bb;  // mass to l;  integral b -> l == u -> both weights 0 (quirk kept)
        pwu[s] = bb - pl[s];
      }
    }
    const float d0 = a.done[(size_t)b * a.n];  // terminal branch keyed on done[:,0]  rainbow.py:212
    float m[4];
    float msum;
    c51_project(pl, pu, pwl, pwu, tp, K, lane, d0, m, msum);
    msum = jh_wave_sum(msum);
    const float norm = fmaxf(msum, 1e-8f);  // rainbow.py:218-220
    float klp = 0.f, mt_sum = 0.f;
    float mt[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      m[s] = m[s] / norm;
      const float pc = fmaxf(p_act[s], 1e-8f);
      klp += (lane + 64 * s < K) ? m[s] * logf(pc) : 0.f;
      mt[s] = (p_act[s] >= 1e-8f) ? m[s] : 0.f;  // clamp(min=1e-8) blocks the gradient below it
      mt_sum += mt[s];
    }
    kl = -jh_wave_sum(klp);  // rainbow.py:227
    mt_sum = jh_wave_sum(mt_sum);
    // ---- effective per-sample weight: rainbow's (B,1)*(B,) broadcast makes it the batch MEAN
    // (every wave summing all B weights itself made this kernel O(B^2): 1.13 ms at B = 65 536, 2 % of the HBM roofline)
    const float weff = (a.flags & JH_C51_PER) ? *a.wmean : 1.f;
    const float scale = weff / (float)a.B;
    for (int aa = 0; aa < a.A; ++aa) {
      float* g = a.grad + ((size_t)b * a.A + aa) * K;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = lane + 64 * s;
        if (k < K) g[k] = (aa == act) ? (-mt[s] + p_act[s] * mt_sum) * scale : 0.f;
      }
    }
    if (lane == 0) {
      if (a.kl) a.kl[b] = kl;
      if (a.prio) a.prio[b] = powf(kl, a.alpha);  // rainbow.py:228
    }
  }
  if (lane == 0) {
    s_part[wid][0] = (b < a.B) ? kl : 0.f;
    s_part[wid][1] = maxq;
    s_part[wid][2] = maxl;
    s_part[wid][3] = minl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sk = 0.f, mq = -3.4e38f, ml = -3.4e38f, nl = 3.4e38f;
    for (int w = 0; w < 4; ++w) {
      sk += s_part[w][0];
      mq = fmaxf(mq, s_part[w][1]);
      ml = fmaxf(ml, s_part[w][2]);
      nl = fminf(nl, s_part[w][3]);
    }
    float* p = a.partial + 4 * (size_t)blockIdx.x;
    p[0] = sk; p[1] = mq; p[2] = ml; p[3] = nl;
  }
}


// Small batches (the configs' B = 32): ONE WORKGROUP per sample, the per-action softmaxes spread over its 4 waves.
// The wave-per-sample kernel above runs 2A + 1 softmaxes (three dependent wave reductions each) one after the other
// on a single wave -- at B = 32 that chain, not bandwidth, is the whole cost.  Every value is computed by the same
// instruction sequence as above (one wave per softmax row, same shuffle trees, same source-atom order in the
// projection), so the results are bit-identical.
// Dynamic LDS: [K] p_act, [A] selector Q, [4][3] per-wave stats (+ [A][K] target logits in the dueling form).
// DUEL (Rainbow's own step, jh_rbnet_c51_step): the logits do not exist yet -- the kernel reads the advantage / value streams
// of the three forwards, forms (xa - mean_a xa) + xv per row as jh_rb_duel_fwd_kernel would (same expression, same order: the
// logits it leaves in d.out are bit-identical) and hands back d(loss)/d(xa), d(loss)/d(xv) instead of d(loss)/d(logits).
template <bool DUEL>
__global__ void __launch_bounds__(256) jh_c51_block_kernel(C51Args a, C51Duel d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x, K = a.K;
  float* s_pact = smem;
  float* s_qsel = s_pact + K;     // [A]
  float* s_stat = s_qsel + a.A;   // [4][3]: max Q, max logit, min logit of the rows this wave saw
  float* s_z2 = s_stat + 12;      // DUEL: [A][K] the target net's combined logits (wave 0 picks the greedy action's row from here)
  int act = (int)a.action[b];
  act = act < 0 ? 0 : (act >= a.A ? a.A - 1 : act);
  // what wave 0 needs after the barrier, requested now so that it arrives under the softmaxes instead of as three more dependent round
  // trips: the n-step rewards / dones (lane i holds step i) and this lane's share of the batch's IS weights (summed in the old order)
  // (round 5: every wave fetches them, unconditionally and from clamped addresses -- inside `if (wid == 0)` the block ended in a wait,
  // and so did the row loads inside `if (wid < A)` below: three dependent round trips where one batch does)
  const bool pre = a.n <= 64;
  const bool per = (a.flags & JH_C51_PER) != 0;
  const int ln = lane < a.n ? lane : a.n - 1;
  const float ld_r = a.reward[(size_t)b * a.n + ln], ld_d = a.done[(size_t)b * a.n + ln];
  const float* wp = per ? a.weights : a.reward;
  const float ld_w0 = wp[(per && lane < a.B) ? lane : 0], ld_w1 = wp[(per && lane + 64 < a.B) ? lane + 64 : 0];
  // ---- phase 1: online softmaxes (stats + the taken action's distribution) and the selector's Q, actions strided over waves
  float maxq = -3.4e38f, maxl = -3.4e38f, minl = 3.4e38f;
  const float* sel = (a.flags & JH_C51_DOUBLE) ? a.next_logit : a.target_logit;
  const int sel_set = (a.flags & JH_C51_DOUBLE) ? 1 : 2;
  float mean0[4], mean1[4], mean2[4];
  float ra[3][4], rv[3][4];
  if (DUEL) {
    // this wave's first row of the three sets and everything the means need: one batch of loads, one wait
    const int aw = wid < a.A ? wid : a.A - 1;
    for (int j = 0; j < 3; ++j) duel_row_load(d, j, b, aw, K, lane, ra[j], rv[j]);
    duel_mean3(d, b, a.A, K, lane, mean0, mean1, mean2);
  }
  const float pre_r = (pre && lane < a.n) ? ld_r : 0.f, pre_d = (pre && lane < a.n) ? ld_d : 0.f;
  float pre_ws = 0.f;
  if (per) {
    if (lane < a.B) pre_ws += ld_w0;
    if (lane + 64 < a.B) pre_ws += ld_w1;
    if (wid == 0)
      for (int i = lane + 128; i < a.B; i += 64) pre_ws += a.weights[i];
  }
  for (int aa = wid; aa < a.A; aa += 4) {
    float p[4], rmx, rmn, q, q2, p2[4];
    if (DUEL) {
      float z0[4], z1[4], z2[4];
      if (aa != wid)
        for (int j = 0; j < 3; ++j) duel_row_load(d, j, b, aa, K, lane, ra[j], rv[j]);
      duel_row_put(d, 0, b, aa, a.A, K, lane, ra[0], rv[0], mean0, z0);
      duel_row_put(d, 1, b, aa, a.A, K, lane, ra[1], rv[1], mean1, z1);  // every row of every set is written: the logits are an output
      duel_row_put(d, 2, b, aa, a.A, K, lane, ra[2], rv[2], mean2, z2);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (lane + 64 * s < K) s_z2[aa * K + lane + 64 * s] = z2[s];
      q = atom_softmax_z(z0, K, lane, a.v_min, a.v_max, p, rmx, rmn);
      float r2x, r2n;
      q2 = atom_softmax_z(sel_set == 1 ? z1 : z2, K, lane, a.v_min, a.v_max, p2, r2x, r2n);
    } else {
      q = atom_softmax(a.logit + ((size_t)b * a.A + aa) * K, K, lane, a.v_min, a.v_max, p, rmx, rmn);
    }
    maxq = fmaxf(maxq, q);
    maxl = fmaxf(maxl, rmx);
    minl = fminf(minl, rmn);
    if (aa == act) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (lane + 64 * s < K) s_pact[lane + 64 * s] = p[s];
    }
    if (!DUEL) q2 = atom_softmax(sel + ((size_t)b * a.A + aa) * K, K, lane, a.v_min, a.v_max, p2, rmx, rmn);
    if (lane == 0) s_qsel[aa] = q2;
  }
  if (lane == 0) { s_stat[wid * 3 + 0] = maxq; s_stat[wid * 3 + 1] = maxl; s_stat[wid * 3 + 2] = minl; }
  __syncthreads();
  if (wid != 0) {
    if (DUEL) return;  // the gradient rows of the other actions are -mean_a g: wave 0 writes them with the taken action's row
    // ---- the other waves only write the zero gradient rows of the actions that were not taken (phase 4 needs nothing from them)
    for (int aa = wid - 1; aa < a.A; aa += 3) {
      if (aa == act) continue;
      float* g = a.grad + ((size_t)b * a.A + aa) * K;
      for (int k = lane; k < K; k += 64) g[k] = 0.f;
    }
    return;
  }
  // ---- wave 0 from here: greedy next action = first maximum (rainbow.py:177-181 / c51.py:76-80)
  int best = 0;
  float bq = -3.4e38f;
  for (int aa = 0; aa < a.A; ++aa) {
    const float q = s_qsel[aa];
    if (q > bq) { bq = q; best = aa; }
  }
  float tp[4], rmx, rmn;
  if (DUEL) {
    float zt[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) zt[s] = (lane + 64 * s < K) ? s_z2[best * K + lane + 64 * s] : -3.4e38f;  // what phase 1 wrote for this row
    (void)atom_softmax_z(zt, K, lane, a.v_min, a.v_max, tp, rmx, rmn);
  } else {
    (void)atom_softmax(a.target_logit + ((size_t)b * a.A + best) * K, K, lane, a.v_min, a.v_max, tp, rmx, rmn);
  }
  float p_act[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) p_act[s] = (lane + 64 * s < K) ? s_pact[lane + 64 * s] : 0.f;
  // ---- n-step Bellman image of every atom and its two neighbours on the support
  const float range = a.v_max - a.v_min;
  const float dz = (float)(((double)a.v_max - (double)a.v_min) / (double)(K - 1));
  float Tzs[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) Tzs[s] = support_z(lane + 64 * s, K, a.v_min, a.v_max);
  for (int i = a.n - 1; i >= 0; --i) {  // rainbow.py:188-193 (every atom sees the same operations in the same order as one loop per atom)
    const float r = pre ? __shfl(pre_r, i, 64) : a.reward[(size_t)b * a.n + i], dn = pre ? __shfl(pre_d, i, 64) : a.done[(size_t)b * a.n + i];
#pragma unroll
    for (int s = 0; s < 4; ++s) Tzs[s] = r + (1.f - dn) * a.gamma * Tzs[s];
  }
  float pl[4], pu[4], pwl[4], pwu[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float bb = fminf(fmaxf(Tzs[s] - a.v_min, 0.f), range) / dz;  // rainbow.py:195
    pl[s] = floorf(bb);
    pu[s] = ceilf(bb);
    pwl[s] = pu[s] - bb;  // mass to l;  integral b -> l == u -> both weights 0 (quirk kept)
    pwu[s] = bb - pl[s];
  }
  const float d0 = pre ? __shfl(pre_d, 0, 64) : a.done[(size_t)b * a.n];  // terminal branch keyed on done[:,0]  rainbow.py:212
  float m[4];
  float msum;
  c51_project(pl, pu, pwl, pwu, tp, K, lane, d0, m, msum);
  msum = jh_wave_sum(msum);
  const float norm = fmaxf(msum, 1e-8f);  // rainbow.py:218-220
  float klp = 0.f, mt_sum = 0.f;
  float mt[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    m[s] = m[s] / norm;
    const float pc = fmaxf(p_act[s], 1e-8f);
    klp += (lane + 64 * s < K) ? m[s] * logf(pc) : 0.f;
    mt[s] = (p_act[s] >= 1e-8f) ? m[s] : 0.f;  // clamp(min=1e-8) blocks the gradient below it
    mt_sum += mt[s];
  }
  const float kl = -jh_wave_sum(klp);  // rainbow.py:227
  mt_sum = jh_wave_sum(mt_sum);
  float weff = 1.f;  // rainbow's (B,1)*(B,) broadcast makes the per-sample weight the batch MEAN
  if (a.flags & JH_C51_PER) weff = jh_wave_sum(pre_ws) / (float)a.B;
  const float scale = weff / (float)a.B;
  if (DUEL) {
    // g is zero outside the taken action's row: sum_a g = that row (jh_rb_duel_bwd_kernel's sum of A terms, A - 1 of them 0.f)
    float* dxa = d.dxa + (size_t)b * d.ld_a;
    float* dxv = d.dxv + (size_t)b * d.ld_v;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = lane + 64 * s;
      if (k < K) {
        const float gv = (-mt[s] + p_act[s] * mt_sum) * scale;
        const float mean = gv / (float)a.A;
        dxv[k] = gv;
        for (int aa = 0; aa < a.A; ++aa) dxa[aa * K + k] = (aa == act ? gv : 0.f) - mean;
      }
    }
  } else {
    float* g = a.grad + ((size_t)b * a.A + act) * K;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = lane + 64 * s;
      if (k < K) g[k] = (-mt[s] + p_act[s] * mt_sum) * scale;
    }
  }
  if (lane == 0) {
    if (a.kl) a.kl[b] = kl;
    if (a.prio) a.prio[b] = powf(kl, a.alpha);  // rainbow.py:228
    float mq = -3.4e38f, ml = -3.4e38f, nl = 3.4e38f;
    const int nw = a.A < 4 ? a.A : 4;  // waves that saw at least one action row
    for (int w = 0; w < nw; ++w) {
      mq = fmaxf(mq, s_stat[w * 3 + 0]);
      ml = fmaxf(ml, s_stat[w * 3 + 1]);
      nl = fminf(nl, s_stat[w * 3 + 2]);
    }
    float* p = a.partial + 4 * (size_t)b;
    p[0] = kl; p[1] = mq; p[2] = ml; p[3] = nl;
  }
}

// PER: the same launch writes the new priorities a.prio into the tree's leaves (jh_per_delta_body; the climb follows).
template <bool PER>
__global__ void __launch_bounds__(256) jh_c51_finish_kernel(int nb, C51Args a, PerDeltaArgs pa) {
  __shared__ float s_red[16];
  if (PER && blockIdx.x == 1) {  // a workgroup of its own beside the statistics: neither waits for the other
    __shared__ unsigned long long s_key[kPerChunk];
    __shared__ double s_new[kPerChunk];
    __shared__ double s_redd[16];
    jh_per_delta_body(pa, a.B, s_key, s_new, s_redd);
    return;
  }
  float sk = 0.f, mq = -3.4e38f, ml = -3.4e38f, nl = 3.4e38f, ws = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256) {
    sk += a.partial[4 * b];
    mq = fmaxf(mq, a.partial[4 * b + 1]);
    ml = fmaxf(ml, a.partial[4 * b + 2]);
    nl = fminf(nl, a.partial[4 * b + 3]);
  }
  if (a.flags & JH_C51_PER)
    for (int i = threadIdx.x; i < a.B; i += 256) ws += a.weights[i];
  sk = jh_block_reduce(sk, s_red, JhAdd(), 0.f);
  mq = jh_block_reduce(mq, s_red, JhMax(), -3.4e38f);
  ml = jh_block_reduce(ml, s_red, JhMax(), -3.4e38f);
  nl = jh_block_reduce(nl, s_red, JhMin(), 3.4e38f);
  ws = jh_block_reduce(ws, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0 && a.stats) {
    const float mean_kl = sk / (float)a.B;
    a.stats[0] = (a.flags & JH_C51_PER) ? (ws / (float)a.B) * mean_kl : mean_kl;  // rainbow.py:235 / c51.py:104
    a.stats[1] = mq;
    a.stats[2] = ml;
    a.stats[3] = nl;
    a.stats[4] = mean_kl;
    a.stats[6] = 0.f;
    __threadfence_system();  // payload before the arrival marks [5], [7] (mapped host memory, jh_host_wait_marks)
    a.stats[5] = a.stats[7] = 0.f;
  }
}

int jh_c51_run(jh_ctx* ctx, C51Args a, const C51Duel* duel, const PerDeltaArgs* per, hipStream_t st) {
  const int B = a.B, K = a.K;
  const bool per_block = B <= 1024;  // latency regime: one workgroup per sample, softmaxes spread over its waves
  if (duel && !per_block) return jh_fail(JH_ERR_ARG, "jh_c51_run: the fused dueling form is the block kernel's (B <= 1024)");
  if (per && B > kPerChunk) return jh_fail(JH_ERR_ARG, "jh_c51_run: fused priority write-back takes B <= %d", kPerChunk);
  const int nb = per_block ? B : (B + 3) / 4;
  void* scratch = nullptr;
  int rc = jh_ctx_scratch(ctx, sizeof(float) * (4 * (size_t)nb + 4), &scratch);
  if (rc) return rc;
  a.partial = (float*)scratch;
  if (!per_block && (a.flags & JH_C51_PER)) {
    float* wmean = (float*)scratch + 4 * (size_t)nb;
    rc = jh_mean_f32(ctx, B, a.weights, wmean, (jh_stream)st);
    if (rc) return rc;
    a.wmean = wmean;
  }
  if (per_block) {
    const size_t lds = sizeof(float) * ((size_t)K + (size_t)a.A + 12 + (duel ? (size_t)a.A * K : 0));
    if (duel) JH_LAUNCH(jh_c51_block_kernel<true>, dim3(nb), dim3(256), lds, st, a, *duel);
    else JH_LAUNCH(jh_c51_block_kernel<false>, dim3(nb), dim3(256), lds, st, a, C51Duel{});
  } else {
    JH_LAUNCH(jh_c51_kernel, dim3(nb), dim3(256), 0, st, a);
  }
  JH_LAUNCH_CHECK();
  if (per) JH_LAUNCH(jh_c51_finish_kernel<true>, dim3(2), dim3(256), 0, st, nb, a, *per);
  else JH_LAUNCH(jh_c51_finish_kernel<false>, dim3(1), dim3(256), 0, st, nb, a, PerDeltaArgs{});
  JH_LAUNCH_CHECK();
  return JH_OK;
}

JH_EXPORT int jh_c51_loss(jh_ctx* ctx, int32_t B, int32_t A, int32_t K, int32_t n_step, int32_t flags,
                          const float* d_logit, const float* d_next_logit_online, const float* d_target_logit,
                          const float* d_action, const float* d_reward, const float* d_done, const float* d_weights,
                          float v_min, float v_max, float gamma, float alpha, float* d_grad_logit, float* d_prio,
                          float* d_kl, float* d_stats, jh_stream stream) {
  JH_ARG(ctx && d_logit && d_target_logit && d_action && d_reward && d_done && d_grad_logit);
  JH_ARG(B > 0 && A > 0 && K > 1 && K <= 256 && n_step >= 0);
  JH_ARG(!(flags & JH_C51_DOUBLE) || d_next_logit_online);
  JH_ARG(!(flags & JH_C51_PER) || d_weights);
  C51Args a{};
  a.B = B; a.A = A; a.K = K; a.n = n_step > 0 ? n_step : 1; a.flags = flags;
  a.logit = d_logit; a.next_logit = d_next_logit_online; a.target_logit = d_target_logit; a.action = d_action;
  a.reward = d_reward; a.done = d_done; a.weights = d_weights; a.v_min = v_min; a.v_max = v_max; a.gamma = gamma;
  a.alpha = alpha; a.grad = d_grad_logit; a.prio = d_prio; a.kl = d_kl; a.stats = d_stats;
  return jh_c51_run(ctx, a, nullptr, nullptr, jh_s(stream));
}

// ============================================================================ batched acting of the value-net agents
// DQN.act / ApeX.act / C51.act / Rainbow.act (dqn.py:76-92, ape_x.py:64-77, c51.py:50-66, rainbow.py:140-152) for N
// actors in ONE call: Q(s) from the network's outputs (K = 1: the outputs are Q; K > 1: expectation of the atoms under
// softmax, rainbow.py:285-292), per-actor epsilon-greedy with the HOST's random draws (the reference draws
// `np.random.random() < epsilon` and `np.random.randint` per act() call: passing them in keeps that RNG stream),
// first maximum like torch.argmax, and Q of the action taken (Ape-X's actor-side priority needs it).
// One wave per actor row: lanes over the atoms (K > 1) / one lane (K = 1).
__global__ void __launch_bounds__(256) jh_value_act_kernel(int N, int A, int K, const float* __restrict__ logits, float v_min, float dz,
                                                           const float* __restrict__ eps, const double* __restrict__ u,
                                                           const int64_t* __restrict__ rand_action, int64_t* __restrict__ action,
                                                           float* __restrict__ q_taken, float* __restrict__ q_all) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float best = -3.4e38f, q_rand = 0.f;
  int best_a = 0;
  const int ra = rand_action ? (int)rand_action[row] : 0;
  for (int a = 0; a < A; ++a) {
    float q;
    const float* z = logits + ((size_t)row * A + a) * K;
    if (K == 1) {
      q = z[0];
    } else {
      float m = -3.4e38f;
      for (int k = lane; k < K; k += 64) m = fmaxf(m, z[k]);
      m = jh_wave_max(m);
      float se = 0.f, sz = 0.f;
      for (int k = lane; k < K; k += 64) {
        const float e = expf(z[k] - m);
        se += e;
        sz += e * (v_min + dz * (float)k);
      }
      q = jh_wave_sum(sz) / jh_wave_sum(se);
    }
    if (q_all && lane == 0) q_all[(size_t)row * A + a] = q;
    if (q > best) { best = q; best_a = a; }
    if (a == ra) q_rand = q;
  }
  if (lane == 0) {
    const bool explore = eps && u && u[row] < (double)eps[row];
    // q first, the action last: a host that waits for the actions to arrive in device-mapped memory (BatchedValueActors)
    // then finds q in place too
    if (q_taken) {
      q_taken[row] = explore ? q_rand : best;
      __threadfence_system();
    }
    action[row] = explore ? ra : best_a;
  }
}

JH_EXPORT int jh_value_act(jh_ctx* ctx, int32_t N, int32_t A, int32_t K, const float* d_logits, float v_min, float v_max,
                           const float* h_eps, const double* h_u, const int64_t* h_rand_action, int64_t* d_action,
                           float* d_q_taken, float* d_q_all, jh_stream stream) {
  JH_ARG(ctx && d_logits && d_action);
  JH_ARG(N > 0 && A > 0 && K > 0);
  JH_ARG((h_eps == nullptr) == (h_u == nullptr) && (h_eps == nullptr) == (h_rand_action == nullptr));
  hipStream_t st = jh_s(stream);
  const float* d_eps = nullptr;
  const double* d_u = nullptr;
  const int64_t* d_ra = nullptr;
  jh_pinned_slab* slab = nullptr;
  if (h_eps) {  // the draws ride in a pinned, device-mapped slab the kernel reads in place
    const size_t o_u = ((sizeof(float) * (size_t)N + 255) & ~(size_t)255), o_r = o_u + ((sizeof(double) * (size_t)N + 255) & ~(size_t)255);
    int rc = jh_ctx_slab(ctx, o_r + sizeof(int64_t) * (size_t)N + 256, &slab);
    if (rc) return rc;
    memcpy(slab->host, h_eps, sizeof(float) * (size_t)N);
    memcpy((char*)slab->host + o_u, h_u, sizeof(double) * (size_t)N);
    memcpy((char*)slab->host + o_r, h_rand_action, sizeof(int64_t) * (size_t)N);
    d_eps = (const float*)slab->dev;
    d_u = (const double*)((char*)slab->dev + o_u);
    d_ra = (const int64_t*)((char*)slab->dev + o_r);
  }
  const float dz = K > 1 ? (v_max - v_min) / (float)(K - 1) : 0.f;
  JH_LAUNCH(jh_value_act_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, A, K, d_logits, v_min, dz, d_eps, d_u, d_ra, d_action, d_q_taken, d_q_all);
  JH_LAUNCH_CHECK();
  return slab ? jh_ctx_slab_release(ctx, slab, st) : JH_OK;
}
