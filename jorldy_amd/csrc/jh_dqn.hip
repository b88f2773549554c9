// Value-based losses for gfx950:
//   jh_td_loss   target-Q (+double-Q, +n-step fold) with Huber or PER-weighted MSE, forward and
//                backward to Q(s), priorities td^alpha   (dqn.py:128-141, double.py:28-39,
//                multistep.py:41-50, per.py:54-74, ape_x.py:96-116)
//   jh_c51_loss  categorical n-step projection + cross-entropy, forward and backward to the
//                online logits, priorities KL^alpha      (rainbow.py:167-239, c51.py:68-109)
// Both are latency/HBM-bound (B = 32..512 rows): one lane per row (TD) / one wave per row with
// atoms across lanes and LDS staging of the per-sample projection operands (C51).
#include "jh_common.h"

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
struct C51Args {
  int B, A, K, n, flags;
  const float *logit, *next_logit, *target_logit, *action, *reward, *done, *weights;
  float v_min, v_max, gamma, alpha;
  float *grad, *prio, *kl, *stats, *partial;
  const float* wmean;  // wave-per-sample kernel: the batch mean of `weights`, computed once by jh_mean_f32 in front of it
};

// torch.linspace(v_min, v_max, K) in float32 (symmetric form used by ATen)
__device__ __forceinline__ float support_z(int k, int K, float v_min, float v_max) {
  const float step = (v_max - v_min) / (float)(K - 1);
  return (k < K / 2) ? v_min + step * (float)k : v_max - step * (float)(K - 1 - k);
}

// softmax over the K atoms of one action row: lane j owns atoms j, j+64, ...  (K <= 256: <= 4 per lane)
// returns this lane's probabilities in p[0..3] and the expectation sum_k z_k p_k (all lanes).
__device__ __forceinline__ float atom_softmax(const float* __restrict__ row, int K, int lane, float v_min, float v_max,
                                              float p[4], float& row_max, float& row_min) {
  float z[4];
  float m = -3.4e38f, mn = 3.4e38f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    z[s] = k < K ? row[k] : -3.4e38f;
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

// 4 waves per block, one wave per sample.  Dynamic LDS: per wave 5*K floats
// (l, u as float, wl, wu, target_p of the chosen action).
__global__ void __launch_bounds__(256) jh_c51_kernel(C51Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float s_part[4][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wid;
  const int K = a.K;
  float* s_l = smem + (size_t)wid * 5 * K;
  float* s_u = s_l + K;
  float* s_wl = s_u + K;
  float* s_wu = s_wl + K;
  float* s_tp = s_wu + K;
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
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = lane + 64 * s;
      if (j < K) {
        float Tz = support_z(j, K, a.v_min, a.v_max);
        for (int i = a.n - 1; i >= 0; --i) {  // rainbow.py:188-193
          const float r = a.reward[(size_t)b * a.n + i], d = a.done[(size_t)b * a.n + i];
          Tz = r + (1.f - d) * a.gamma * Tz;
        }
        const float bb = fminf(fmaxf(Tz - a.v_min, 0.f), range) / dz;  // rainbow.py:195
        const float l = floorf(bb), u = ceilf(bb);
        s_l[j] = l;
        s_u[j] = u;
        s_wl[j] = u - bb;  // mass to l;  integral b -> l == u -> both weights 0 (quirk kept)
        s_wu[j] = bb - l;
        s_tp[j] = tp[s];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const float d0 = a.done[(size_t)b * a.n];  // terminal branch keyed on done[:,0]  rainbow.py:212
    float m[4];
    float msum = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = lane + 64 * s;
      float term = 0.f, non = 0.f;
      if (k < K) {
        const float kf = (float)k;
        for (int j = 0; j < K; ++j) {  // ascending source atom, like the sum over dim 1
          const float l = s_l[j], u = s_u[j];
          const float val = (l == kf ? s_wl[j] : 0.f) + (u == kf ? s_wu[j] : 0.f);
          term += ((l == kf && u == kf) ? 1.f : 0.f) + val;  // rainbow.py:212-214
          non += s_tp[j] * val;                              // rainbow.py:215-217
        }
        term = term / (float)K;  // torch.mean over the source atoms
      }
      m[s] = k < K ? d0 * term + (1.f - d0) * non : 0.f;
      msum += m[s];
    }
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
// Dynamic LDS: [K] p_act, [K] l, u, wl, wu, tp, [A] selector Q, [4][3] per-wave stats.
__global__ void __launch_bounds__(256) jh_c51_block_kernel(C51Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x, K = a.K;
  float* s_pact = smem;
  float* s_l = s_pact + K;
  float* s_u = s_l + K;
  float* s_wl = s_u + K;
  float* s_wu = s_wl + K;
  float* s_tp = s_wu + K;
  float* s_qsel = s_tp + K;       // [A]
  float* s_stat = s_qsel + a.A;   // [4][3]: max Q, max logit, min logit of the rows this wave saw
  int act = (int)a.action[b];
  act = act < 0 ? 0 : (act >= a.A ? a.A - 1 : act);
  // ---- phase 1: online softmaxes (stats + the taken action's distribution) and the selector's Q, actions strided over waves
  float maxq = -3.4e38f, maxl = -3.4e38f, minl = 3.4e38f;
  const float* sel = (a.flags & JH_C51_DOUBLE) ? a.next_logit : a.target_logit;
  for (int aa = wid; aa < a.A; aa += 4) {
    float p[4], rmx, rmn;
    const float q = atom_softmax(a.logit + ((size_t)b * a.A + aa) * K, K, lane, a.v_min, a.v_max, p, rmx, rmn);
    maxq = fmaxf(maxq, q);
    maxl = fmaxf(maxl, rmx);
    minl = fminf(minl, rmn);
    if (aa == act) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (lane + 64 * s < K) s_pact[lane + 64 * s] = p[s];
    }
    float p2[4];
    const float q2 = atom_softmax(sel + ((size_t)b * a.A + aa) * K, K, lane, a.v_min, a.v_max, p2, rmx, rmn);
    if (lane == 0) s_qsel[aa] = q2;
  }
  if (lane == 0) { s_stat[wid * 3 + 0] = maxq; s_stat[wid * 3 + 1] = maxl; s_stat[wid * 3 + 2] = minl; }
  __syncthreads();
  if (wid != 0) {
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
  (void)atom_softmax(a.target_logit + ((size_t)b * a.A + best) * K, K, lane, a.v_min, a.v_max, tp, rmx, rmn);
  float p_act[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) p_act[s] = (lane + 64 * s < K) ? s_pact[lane + 64 * s] : 0.f;
  // ---- n-step Bellman image of every atom and its two neighbours on the support
  const float range = a.v_max - a.v_min;
  const float dz = (float)(((double)a.v_max - (double)a.v_min) / (double)(K - 1));
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int j = lane + 64 * s;
    if (j < K) {
      float Tz = support_z(j, K, a.v_min, a.v_max);
      for (int i = a.n - 1; i >= 0; --i) {  // rainbow.py:188-193
        const float r = a.reward[(size_t)b * a.n + i], d = a.done[(size_t)b * a.n + i];
        Tz = r + (1.f - d) * a.gamma * Tz;
      }
      const float bb = fminf(fmaxf(Tz - a.v_min, 0.f), range) / dz;  // rainbow.py:195
      const float l = floorf(bb), u = ceilf(bb);
      s_l[j] = l;
      s_u[j] = u;
      s_wl[j] = u - bb;  // mass to l;  integral b -> l == u -> both weights 0 (quirk kept)
      s_wu[j] = bb - l;
      s_tp[j] = tp[s];
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const float d0 = a.done[(size_t)b * a.n];  // terminal branch keyed on done[:,0]  rainbow.py:212
  float m[4];
  float msum = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    float term = 0.f, non = 0.f;
    if (k < K) {
      const float kf = (float)k;
      for (int j = 0; j < K; ++j) {  // ascending source atom, like the sum over dim 1
        const float l = s_l[j], u = s_u[j];
        const float val = (l == kf ? s_wl[j] : 0.f) + (u == kf ? s_wu[j] : 0.f);
        term += ((l == kf && u == kf) ? 1.f : 0.f) + val;  // rainbow.py:212-214
        non += s_tp[j] * val;                              // rainbow.py:215-217
      }
      term = term / (float)K;  // torch.mean over the source atoms
    }
    m[s] = k < K ? d0 * term + (1.f - d0) * non : 0.f;
    msum += m[s];
  }
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
  if (a.flags & JH_C51_PER) {
    float ws = 0.f;
    for (int i = lane; i < a.B; i += 64) ws += a.weights[i];
    weff = jh_wave_sum(ws) / (float)a.B;
  }
  const float scale = weff / (float)a.B;
  float* g = a.grad + ((size_t)b * a.A + act) * K;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = lane + 64 * s;
    if (k < K) g[k] = (-mt[s] + p_act[s] * mt_sum) * scale;
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

__global__ void __launch_bounds__(256) jh_c51_finish_kernel(int nb, C51Args a) {
  __shared__ float s_red[16];
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
  const bool per_block = B <= 1024;  // latency regime: one workgroup per sample, softmaxes spread over its waves
  const int nb = per_block ? B : (B + 3) / 4;
  void* scratch = nullptr;
  int rc = jh_ctx_scratch(ctx, sizeof(float) * (4 * (size_t)nb + 4), &scratch);
  if (rc) return rc;
  a.partial = (float*)scratch;
  if (!per_block && (flags & JH_C51_PER)) {
    float* wmean = (float*)scratch + 4 * (size_t)nb;
    rc = jh_mean_f32(ctx, B, d_weights, wmean, stream);
    if (rc) return rc;
    a.wmean = wmean;
  }
  if (per_block) {
    const size_t lds = sizeof(float) * (6 * (size_t)K + (size_t)A + 12);
    JH_LAUNCH(jh_c51_block_kernel, dim3(nb), dim3(256), lds, jh_s(stream), a);
  } else {
    const size_t lds = sizeof(float) * 4 * 5 * (size_t)K;
    JH_LAUNCH(jh_c51_kernel, dim3(nb), dim3(256), lds, jh_s(stream), a);
  }
  JH_LAUNCH_CHECK();
  JH_LAUNCH(jh_c51_finish_kernel, dim3(1), dim3(256), 0, jh_s(stream), nb, a);
  JH_LAUNCH_CHECK();
  return JH_OK;
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
