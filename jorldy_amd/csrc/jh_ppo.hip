// PPO hot-path math for gfx950 (core/agent/ppo.py:83-165):
//   jh_gae                 GAE reverse scan + return + per-row standardisation (ppo.py:95-110)
//   jh_logp_*              log pi_old(a|s)                                      (ppo.py:84-93)
//   jh_ppo_loss_*          clipped surrogate + clipped value + entropy, fwd + bwd to the heads
// All of it is HBM/latency-bound elementwise + reduction work: SoA float32 columns, coalesced
// loads, wave-shuffle scans/reductions, no GEMM.  Compiled with -ffp-contract=off so products and
// sums round separately like the torch ops they replace.
#include "jh_common.h"
#include "jh_peer.h"
#include "jh_ppo_finish.h"

// ============================================================================ GAE
// One wave (64 lanes) per rollout row.  The recurrence adv[t] = delta[t] + c[t]*adv[t+1],
// c[t] = (1-d[t])*gamma*lambda, is a first-order linear recurrence: with f_t(x) = b_t + a_t*x the
// suffix composition F_t = f_t o f_{t+1} o ... is associative, so a 64-wide tile is scanned with
// 6 shuffle steps; tiles are walked from the row end carrying adv of the next tile's first lane.
// Algorithmic traffic: 4 reads + 2 writes = 24 B per transition (+ L2-resident re-reads for the
// standardisation passes).
__global__ void __launch_bounds__(256) jh_gae_kernel(int W, int T, float gamma, float lambda,
                                                     const float* __restrict__ reward,
                                                     const float* __restrict__ done,
                                                     const float* __restrict__ value,
                                                     const float* __restrict__ next_value,
                                                     float* __restrict__ adv, float* __restrict__ ret,
                                                     int standardize) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= W) return;  // whole wave exits together
  const size_t base = (size_t)row * (size_t)T;
  const int ntiles = (T + 63) >> 6;
  float carry = 0.f;  // adv[t+1] for the last lane of the current tile
  float sum = 0.f;
  for (int k = ntiles - 1; k >= 0; --k) {
    const int t = k * 64 + lane;
    float a = 1.f, b = 0.f, v = 0.f;
    if (t < T) {
      const float r = reward[base + t], d = done[base + t], vn = next_value[base + t];
      v = value[base + t];
      const float nd_g = (1.f - d) * gamma;  // ((1-done)*gamma) first, as in ppo.py:95,99-100
      b = r + nd_g * vn - v;                 // delta
      a = (t == T - 1) ? 0.f : nd_g * lambda;  // no bootstrap across the row end (ppo.py:98)
    }
    // inclusive suffix scan of (a,b) over the 64 lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float a2 = __shfl_down(a, off, 64);
      const float b2 = __shfl_down(b, off, 64);
      if (lane + off < 64) {
        b = b + a * b2;
        a = a * a2;
      }
    }
    const float x = b + a * carry;
    carry = __shfl(x, 0, 64);
    if (t < T) {
      adv[base + t] = x;
      ret[base + t] = x + v;  // ppo.py:103
      sum += x;
    }
  }
  if (!standardize) return;
  // per-row (adv - mean) / (std_unbiased + 1e-7)   ppo.py:105-108.  Each lane re-reads only the
  // elements it wrote itself, so no cross-lane visibility is needed.
  const float mean = jh_wave_sum(sum) / (float)T;
  float ssq = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float c = adv[base + t] - mean;
    ssq += c * c;
  }
  const float var = jh_wave_sum(ssq) / (float)(T - 1);
  const float inv = 1.f / (sqrtf(var) + 1e-7f);
  for (int t = lane; t < T; t += 64) adv[base + t] = (adv[base + t] - mean) * inv;
}

// Long rows (config.ppo.mujoco: T = 2048): one WORKGROUP of up to 16 waves per row instead of one wave walking T / 64 tiles serially
// (W = 32, T = 2048 ran on 32 waves of a 256-CU GPU: 46.8 us).  Every wave scans its tiles with the same 6-step shuffle scan,
// the tiles' aggregates (A_k, B_k) = f_{64k} o ... o f_{64k+63} go to LDS, wave 0 scans THOSE (the maps compose: the same associative
// operator one level up, <= 128 tiles = two passes of 64), and every element finishes as x = b + a * carry[tile + 1] from registers.
// E = elements per thread (T <= 1024 E).  Standardisation from registers too: two block reductions, no re-read of adv.
template <int E>
__global__ void __launch_bounds__(1024) jh_gae_long_kernel(int T, int nw, float gamma, float lambda, const float* __restrict__ reward,
                                                           const float* __restrict__ done, const float* __restrict__ value,
                                                           const float* __restrict__ next_value, float* __restrict__ adv, float* __restrict__ ret,
                                                           int standardize) {
  __shared__ float s_A[16 * E], s_B[16 * E], s_carry[16 * E + 1];
  __shared__ float s_red[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * (size_t)T;
  const int ntiles = (T + 63) >> 6;
  float a[E], b[E], v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {  // all loads of the thread in flight first
    const int k = wid + nw * e, t = k * 64 + lane;
    a[e] = 1.f; b[e] = 0.f; v[e] = 0.f;
    if (k < ntiles && t < T) {
      const float r = reward[base + t], d = done[base + t], vn = next_value[base + t];
      v[e] = value[base + t];
      const float nd_g = (1.f - d) * gamma;    // ((1-done)*gamma) first, as in ppo.py:95,99-100
      b[e] = r + nd_g * vn - v[e];             // delta
      a[e] = (t == T - 1) ? 0.f : nd_g * lambda;  // no bootstrap across the row end (ppo.py:98)
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float a2 = __shfl_down(a[e], off, 64);
      const float b2 = __shfl_down(b[e], off, 64);
      if (lane + off < 64) {
        b[e] = b[e] + a[e] * b2;
        a[e] = a[e] * a2;
      }
    }
    const int k = wid + nw * e;
    if (lane == 0 && k < ntiles) { s_A[k] = a[e]; s_B[k] = b[e]; }
  }
  __syncthreads();
  if (wid == 0) {  // carry[k] = adv at the first element of tile k = (F_k o F_{k+1} o ... o F_last)(0): suffix scan of the tile aggregates
    float cin = 0.f;
    for (int h = (ntiles - 1) >> 6; h >= 0; --h) {
      const int k = h * 64 + lane;
      float ta = k < ntiles ? s_A[k] : 1.f, tb = k < ntiles ? s_B[k] : 0.f;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float a2 = __shfl_down(ta, off, 64);
        const float b2 = __shfl_down(tb, off, 64);
        if (lane + off < 64) {
          tb = tb + ta * b2;
          ta = ta * a2;
        }
      }
      const float x = tb + ta * cin;
      if (k < ntiles) s_carry[k] = x;
      cin = __shfl(x, 0, 64);
    }
    if (lane == 0) s_carry[ntiles] = 0.f;
  }
  __syncthreads();
  float x[E], sum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = wid + nw * e, t = k * 64 + lane;
    x[e] = 0.f;
    if (k < ntiles && t < T) {
      x[e] = b[e] + a[e] * s_carry[k + 1];
      ret[base + t] = x[e] + v[e];  // ppo.py:103
      sum += x[e];
    }
  }
  if (!standardize) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = wid + nw * e, t = k * 64 + lane;
      if (k < ntiles && t < T) adv[base + t] = x[e];
    }
    return;
  }
  // per-row (adv - mean) / (std_unbiased + 1e-7)   ppo.py:105-108
  const float mean = jh_block_reduce(sum, s_red, JhAdd(), 0.f) / (float)T;
  float ssq = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = wid + nw * e, t = k * 64 + lane;
    if (k < ntiles && t < T) {
      const float c = x[e] - mean;
      ssq += c * c;
    }
  }
  const float var = jh_block_reduce(ssq, s_red, JhAdd(), 0.f) / (float)(T - 1);
  const float inv = 1.f / (sqrtf(var) + 1e-7f);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = wid + nw * e, t = k * 64 + lane;
    if (k < ntiles && t < T) adv[base + t] = (x[e] - mean) * inv;
  }
}

JH_EXPORT int jh_gae(jh_ctx* ctx, int32_t W, int32_t T, float gamma, float lambda, const float* d_reward,
                     const float* d_done, const float* d_value, const float* d_next_value, float* d_adv, float* d_ret,
                     int32_t standardize, jh_stream stream) {
  JH_ARG(ctx && d_reward && d_done && d_value && d_next_value && d_adv && d_ret);
  JH_ARG(W > 0 && T > 0);
  static const int kLongFrom = getenv("JH_GAE_LONG_FROM") ? atoi(getenv("JH_GAE_LONG_FROM")) : 257;  // rows of > 4 tiles: a workgroup per row
  const int ntiles = (T + 63) / 64;
  if (T >= kLongFrom && ntiles <= 128) {
    const int nw = ntiles < 16 ? ntiles : 16;
    const int e = (ntiles + nw - 1) / nw;
#define JH_GAE_LONG(E)                                                                                                                          \
    JH_LAUNCH_NAMED("jh_gae_long_kernel", (jh_gae_long_kernel<E>), dim3(W), dim3(64 * nw), 0, jh_s(stream), T, nw, gamma, lambda, d_reward, d_done, \
                    d_value, d_next_value, d_adv, d_ret, standardize)
    if (e <= 1) JH_GAE_LONG(1);
    else if (e <= 2) JH_GAE_LONG(2);
    else if (e <= 4) JH_GAE_LONG(4);
    else JH_GAE_LONG(8);
#undef JH_GAE_LONG
    JH_LAUNCH_CHECK();
    return JH_OK;
  }
  JH_LAUNCH(jh_gae_kernel, dim3((W + 3) / 4), dim3(256), 0, jh_s(stream), W, T, gamma, lambda, d_reward, d_done,
                     d_value, d_next_value, d_adv, d_ret, standardize);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// ============================================================================ minibatch rows of every epoch, once
// ppo.py:118-125 gathers state[idx], action[idx], adv[idx], ... inside the minibatch loop.  The index lists of all
// epochs are known before the loop, so the gathers are done ONCE here: the minibatch kernels then read plain
// consecutive rows (no idx -> row dependent load chain in front of every kernel of the update).
struct RowsArgs {
  int64_t n;
  const int64_t* idx;
  int n_cols, row_len;
  int off[8];        // first element of column c inside a packed row of row_len floats
  const float* src[8];
  float* dst[8];
};
__global__ void __launch_bounds__(256) jh_rows_gather_kernel(RowsArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n * a.row_len) return;
  const int64_t row = i / a.row_len;
  const int e = (int)(i - row * a.row_len);
  int c = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k)
    if (k < a.n_cols && e >= a.off[k]) c = k;
  const int w = (c + 1 < a.n_cols ? a.off[c + 1] : a.row_len) - a.off[c];
  const int j = e - a.off[c];
  a.dst[c][row * w + j] = a.src[c][a.idx[row] * w + j];
}

JH_EXPORT int jh_ppo_minibatch_rows(jh_ctx* ctx, int64_t n, const int64_t* d_idx, int32_t n_cols, const int32_t* elems,
                                    const float* const* d_src, float* const* d_dst, jh_stream stream) {
  JH_ARG(ctx && d_idx && elems && d_src && d_dst);
  JH_ARG(n > 0 && n_cols > 0 && n_cols <= 8);
  RowsArgs a{};
  a.n = n; a.idx = d_idx; a.n_cols = n_cols;
  int o = 0;
  for (int c = 0; c < n_cols; ++c) {
    JH_ARG(elems[c] > 0 && d_src[c] && d_dst[c]);
    a.off[c] = o; o += elems[c]; a.src[c] = d_src[c]; a.dst[c] = d_dst[c];
  }
  a.row_len = o;
  const int64_t tot = n * o;
  JH_LAUNCH(jh_rows_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, jh_s(stream), a);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// mean of n floats (ppo.py:112 `ret.mean()`): one workgroup, fixed order
__global__ void __launch_bounds__(1024) jh_mean_kernel(int64_t n, const float* __restrict__ x, float* __restrict__ out) {
  __shared__ float s_red[16];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += x[i];
  acc = jh_block_reduce(acc, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0) *out = acc / (float)n;
}

JH_EXPORT int jh_mean_f32(jh_ctx* ctx, int64_t n, const float* d_x, float* d_out, jh_stream stream) {
  JH_ARG(ctx && d_x && d_out && n > 0);
  JH_LAUNCH(jh_mean_kernel, dim3(1), dim3(1024), 0, jh_s(stream), n, d_x, d_out);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// ============================================================================ shared row math
#define JH_EPS32 1.1920929e-07f        // torch.finfo(float32).eps used by Categorical's clamp
#define JH_HALF_LOG_2PI 0.9189385332f  // log(sqrt(2*pi))
#define JH_ATANH_HI 0.99999988f        // fp32(1 - 1e-7)   ppo.py:87
#define JH_ATANH_LO -0.99999988f

struct PpoParams {
  float eps, vf, ent;
};

struct RowFwd {
  float ratio, surr1, surr2, smin, v, vclip, ret, adv, e1, e2, ent, minp, d_logp_scale;
};

// ---- discrete head -----------------------------------------------------------------------------
struct NormalD {
  double mu, std, z, th, lp;
};
constexpr int kContCache = 4;  // action dimensions whose double-precision terms the forward half of a row hands to its backward half
struct DiscRow {
  float m, lse, s;  // row max, log-sum-exp of (z-m), sum of p (re-normalisation by Categorical)
  NormalD n[kContCache];  // continuous policies only (dead code for discrete ones)
};

__device__ __forceinline__ DiscRow disc_prepare(const float* __restrict__ z, int A) {
  DiscRow r;
  float m = z[0];
  for (int k = 1; k < A; ++k) m = fmaxf(m, z[k]);
  float se = 0.f;
  for (int k = 0; k < A; ++k) se += expf(z[k] - m);
  r.m = m;
  r.lse = logf(se);
  float s = 0.f;
  for (int k = 0; k < A; ++k) s += expf((z[k] - m) - r.lse);
  r.s = s;
  return r;
}

// Categorical(probs=pi): pn = pi/sum(pi); logits = log(clamp(pn, eps, 1-eps))
__device__ __forceinline__ void disc_terms(const float* __restrict__ z, int k, const DiscRow& r, float& p, float& pn,
                                           float& c, float& lg, bool& inr) {
  p = expf((z[k] - r.m) - r.lse);
  pn = p / r.s;
  inr = (pn >= JH_EPS32) && (pn <= 1.f - JH_EPS32);
  c = fminf(fmaxf(pn, JH_EPS32), 1.f - JH_EPS32);
  lg = logf(c);
}

// ============================================================================ log pi_old
__global__ void __launch_bounds__(256) jh_logp_discrete_kernel(int64_t M, int A, const float* __restrict__ logits,
                                                               const float* __restrict__ action,
                                                               float* __restrict__ logp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const float* z = logits + i * A;
  float m = z[0];
  for (int k = 1; k < A; ++k) m = fmaxf(m, z[k]);
  float se = 0.f;
  for (int k = 0; k < A; ++k) se += expf(z[k] - m);
  int a = (int)action[i];
  a = a < 0 ? 0 : (a >= A ? A - 1 : a);
  const float pi = expf((z[a] - m) - logf(se));  // exp(log_softmax)  policy_value.py:22
  logp[i] = logf(pi);                            // pi.gather(1, a).log()  ppo.py:92
}

// The continuous head's per-element terms in DOUBLE.  d(loss)/d(log_std_raw) cancels (z - mu)^2 / (var std) against 1 / std, and
// (1 - tanh^2) against 1: with the fp32 device transcendentals (1-2 ulp) the head gradient sat 8.5e-6 of its largest entry from the
// float64 gradient, the reference's own torch-CPU fp32 3.3e-6 (round 4's float64 criterion, tests/test_baseline_width_gpu.py).  The
// terms are a handful of operations per (row, action dimension) of a latency-bound kernel: evaluating them in double costs nothing
// measurable and puts the kernel closer to the exact gradient than either fp32 evaluation.
__device__ __forceinline__ NormalD normal_terms(float mu_raw, float ls_raw, float act) {
  NormalD n;
  n.mu = fmin(fmax((double)mu_raw, -5.0), 5.0);  // policy_value.py:54
  n.th = tanh((double)ls_raw);
  n.std = exp(n.th);                              // policy_value.py:55-56
  const float a = fminf(fmaxf(act, JH_ATANH_LO), JH_ATANH_HI);  // the reference clamps the fp32 action tensor (ppo.py:87)
  n.z = atanh((double)a);
  const double dm = n.z - n.mu;
  n.lp = -(dm * dm) / (2.0 * n.std * n.std) - log(n.std) - 0.91893853320467274178;  // Normal.log_prob
  return n;
}
__device__ __forceinline__ float normal_logp(float mu_raw, float ls_raw, float act, float& mu, float& std, float& z) {
  const NormalD n = normal_terms(mu_raw, ls_raw, act);
  mu = (float)n.mu; std = (float)n.std; z = (float)n.z;
  return (float)n.lp;
}

__global__ void __launch_bounds__(256) jh_logp_continuous_kernel(int64_t MA, const float* __restrict__ mu_raw,
                                                                 const float* __restrict__ ls_raw,
                                                                 const float* __restrict__ action,
                                                                 float* __restrict__ logp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= MA) return;
  float mu, std, z;
  logp[i] = normal_logp(mu_raw[i], ls_raw[i], action[i], mu, std, z);
}

JH_EXPORT int jh_logp_discrete(jh_ctx* ctx, int64_t M, int32_t A, const float* d_logits, const float* d_action,
                               float* d_logp, jh_stream stream) {
  JH_ARG(ctx && d_logits && d_action && d_logp);
  JH_ARG(M > 0 && A > 0);
  JH_LAUNCH(jh_logp_discrete_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, jh_s(stream), M, A,
                     d_logits, d_action, d_logp);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

JH_EXPORT int jh_logp_continuous(jh_ctx* ctx, int64_t M, int32_t A, const float* d_mu_raw, const float* d_log_std_raw,
                                 const float* d_action, float* d_logp, jh_stream stream) {
  JH_ARG(ctx && d_mu_raw && d_log_std_raw && d_action && d_logp);
  JH_ARG(M > 0 && A > 0);
  const int64_t MA = M * A;
  JH_LAUNCH(jh_logp_continuous_kernel, dim3((unsigned)((MA + 255) / 256)), dim3(256), 0, jh_s(stream), MA,
                     d_mu_raw, d_log_std_raw, d_action, d_logp);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// ============================================================================ PPO loss
// Per-row forward terms shared by both heads once logp-sum is known.
struct RowCommon {
  float ratio, adv, v, vold, ret, vclip, smin;
  float g1, g2;  // sub-gradients of min(surr1, surr2) (ties split 1/2 like torch.min)
  bool in_clip, in_v;
};

__device__ __forceinline__ RowCommon row_common(float logp_diff_sum, float adv, float v, float vold, float ret,
                                                float eps) {
  RowCommon c;
  c.ratio = expf(logp_diff_sum);
  c.adv = adv;
  c.v = v;
  c.vold = vold;
  c.ret = ret;
  const float surr1 = c.ratio * adv;
  const float rc = fminf(fmaxf(c.ratio, 1.f - eps), 1.f + eps);
  const float surr2 = rc * adv;
  c.smin = fminf(surr1, surr2);
  c.g1 = surr1 < surr2 ? 1.f : (surr1 == surr2 ? 0.5f : 0.f);
  c.g2 = surr2 < surr1 ? 1.f : (surr1 == surr2 ? 0.5f : 0.f);
  c.in_clip = (c.ratio >= 1.f - eps) && (c.ratio <= 1.f + eps);
  const float dv = v - vold;
  c.in_v = (dv >= -eps) && (dv <= eps);
  c.vclip = vold + fminf(fmaxf(dv, -eps), eps);
  return c;
}

// partial layout per block: {sum_smin, sum_e1, sum_e2, sum_ent, max_ratio, min_prob}

template <bool CONT>
struct PpoArgs {
  int B, A;
  const float* h0;  // logits | mu_raw           [B][A]
  const float* h1;  // unused | log_std_raw      [B][A]
  const float* value_pred;  // [B]
  const int64_t* idx;       // [B] or null
  const float* action;      // [M] (disc) | [M][A] (cont)
  const float* adv;         // [M]
  const float* ret;         // [M]
  const float* value_old;   // [M]
  const float* logp_old;    // [M] (disc) | [M][A] (cont)
  float eps, vf, ent;
  float* g0;  // d logits | d mu_raw
  float* g1;  // unused   | d log_std_raw
  float* gv;  // d value_pred
  int ldg, ldv;  // row strides of g0 / g1 and of gv (A and 1; 8 and 8 when all three are packed into [B][8])
  int ldh, ldvp; // row strides of h0 / h1 and of value_pred (0: A and 1; jh_ppo_loss_packed: the row width of one [B][ld] output matrix)
  float* partial;
  float* stats;
  int nb;
  const float* totals;  // optional [6]: the partials already reduced (jh_ppo_totals_kernel) -- many blocks: every block re-reducing all nb partials is nb^2 reads
  // optional: the heads arrive as per-column-tile PARTIAL sums straight from the encoder GEMM epilogue
  // (hpart[tile][row][8], flat output order: head0[A], head1[A] (continuous), value); the fused kernel
  // sums them in tile order into LDS, which saves the separate heads kernel of the forward pass
  const float* hpart;
  int hp_tiles, hp_rows, hp_ld;  // hp_ld = floats per (tile, row): 4 when there are <= 4 head outputs, else 8
  // optional (single-workgroup launch only): Adam's hyper block; the step advances here when the minibatch update has
  // no norm kernel (jh_mlp.hip: four launches) -- the backward kernel must stay idempotent for the profiler's repeats
  float* hyper_advance;
  // optional, data-parallel learners (exact critic): the critic is max(mean(e1), mean(e2)) over the WHOLE minibatch (ppo.py:147-154), a
  // max of two means -- which of the two carries the gradient is known only after the ranks' sums have been reduced.  With defer_dv2
  // the kernel writes gv = d(vf mean(e1))/dv for every row and defer_dv2[i] = d(vf mean(e2))/dv, and critic_sums = {sum e1, sum e2} of
  // THIS rank's rows; jh_ppo_critic_select_kernel mixes the two with the branch weights of the reduced sums.
  float* defer_dv2;
  float* critic_sums;
};

// The per-row inputs of the loss that do not depend on the heads, fetched apart from their use: the single-workgroup kernel issues
// them BEFORE it sums the partial heads (they used to follow the partial sums and a barrier as two more dependent round trips).
struct RowIn {
  int64_t r;
  float adv, ret, vold;
  float action, logp_old;  // discrete policies only (continuous ones read A of each in the loop)
};
template <bool CONT>
__device__ __forceinline__ RowIn ppo_row_load(const PpoArgs<CONT>& a, int i) {
  RowIn in;
  in.r = a.idx ? a.idx[i] : (int64_t)i;
  in.adv = a.adv[in.r]; in.ret = a.ret[in.r]; in.vold = a.value_old[in.r];
  in.action = 0.f; in.logp_old = 0.f;
  if (!CONT) { in.action = a.action[in.r]; in.logp_old = a.logp_old[in.r]; }
  return in;
}

// The continuous policy's per-dimension inputs of one row -- head outputs, action, log pi_old -- for the first kContCache dimensions,
// fetched in ONE batch (the multi-workgroup kernels: the row loop read four values per dimension, waited, ran ~470 double-precision
// instructions and only then asked for the next dimension's four).  Dimensions past A re-read dimension A-1 (never used).
struct ContPre {
  float z0[kContCache], z1[kContCache], act[kContCache], lpo[kContCache];
};
template <bool CONT>
__device__ __forceinline__ ContPre ppo_cont_prefetch(const PpoArgs<CONT>& a, const float* z0, const float* z1, int64_t r) {
  ContPre c;
#pragma unroll
  for (int k = 0; k < kContCache; ++k) {
    const int kk = k < a.A ? k : a.A - 1;
    c.z0[k] = z0[kk]; c.z1[k] = z1[kk]; c.act[k] = a.action[r * a.A + kk]; c.lpo[k] = a.logp_old[r * a.A + kk];
  }
  return c;
}

// z0 / z1: this row's head-0 / head-1 vectors (global memory or the LDS staging), v: value prediction
// pre: optional (continuous policies), the first kContCache dimensions' inputs already in registers
template <bool CONT>
__device__ __forceinline__ void ppo_row_fwd(const PpoArgs<CONT>& a, const RowIn& in, const float* z0, const float* z1, float v,
                                            RowCommon& rc, float& ent_row, float& minp_row, DiscRow& dr, int& act_k,
                                            const ContPre* pre = nullptr) {
  const int64_t r = in.r;
  const float adv = in.adv, ret = in.ret, vold = in.vold;
  if (!CONT) {
    const float* z = z0;
    dr = disc_prepare(z, a.A);
    int ak = (int)in.action;
    ak = ak < 0 ? 0 : (ak >= a.A ? a.A - 1 : ak);
    act_k = ak;
    float ent = 0.f, logp = 0.f;
    for (int k = 0; k < a.A; ++k) {
      float p, pn, c, lg;
      bool inr;
      disc_terms(z, k, dr, p, pn, c, lg, inr);
      ent += pn * lg;
      if (k == ak) logp = lg;
    }
    ent_row = -ent;  // Categorical.entropy
    rc = row_common(logp - in.logp_old, adv, v, vold, ret, a.eps);
    minp_row = expf(logp);
  } else {
    double lsum = 0.0, ent = 0.0;
    float minp = 3.4e38f;
    auto dim = [&](const NormalD& n, float lpo) {
      lsum += n.lp - (double)lpo;
      ent += 0.5 + 0.91893853320467274178 + n.th;  // Normal.entropy = 1/2 + log sqrt(2 pi) + log(std), log(std) = tanh(log_std_raw)
      minp = fminf(minp, (float)exp(n.lp));
    };
    int k0 = 0;
    if (pre) {  // same terms, same order; the unrolled form keeps dr.n[] in registers under static indices
#pragma unroll
      for (int k = 0; k < kContCache; ++k)
        if (k < a.A) { dr.n[k] = normal_terms(pre->z0[k], pre->z1[k], pre->act[k]); dim(dr.n[k], pre->lpo[k]); }
      k0 = kContCache;
    }
    for (int k = k0; k < a.A; ++k) {
      const NormalD n = normal_terms(z0[k], z1[k], a.action[r * a.A + k]);
      if (k < kContCache) dr.n[k] = n;
      dim(n, a.logp_old[r * a.A + k]);
    }
    ent_row = (float)ent;  // summed over dims; the mean is over B*A elements (ppo.py:156)
    rc = row_common((float)lsum, adv, v, vold, ret, a.eps);
    minp_row = minp;
  }
}

template <bool CONT>
__device__ __forceinline__ void ppo_row_bwd(const PpoArgs<CONT>& a, int i, const float* z0, const float* z1,
                                            const RowCommon& rc, const DiscRow& dr, int act_k, float w1, float w2,
                                            const ContPre* pre = nullptr) {
  const float invB = 1.f / (float)a.B;
  const float d_ratio = -invB * (rc.g1 * rc.adv + (rc.in_clip ? rc.g2 * rc.adv : 0.f));
  const float d_logp = d_ratio * rc.ratio;
  // critic = max(c1, c2): weights w1/w2 (ties split); clamp passes grad inside [-eps, eps]
  if (a.defer_dv2) {
    a.gv[(size_t)i * a.ldv] = a.vf * (2.f * (rc.v - rc.ret) * invB);
    a.defer_dv2[i] = rc.in_v ? a.vf * (2.f * (rc.vclip - rc.ret) * invB) : 0.f;
  } else {
    const float dv = a.vf * (w1 * 2.f * (rc.v - rc.ret) * invB + (rc.in_v ? w2 * 2.f * (rc.vclip - rc.ret) * invB : 0.f));
    a.gv[(size_t)i * a.ldv] = dv;
  }
  if (!CONT) {
    const float* z = z0;
    const float ce = a.ent * invB;  // loss += ent_coef * (-mean(H)) = ent_coef/B * sum pn*lg
    float T1 = 0.f;
    for (int k = 0; k < a.A; ++k) {
      float p, pn, c, lg;
      bool inr;
      disc_terms(z, k, dr, p, pn, c, lg, inr);
      const float g_lg = (k == act_k ? d_logp : 0.f) + ce * pn;
      const float g_pn = ce * lg + (inr ? g_lg / c : 0.f);
      T1 += g_pn * p;
    }
    float T2 = 0.f;
    const float s = dr.s;
    for (int k = 0; k < a.A; ++k) {
      float p, pn, c, lg;
      bool inr;
      disc_terms(z, k, dr, p, pn, c, lg, inr);
      const float g_lg = (k == act_k ? d_logp : 0.f) + ce * pn;
      const float g_pn = ce * lg + (inr ? g_lg / c : 0.f);
      const float g_p = g_pn / s - T1 / (s * s);
      T2 += g_p * p;
    }
    for (int k = 0; k < a.A; ++k) {
      float p, pn, c, lg;
      bool inr;
      disc_terms(z, k, dr, p, pn, c, lg, inr);
      const float g_lg = (k == act_k ? d_logp : 0.f) + ce * pn;
      const float g_pn = ce * lg + (inr ? g_lg / c : 0.f);
      const float g_p = g_pn / s - T1 / (s * s);
      a.g0[(size_t)i * a.ldg + k] = g_p * p - p * T2;  // log_softmax backward
    }
  } else {
    const int64_t r = a.idx ? a.idx[i] : (int64_t)i;
    const double ce = -(double)a.ent / (double)(a.B * a.A);  // d(ent_coef * -mean(H)) / d log(std)
    const double dlp = (double)d_logp;
    auto dim = [&](int k, float mr, const NormalD& n) {
      const double var = n.std * n.std, dm = n.z - n.mu;
      const double d_mu = dlp * dm / var;
      // d/d(std): d_logp ((dm^2 - var) / (var std)) + ce / std; then std' = std (1 - th^2) through exp(tanh(.))
      const double d_std = dlp * ((dm * dm - var) / (var * n.std)) + ce / n.std;
      a.g0[(size_t)i * a.ldg + k] = (mr >= -5.f && mr <= 5.f) ? (float)d_mu : 0.f;
      a.g1[(size_t)i * a.ldg + k] = (float)(d_std * n.std * (1.0 - n.th * n.th));
    };
    int k0 = 0;
    if (pre) {
#pragma unroll
      for (int k = 0; k < kContCache; ++k)
        if (k < a.A) dim(k, pre->z0[k], dr.n[k]);
      k0 = kContCache;
    }
    for (int k = k0; k < a.A; ++k) {
      const float mr = z0[k], lr = z1[k];
      const NormalD n = k < kContCache ? dr.n[k] : normal_terms(mr, lr, a.action[r * a.A + k]);  // (the forward half evaluated them: ~10 double transcendentals per dimension)
      dim(k, mr, n);
    }
  }
}

// {sum, sum, sum, sum, max, min} over the workgroup (<= 16 waves)
__device__ __forceinline__ void ppo_block_reduce6(float (&v)[6], float (*red)[6]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += __shfl_xor(v[k], o, 64);
    v[4] = fmaxf(v[4], __shfl_xor(v[4], o, 64));
    v[5] = fminf(v[5], __shfl_xor(v[5], o, 64));
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[wid][k] = v[k];
  }
  __syncthreads();
  float r[6] = {0.f, 0.f, 0.f, 0.f, -3.4e38f, 3.4e38f};
  for (int w = 0; w < nw; ++w) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] += red[w][k];
    r[4] = fmaxf(r[4], red[w][4]);
    r[5] = fminf(r[5], red[w][5]);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = r[k];
}

// B <= 1024: one workgroup does forward, the 6 block reductions and backward with the row terms
// still in registers (one launch per minibatch instead of two).
// MAXT: the launch's thread limit (256-row minibatches get the 512-VGPR budget: all 32 tiles of a 512-wide net in flight at once)
template <bool CONT, int MAXT>
__global__ void __launch_bounds__(MAXT) jh_ppo_fused_kernel(PpoArgs<CONT> a) {
  __shared__ float s_red6[16][6];
  extern __shared__ __attribute__((aligned(16))) float s_z[];  // [B][8] when the heads come as partials
  const int i = threadIdx.x;
  const bool on = i < a.B;
  const float *z0 = nullptr, *z1 = nullptr;
  float vpred = 0.f;
  // first in the queue (round 5), the partial heads behind them.  Unconditional (threads beyond B re-read row 0): behind `if (on)` the
  // block ends in copies of the fetched values = a wait before the partial heads are even requested
  const RowIn rin = ppo_row_load<CONT>(a, on ? i : 0);
  if (a.hpart) {
    if (on) {
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bool wide = a.hp_ld == 8;
      if (!wide) {  // <= 4 head outputs (CartPole: 2 logits + value): TU tiles' float4 per round trip, sums in tile order
        constexpr int TU = MAXT <= 256 ? 32 : 16;
        for (int t0 = 0; t0 < a.hp_tiles; t0 += TU) {
          float4 q0[TU];
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            const int t = t0 + u < a.hp_tiles ? t0 + u : a.hp_tiles - 1;
            q0[u] = *reinterpret_cast<const float4*>(a.hpart + ((size_t)t * a.hp_rows + i) * 4);
          }
#pragma unroll
          for (int u = 0; u < TU; ++u)
            if (t0 + u < a.hp_tiles) { z[0] += q0[u].x; z[1] += q0[u].y; z[2] += q0[u].z; z[3] += q0[u].w; }
        }
      } else {
        constexpr int TU = MAXT <= 256 ? 16 : 8;
        for (int t0 = 0; t0 < a.hp_tiles; t0 += TU) {  // tile order: deterministic; 2 TU loads in flight per batch
          float4 q0[TU], q1[TU];
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            const int t = t0 + u < a.hp_tiles ? t0 + u : a.hp_tiles - 1;
            const float4* q = reinterpret_cast<const float4*>(a.hpart + ((size_t)t * a.hp_rows + i) * 8);
            q0[u] = q[0];
            q1[u] = q[1];
          }
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            if (t0 + u < a.hp_tiles) {
              z[0] += q0[u].x; z[1] += q0[u].y; z[2] += q0[u].z; z[3] += q0[u].w;
              z[4] += q1[u].x; z[5] += q1[u].y; z[6] += q1[u].z; z[7] += q1[u].w;
            }
          }
        }
      }
      float4* dst = reinterpret_cast<float4*>(s_z + (size_t)i * 8);
      dst[0] = make_float4(z[0], z[1], z[2], z[3]);
      dst[1] = make_float4(z[4], z[5], z[6], z[7]);
    }
    __syncthreads();
    z0 = s_z + (size_t)i * 8;
    z1 = z0 + a.A;
    if (on) vpred = z0[CONT ? 2 * a.A : a.A];
  } else if (on) {
    z0 = a.h0 + (size_t)i * a.ldh;
    z1 = CONT ? a.h1 + (size_t)i * a.ldh : nullptr;
    vpred = a.value_pred[(size_t)i * a.ldvp];
  }
  RowCommon rc{};
  DiscRow dr{};
  int act_k = 0;
  float ent_row = 0.f, minp = 3.4e38f;
  if (on) ppo_row_fwd<CONT>(a, rin, z0, z1, vpred, rc, ent_row, minp, dr, act_k);
  const float e1 = on ? (rc.v - rc.ret) * (rc.v - rc.ret) : 0.f;
  const float e2 = on ? (rc.vclip - rc.ret) * (rc.vclip - rc.ret) : 0.f;
  // the six block reductions share one shuffle tree pass and ONE LDS exchange (same arithmetic order as six
  // jh_block_reduce calls: fixed tree inside a wave, wave partials combined in wave order)
  float v6[6] = {on ? rc.smin : 0.f, e1, e2, on ? ent_row : 0.f, on ? rc.ratio : -3.4e38f, on ? minp : 3.4e38f};
  ppo_block_reduce6(v6, s_red6);
  float w1, w2;
  ppo_finish_stats(v6[0], v6[1], v6[2], v6[3], v6[4], v6[5], a.B, CONT ? a.B * a.A : a.B, a.vf, a.ent, w1, w2,
                   threadIdx.x == 0 ? a.stats : nullptr);
  if (a.critic_sums && threadIdx.x == 0) { a.critic_sums[0] = v6[1]; a.critic_sums[1] = v6[2]; }
  if (on) ppo_row_bwd<CONT>(a, i, z0, z1, rc, dr, act_k, w1, w2);
  // Adam's step counter + bias corrections (two double-precision pow: ~400 instructions) by the LAST wave, AFTER its rows: in front of
  // the partial sums (rounds 2-4) that wave reached the barrier ~1 us behind the others, every minibatch (nobody reads hyper here)
  if (a.hyper_advance && threadIdx.x == blockDim.x - 1) jh_adam_advance(a.hyper_advance);
}

// B > 1024: pass 1 writes per-block partials, pass 2 re-reduces them in every block (deterministic,
// no atomics) and recomputes the row terms instead of spilling them to HBM.
// Fetch order (round 5, both passes): rows past B work on row B-1 and drop the result, so that every load is unconditional and the
// compiler can batch them -- row index, then {adv, ret, value_old, value prediction, the policy's inputs} in one round trip; the six
// block reductions share one LDS exchange (12 barriers -> 1; same arithmetic order, see ppo_block_reduce6).
template <bool CONT>
__global__ void __launch_bounds__(256) jh_ppo_fwd_kernel(PpoArgs<CONT> a) {
  __shared__ float s_red6[16][6];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < a.B;
  const int ic = on ? i : a.B - 1;
  RowCommon rc{};
  DiscRow dr{};
  int act_k = 0;
  float ent_row = 0.f, minp = 3.4e38f;
  const float* z0 = a.h0 + (size_t)ic * a.ldh;
  const float* z1 = CONT ? a.h1 + (size_t)ic * a.ldh : nullptr;
  const RowIn rin = ppo_row_load<CONT>(a, ic);
  const float vpred = a.value_pred[(size_t)ic * a.ldvp];
  ContPre pre;
  if (CONT) pre = ppo_cont_prefetch<CONT>(a, z0, z1, rin.r);
  ppo_row_fwd<CONT>(a, rin, z0, z1, vpred, rc, ent_row, minp, dr, act_k, CONT ? &pre : nullptr);
  const float e1 = on ? (rc.v - rc.ret) * (rc.v - rc.ret) : 0.f;
  const float e2 = on ? (rc.vclip - rc.ret) * (rc.vclip - rc.ret) : 0.f;
  float v6[6] = {on ? rc.smin : 0.f, e1, e2, on ? ent_row : 0.f, on ? rc.ratio : -3.4e38f, on ? minp : 3.4e38f};
  ppo_block_reduce6(v6, s_red6);
  if (threadIdx.x == 0) {
    float* p = a.partial + (size_t)blockIdx.x * PPO_NPART;
    p[0] = v6[0]; p[1] = v6[1]; p[2] = v6[2]; p[3] = v6[3]; p[4] = v6[4]; p[5] = v6[5];
  }
}

// The per-block partials -> six totals, in the order every block of the backward pass would reduce them itself (same strided loop,
// same block reduction: bit-identical).  From a few dozen blocks on, nb blocks x nb partials of re-reads cost more than this launch:
// B = 1M rows = 4096 blocks re-read 400 MB of partials for a 44 MB problem (round 2: 14 % of the HBM roofline).
__device__ __forceinline__ void ppo_reduce_partials(const float* __restrict__ partial, int nb, float* s_red, float& t0, float& t1, float& t2, float& t3,
                                                    float& t4, float& t5) {
  t0 = t1 = t2 = t3 = 0.f; t4 = -3.4e38f; t5 = 3.4e38f;
  for (int b = threadIdx.x; b < nb; b += 256) {
    const float* p = partial + (size_t)b * PPO_NPART;
    t0 += p[0]; t1 += p[1]; t2 += p[2]; t3 += p[3];
    t4 = fmaxf(t4, p[4]); t5 = fminf(t5, p[5]);
  }
  t0 = jh_block_reduce(t0, s_red, JhAdd(), 0.f);
  t1 = jh_block_reduce(t1, s_red, JhAdd(), 0.f);
  t2 = jh_block_reduce(t2, s_red, JhAdd(), 0.f);
  t3 = jh_block_reduce(t3, s_red, JhAdd(), 0.f);
  t4 = jh_block_reduce(t4, s_red, JhMax(), -3.4e38f);
  t5 = jh_block_reduce(t5, s_red, JhMin(), 3.4e38f);
}

__global__ void __launch_bounds__(256) jh_ppo_totals_kernel(const float* __restrict__ partial, int nb, float* __restrict__ totals) {
  __shared__ float s_red[16];
  float t0, t1, t2, t3, t4, t5;
  ppo_reduce_partials(partial, nb, s_red, t0, t1, t2, t3, t4, t5);
  if (threadIdx.x == 0) { totals[0] = t0; totals[1] = t1; totals[2] = t2; totals[3] = t3; totals[4] = t4; totals[5] = t5; }
}

template <bool CONT>
__global__ void __launch_bounds__(256) jh_ppo_bwd_kernel(PpoArgs<CONT> a) {
  // the row's inputs first: they do not depend on the totals (see jh_ppo_fwd_kernel for the clamped row)
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < a.B;
  const int ic = on ? i : a.B - 1;
  const float* z0 = a.h0 + (size_t)ic * a.ldh;
  const float* z1 = CONT ? a.h1 + (size_t)ic * a.ldh : nullptr;
  const RowIn rin = ppo_row_load<CONT>(a, ic);
  const float vpred = a.value_pred[(size_t)ic * a.ldvp];
  ContPre pre;
  if (CONT) pre = ppo_cont_prefetch<CONT>(a, z0, z1, rin.r);
  float t[6];
  if (a.totals) {
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = a.totals[k];
  } else {  // ppo_launch: more than 64 partials always come reduced (a.totals)
    ppo_reduce_partials_wave(a.partial, a.nb, t);
  }
  float w1, w2;
  ppo_finish_stats(t[0], t[1], t[2], t[3], t[4], t[5], a.B, CONT ? a.B * a.A : a.B, a.vf, a.ent, w1, w2,
                   (blockIdx.x == 0 && threadIdx.x == 0) ? a.stats : nullptr);
  if (a.critic_sums && blockIdx.x == 0 && threadIdx.x == 0) { a.critic_sums[0] = t[1]; a.critic_sums[1] = t[2]; }
  RowCommon rc;
  DiscRow dr{};
  int act_k = 0;
  float ent_row, minp;
  ppo_row_fwd<CONT>(a, rin, z0, z1, vpred, rc, ent_row, minp, dr, act_k, CONT ? &pre : nullptr);
  if (on) ppo_row_bwd<CONT>(a, i, z0, z1, rc, dr, act_k, w1, w2, CONT ? &pre : nullptr);
}

// ONE launch for B > 1024 rows (round 6): the two passes above exist because the critic's branch -- max(mean(e1), mean(e2)), ppo.py:147-154 -- is
// known only after the whole minibatch has been reduced.  But nothing else of the backward half depends on the totals: the policy / entropy gradients are
// per-row, and the value gradient is one of two per-row candidates.  So every workgroup runs forward AND backward for its rows while they are in registers
// (the deferred form of the data-parallel learners: gv = branch 1's value gradient, defer_dv2 = branch 2's), leaves its six partials, and the LAST
// workgroup to arrive reduces them -- in the order the second pass used, so the totals, the statistics row and the branch weights are bit-identical -- and
// writes mix = {w1, w2}.  Whoever consumes the value gradient applies gv = w1 gv + w2 dv2 (the backward's first kernel, jh_mlp.hip: heads_bwd_dh).
// Partials travel as agent-scope (write-through) atomics like the tile engine's split-K hand-off: no fence, no L2 flush on eight XCDs.
template <bool CONT>
__global__ void __launch_bounds__(256) jh_ppo_onepass_kernel(PpoArgs<CONT> a, unsigned* __restrict__ ticket, float* __restrict__ mix) {
  __shared__ float s_red6[16][6];
  __shared__ float s_red[16];
  __shared__ int s_last;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < a.B;
  const int ic = on ? i : a.B - 1;
  RowCommon rc{};
  DiscRow dr{};
  int act_k = 0;
  float ent_row = 0.f, minp = 3.4e38f;
  const float* z0 = a.h0 + (size_t)ic * a.ldh;
  const float* z1 = CONT ? a.h1 + (size_t)ic * a.ldh : nullptr;
  const RowIn rin = ppo_row_load<CONT>(a, ic);
  const float vpred = a.value_pred[(size_t)ic * a.ldvp];
  ContPre pre;
  if (CONT) pre = ppo_cont_prefetch<CONT>(a, z0, z1, rin.r);
  ppo_row_fwd<CONT>(a, rin, z0, z1, vpred, rc, ent_row, minp, dr, act_k, CONT ? &pre : nullptr);
  const float e1 = on ? (rc.v - rc.ret) * (rc.v - rc.ret) : 0.f;
  const float e2 = on ? (rc.vclip - rc.ret) * (rc.vclip - rc.ret) : 0.f;
  float v6[6] = {on ? rc.smin : 0.f, e1, e2, on ? ent_row : 0.f, on ? rc.ratio : -3.4e38f, on ? minp : 3.4e38f};
  ppo_block_reduce6(v6, s_red6);
  if (threadIdx.x < PPO_NPART && threadIdx.x < 6) {
    float mine = v6[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) mine = threadIdx.x == k ? v6[k] : mine;
    __hip_atomic_store(a.partial + (size_t)blockIdx.x * PPO_NPART + threadIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (on) ppo_row_bwd<CONT>(a, i, z0, z1, rc, dr, act_k, 0.f, 0.f, CONT ? &pre : nullptr);  // a.defer_dv2: both branches, no weights
  if (!ticket) return;  // nb <= 64: whoever consumes the value gradient reduces the partials itself (PpoFinish, jh_mlp.hip) -- no ticket, no tail
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial stores have completed before its workgroup takes a ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == gridDim.x - 1;
    if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  // the second pass's reduction over the partials, fetched agent-coherently: nb <= 64 the wave form, else the block form (ppo_launch gave
  // the second pass `totals` from jh_ppo_totals_kernel = ppo_reduce_partials there)
  float t[6];
  const int nb = (int)gridDim.x;
  if (nb <= 64) {
    const int lane = threadIdx.x & 63;
    const float* p = a.partial + (size_t)(lane < nb ? lane : 0) * PPO_NPART;
    float q[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) q[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool mine = lane < nb;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = mine ? 0.f + q[k] : 0.f;
    t[4] = mine ? fmaxf(-3.4e38f, q[4]) : -3.4e38f;
    t[5] = mine ? fminf(3.4e38f, q[5]) : 3.4e38f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] += __shfl_xor(t[k], o, 64);
      t[4] = fmaxf(t[4], __shfl_xor(t[4], o, 64));
      t[5] = fminf(t[5], __shfl_xor(t[5], o, 64));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = 0.f + t[k];
    t[4] = fmaxf(-3.4e38f, t[4]);
    t[5] = fminf(3.4e38f, t[5]);
  } else {
    t[0] = t[1] = t[2] = t[3] = 0.f; t[4] = -3.4e38f; t[5] = 3.4e38f;
    for (int b = threadIdx.x; b < nb; b += 256) {
      const float* p = a.partial + (size_t)b * PPO_NPART;
      float q[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) q[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t[0] += q[0]; t[1] += q[1]; t[2] += q[2]; t[3] += q[3];
      t[4] = fmaxf(t[4], q[4]); t[5] = fminf(t[5], q[5]);
    }
    t[0] = jh_block_reduce(t[0], s_red, JhAdd(), 0.f);
    t[1] = jh_block_reduce(t[1], s_red, JhAdd(), 0.f);
    t[2] = jh_block_reduce(t[2], s_red, JhAdd(), 0.f);
    t[3] = jh_block_reduce(t[3], s_red, JhAdd(), 0.f);
    t[4] = jh_block_reduce(t[4], s_red, JhMax(), -3.4e38f);
    t[5] = jh_block_reduce(t[5], s_red, JhMin(), 3.4e38f);
  }
  if (threadIdx.x == 0) {
    float w1, w2;
    ppo_finish_stats(t[0], t[1], t[2], t[3], t[4], t[5], a.B, CONT ? a.B * a.A : a.B, a.vf, a.ent, w1, w2, a.stats);
    mix[0] = w1; mix[1] = w2;
    if (a.critic_sums) { a.critic_sums[0] = t[1]; a.critic_sums[1] = t[2]; }
  }
}

// Internal entry (jh_mlp.hip: jh_pponet_ppo_update_rows): heads and gradients as separate arrays, d_dv2 [B] + d_mix [2] + d_ticket (one zeroed word, left zero)
// + d_partial (>= 8 floats per 256 rows) from the caller.
int jh_ppo_loss_onepass(int continuous, int B, int A, const float* d_head0, const float* d_head1, const float* d_value_pred, const int64_t* d_idx,
                        const float* d_action, const float* d_adv, const float* d_ret, const float* d_value_old, const float* d_logp_old, float eps_clip,
                        float vf_coef, float ent_coef, float* d_g0, float* d_g1, float* d_gv, float* d_dv2, float* d_mix, unsigned* d_ticket, float* d_partial,
                        float* d_stats, hipStream_t st) {
  JH_ARG(B > 0 && A > 0 && d_head0 && d_value_pred && d_g0 && d_gv && d_dv2 && d_partial && (d_ticket ? d_mix != nullptr : B <= 64 * 256));
  const int nb = (B + 255) / 256;
  if (continuous) {
    PpoArgs<true> a{};
    a.B = B; a.A = A; a.h0 = d_head0; a.h1 = d_head1; a.value_pred = d_value_pred; a.ldh = A; a.ldvp = 1; a.idx = d_idx;
    a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
    a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_g0; a.g1 = d_g1; a.gv = d_gv; a.ldg = A; a.ldv = 1;
    a.stats = d_stats; a.defer_dv2 = d_dv2; a.partial = d_partial; a.nb = nb;
    JH_LAUNCH(jh_ppo_onepass_kernel<true>, dim3(nb), dim3(256), 0, st, a, d_ticket, d_mix);
  } else {
    PpoArgs<false> a{};
    a.B = B; a.A = A; a.h0 = d_head0; a.h1 = nullptr; a.value_pred = d_value_pred; a.ldh = A; a.ldvp = 1; a.idx = d_idx;
    a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
    a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_g0; a.g1 = nullptr; a.gv = d_gv; a.ldg = A; a.ldv = 1;
    a.stats = d_stats; a.defer_dv2 = d_dv2; a.partial = d_partial; a.nb = nb;
    JH_LAUNCH(jh_ppo_onepass_kernel<false>, dim3(nb), dim3(256), 0, st, a, d_ticket, d_mix);
  }
  JH_LAUNCH_CHECK();
  return JH_OK;
}

template <bool CONT>
static int ppo_launch(jh_ctx* ctx, PpoArgs<CONT>& a, hipStream_t st) {
  if (a.ldh == 0) a.ldh = a.A;
  if (a.ldvp == 0) a.ldvp = 1;
  if (a.B <= 1024) {
    const int threads = ((a.B + 63) / 64) * 64;
    a.nb = 1;
    const size_t lds = a.hpart ? sizeof(float) * 8 * (size_t)a.B : 0;
    if (threads <= 256) JH_LAUNCH((jh_ppo_fused_kernel<CONT, 256>), dim3(1), dim3(threads), lds, st, a);
    else JH_LAUNCH((jh_ppo_fused_kernel<CONT, 1024>), dim3(1), dim3(threads), lds, st, a);
    JH_LAUNCH_CHECK();
    return JH_OK;
  }
  a.nb = (a.B + 255) / 256;
  void* scratch = nullptr;
  int rc = jh_ctx_scratch(ctx, sizeof(float) * (PPO_NPART * (size_t)a.nb + 8), &scratch);
  if (rc) return rc;
  a.partial = (float*)scratch;
  JH_LAUNCH(jh_ppo_fwd_kernel<CONT>, dim3(a.nb), dim3(256), 0, st, a);
  JH_LAUNCH_CHECK();
  if (a.nb > 64) {  // a third launch (~5 us) beats nb^2 partial reads from here on
    float* totals = a.partial + PPO_NPART * (size_t)a.nb;
    JH_LAUNCH(jh_ppo_totals_kernel, dim3(1), dim3(256), 0, st, a.partial, a.nb, totals);
    JH_LAUNCH_CHECK();
    a.totals = totals;
  }
  JH_LAUNCH(jh_ppo_bwd_kernel<CONT>, dim3(a.nb), dim3(256), 0, st, a);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

JH_EXPORT int jh_ppo_loss_discrete(jh_ctx* ctx, int32_t B, int32_t A, const float* d_logits, const float* d_value_pred,
                                   const int64_t* d_idx, const float* d_action, const float* d_adv, const float* d_ret,
                                   const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef,
                                   float ent_coef, float* d_grad_logits, float* d_grad_value, float* d_stats,
                                   jh_stream stream) {
  JH_ARG(ctx && d_logits && d_value_pred && d_action && d_adv && d_ret && d_value_old && d_logp_old);
  JH_ARG(d_grad_logits && d_grad_value);
  JH_ARG(B > 0 && A > 0);
  PpoArgs<false> a{};
  a.B = B; a.A = A; a.h0 = d_logits; a.h1 = nullptr; a.value_pred = d_value_pred; a.idx = d_idx;
  a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
  a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_grad_logits; a.g1 = nullptr; a.gv = d_grad_value;
  a.ldg = A; a.ldv = 1;
  a.stats = d_stats;
  return ppo_launch<false>(ctx, a, jh_s(stream));
}

JH_EXPORT int jh_ppo_loss_continuous(jh_ctx* ctx, int32_t B, int32_t A, const float* d_mu_raw,
                                     const float* d_log_std_raw, const float* d_value_pred, const int64_t* d_idx,
                                     const float* d_action, const float* d_adv, const float* d_ret,
                                     const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef,
                                     float ent_coef, float* d_grad_mu_raw, float* d_grad_log_std_raw,
                                     float* d_grad_value, float* d_stats, jh_stream stream) {
  JH_ARG(ctx && d_mu_raw && d_log_std_raw && d_value_pred && d_action && d_adv && d_ret && d_value_old && d_logp_old);
  JH_ARG(d_grad_mu_raw && d_grad_log_std_raw && d_grad_value);
  JH_ARG(B > 0 && A > 0);
  PpoArgs<true> a{};
  a.B = B; a.A = A; a.h0 = d_mu_raw; a.h1 = d_log_std_raw; a.value_pred = d_value_pred; a.idx = d_idx;
  a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
  a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_grad_mu_raw; a.g1 = d_grad_log_std_raw; a.gv = d_grad_value;
  a.ldg = A; a.ldv = 1;
  a.stats = d_stats;
  return ppo_launch<true>(ctx, a, jh_s(stream));
}

// Exact critic for data-parallel learners, second half (see PpoArgs::defer_dv2).  sums = the ranks' {sum e1, sum e2} averaged over the
// ranks (one 8-byte all-reduce), B = rows per rank: c1 = sums[0] / B and c2 = sums[1] / B are then the means over the GLOBAL minibatch,
// every rank takes the same branch, and after the gradient all-reduce the update equals one learner's on the concatenated batch.
// stats_local: the loss kernel's row for this rank; stats_out gets it with the critic terms replaced by the global ones (written in the
// form ppo_finish_stats uses, jh_ppo_stats_row: the arrival markers of the mapped statistics are its [2] and [7]).
// pa.nranks > 1 (round 6, peer-pointer transport): the ranks' sums meet INSIDE this launch (jh_peer_small_exchange: mailboxes in the peers' arenas,
// one workgroup, so B <= 256) -- the separate 8-byte all-reduce launch between the loss and the backward is gone; `sums` then holds THIS rank's
// sums on entry and the ranks' mean on exit.
__global__ void __launch_bounds__(256) jh_ppo_critic_select_kernel(int B, float* __restrict__ sums, float vf, float ent, float* __restrict__ gv, int ldv,
                                                                   const float* __restrict__ dv2, const float* __restrict__ stats_local,
                                                                   float* __restrict__ stats_out, PeerArgs pa) {
  float c1, c2;
  if (pa.nranks > 1) {
    __shared__ float s_mean[2];
    const float m = jh_peer_small_exchange(pa, threadIdx.x < 2 ? sums[threadIdx.x] : 0.f, 2, 1.0f / (float)pa.nranks);
    if (threadIdx.x < 2) { s_mean[threadIdx.x] = m; sums[threadIdx.x] = m; }
    __syncthreads();
    c1 = s_mean[0] / (float)B;
    c2 = s_mean[1] / (float)B;
  } else {
    c1 = sums[0] / (float)B;
    c2 = sums[1] / (float)B;
  }
  const float w1 = c1 > c2 ? 1.f : (c1 == c2 ? 0.5f : 0.f), w2 = 1.f - w1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) gv[(size_t)i * ldv] = w1 * gv[(size_t)i * ldv] + w2 * dv2[i];
  if (i == 0 && stats_out) {
    const float critic = fmaxf(c1, c2);
    jh_ppo_stats_row(stats_out, stats_local[1] + vf * critic + ent * stats_local[3], stats_local[1], critic, stats_local[3], stats_local[4], stats_local[5], c1, c2);
  }
}

int jh_ppo_critic_select(int B, float* d_sums, float vf, float ent, float* d_gv, int ldv, const float* d_dv2, const float* d_stats_local,
                         float* d_stats_out, hipStream_t st, jh_peer* peer) {
  PeerArgs pa{};
  if (peer) {
    if (B > 256) return jh_fail(JH_ERR_ARG, "the critic select with the ranks' exchange inside is one workgroup: B <= 256 (got %d)", B);
    const int rc = jh_peer_args_for(peer, &pa);
    if (rc) return rc;
  }
  JH_LAUNCH(jh_ppo_critic_select_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, d_sums, vf, ent, d_gv, ldv, d_dv2, d_stats_local, d_stats_out, pa);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// Public form of the two halves for minibatches that do not take jh_pponet_ppo_update_dp_* (>= 1024 rows per rank): the loss with both
// critic branches kept (d_dv2 [B], d_critic_sums [2] = this rank's {sum e1, sum e2}, d_stats_local [8]), then -- after the caller's
// all-reduce (mean) of d_critic_sums -- jh_ppo_critic_select mixes d_grad_value in place and writes the final statistics row.
JH_EXPORT int jh_ppo_loss_deferred(jh_ctx* ctx, int32_t continuous, int32_t B, int32_t A, const float* d_head0, const float* d_head1,
                                   const float* d_value_pred, const int64_t* d_idx, const float* d_action, const float* d_adv, const float* d_ret,
                                   const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef, float ent_coef,
                                   float* d_grad_head0, float* d_grad_head1, float* d_grad_value, float* d_dv2, float* d_critic_sums,
                                   float* d_stats_local, jh_stream stream) {
  JH_ARG(ctx && d_head0 && d_value_pred && d_action && d_adv && d_ret && d_value_old && d_logp_old);
  JH_ARG(d_grad_head0 && d_grad_value && d_dv2 && d_critic_sums && B > 0 && A > 0 && (!continuous || (d_head1 && d_grad_head1)));
  if (continuous) {
    PpoArgs<true> a{};
    a.B = B; a.A = A; a.h0 = d_head0; a.h1 = d_head1; a.value_pred = d_value_pred; a.idx = d_idx;
    a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
    a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_grad_head0; a.g1 = d_grad_head1; a.gv = d_grad_value;
    a.ldg = A; a.ldv = 1; a.stats = d_stats_local; a.defer_dv2 = d_dv2; a.critic_sums = d_critic_sums;
    return ppo_launch<true>(ctx, a, jh_s(stream));
  }
  PpoArgs<false> a{};
  a.B = B; a.A = A; a.h0 = d_head0; a.h1 = nullptr; a.value_pred = d_value_pred; a.idx = d_idx;
  a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
  a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_grad_head0; a.g1 = nullptr; a.gv = d_grad_value;
  a.ldg = A; a.ldv = 1; a.stats = d_stats_local; a.defer_dv2 = d_dv2; a.critic_sums = d_critic_sums;
  return ppo_launch<false>(ctx, a, jh_s(stream));
}

// The heads in ONE row-major matrix, as a network whose last layer stacks them leaves them (jh_rbnet kind q with A + 1 | 2 A + 1 output rows: the policy-
// value net on the CNN head, policy_value.py:8-22 | 38-57): d_heads [B][ld] = (head0 [A] | head1 [A] (continuous) | value | padding), and the gradient goes back
// in the same layout (d_grad_heads [B][ld]; padding columns are not written).  d_dv2 / d_critic_sums / non-null: the deferred critic of jh_ppo_loss_deferred.
JH_EXPORT int jh_ppo_loss_packed(jh_ctx* ctx, int32_t continuous, int32_t B, int32_t A, const float* d_heads, int32_t ld, const int64_t* d_idx,
                                 const float* d_action, const float* d_adv, const float* d_ret, const float* d_value_old, const float* d_logp_old,
                                 float eps_clip, float vf_coef, float ent_coef, float* d_grad_heads, float* d_dv2, float* d_critic_sums, float* d_stats,
                                 jh_stream stream) {
  JH_ARG(ctx && d_heads && d_action && d_adv && d_ret && d_value_old && d_logp_old && d_grad_heads);
  JH_ARG(B > 0 && A > 0 && ld >= (continuous ? 2 * A + 1 : A + 1) && (!d_dv2 == !d_critic_sums));
  const int nv = continuous ? 2 * A : A;
  if (continuous) {
    PpoArgs<true> a{};
    a.B = B; a.A = A; a.h0 = d_heads; a.h1 = d_heads + A; a.value_pred = d_heads + nv; a.ldh = ld; a.ldvp = ld; a.idx = d_idx;
    a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
    a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_grad_heads; a.g1 = d_grad_heads + A; a.gv = d_grad_heads + nv;
    a.ldg = ld; a.ldv = ld; a.stats = d_stats; a.defer_dv2 = d_dv2; a.critic_sums = d_critic_sums;
    return ppo_launch<true>(ctx, a, jh_s(stream));
  }
  PpoArgs<false> a{};
  a.B = B; a.A = A; a.h0 = d_heads; a.h1 = nullptr; a.value_pred = d_heads + nv; a.ldh = ld; a.ldvp = ld; a.idx = d_idx;
  a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old; a.logp_old = d_logp_old;
  a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef; a.g0 = d_grad_heads; a.g1 = nullptr; a.gv = d_grad_heads + nv;
  a.ldg = ld; a.ldv = ld; a.stats = d_stats; a.defer_dv2 = d_dv2; a.critic_sums = d_critic_sums;
  return ppo_launch<false>(ctx, a, jh_s(stream));
}

// ... and apart again for the kernels that want them apart (log pi_old, GAE: once per learn() over the whole rollout): d_h0 [rows][A], d_h1 [rows][A] or null,
// d_value [rows].
__global__ void __launch_bounds__(256) jh_heads_unpack_kernel(int64_t rows, int A, int n_heads, const float* __restrict__ packed, int ld, float* __restrict__ h0,
                                                              float* __restrict__ h1, float* __restrict__ value) {
  const int w = n_heads * A + 1;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * w) return;
  const int64_t r = i / w;
  const int c = (int)(i - r * w);
  const float x = packed[r * ld + c];
  if (c < A) h0[r * A + c] = x;
  else if (c < n_heads * A) h1[r * A + (c - A)] = x;
  else value[r] = x;
}
JH_EXPORT int jh_heads_unpack(jh_ctx* ctx, int64_t rows, int32_t A, const float* d_packed, int32_t ld, float* d_h0, float* d_h1, float* d_value, jh_stream stream) {
  JH_ARG(ctx && d_packed && d_h0 && d_value && rows > 0 && A > 0);
  const int n_heads = d_h1 ? 2 : 1;
  JH_ARG(ld >= n_heads * A + 1);
  const int64_t tot = rows * (n_heads * A + 1);
  JH_LAUNCH(jh_heads_unpack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, jh_s(stream), rows, A, n_heads, d_packed, ld, d_h0, d_h1, d_value);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// PPO.act for a discrete policy whose heads are already on the device (ppo.py:63-69): Categorical(softmax(z)).sample() by inverse CDF on the SAME
// counter-based stream as the host-side sampler of the MLP policy (jh_common.h: jh_sample_discrete; key = seed, timestep counter, row), argmax (first
// maximum, torch.argmax) when !training.  One thread per row; d_action [W] int64 (device or device-mapped host memory).
__device__ __forceinline__ static uint64_t jh_mix64_dev(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void __launch_bounds__(64) jh_policy_act_discrete_kernel(int W, int A, const float* __restrict__ heads, int ld, uint64_t seed, uint64_t ctr, int training,
                                                                    int64_t* __restrict__ action) {
  const int w = blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  const float* z = heads + (size_t)w * ld;
  int act = 0;
  float mx = z[0];
  for (int k = 1; k < A; ++k)
    if (z[k] > mx) { mx = z[k]; act = k; }
  if (training) {
    float se = 0.f;
    for (int k = 0; k < A; ++k) se += expf(z[k] - mx);
    const double u01 = (double)(jh_mix64_dev(seed * 0x100000001B3ull + ctr * 0x9E3779B97F4A7C15ull + (uint64_t)w) >> 11) * (1.0 / 9007199254740992.0);
    const float u = (float)u01 * se;
    float c = 0.f;
    act = A - 1;
    for (int k = 0; k < A; ++k) {
      c += expf(z[k] - mx);
      if (u < c) { act = k; break; }
    }
  }
  action[w] = act;
}
JH_EXPORT int jh_policy_act_discrete(jh_ctx* ctx, int32_t W, int32_t A, const float* d_heads, int32_t ld, uint64_t seed, uint64_t counter, int32_t training,
                                     int64_t* d_action, jh_stream stream) {
  JH_ARG(ctx && d_heads && d_action && W > 0 && A > 0 && ld >= A);
  JH_LAUNCH(jh_policy_act_discrete_kernel, dim3((W + 63) / 64), dim3(64), 0, jh_s(stream), W, A, d_heads, ld, seed, counter, training, d_action);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

JH_EXPORT int jh_ppo_critic_select_rows(jh_ctx* ctx, int32_t B, const float* d_critic_sums, float vf_coef, float ent_coef, float* d_grad_value,
                                        const float* d_dv2, const float* d_stats_local, float* d_stats, jh_stream stream) {
  JH_ARG(ctx && B > 0 && d_critic_sums && d_grad_value && d_dv2 && d_stats_local);
  return jh_ppo_critic_select(B, const_cast<float*>(d_critic_sums), vf_coef, ent_coef, d_grad_value, 1, d_dv2, d_stats_local, d_stats, jh_s(stream), nullptr);
}

// ... for a gradient that sits in a packed [B][ld] matrix (jh_ppo_loss_packed): d_grad_value = its value column, ldv = ld
JH_EXPORT int jh_ppo_critic_select_strided(jh_ctx* ctx, int32_t B, const float* d_critic_sums, float vf_coef, float ent_coef, float* d_grad_value, int32_t ldv,
                                           const float* d_dv2, const float* d_stats_local, float* d_stats, jh_stream stream) {
  JH_ARG(ctx && B > 0 && ldv > 0 && d_critic_sums && d_grad_value && d_dv2 && d_stats_local);
  return jh_ppo_critic_select(B, const_cast<float*>(d_critic_sums), vf_coef, ent_coef, d_grad_value, ldv, d_dv2, d_stats_local, d_stats, jh_s(stream), nullptr);
}

// Internal entry (jh_mlp.hip): same losses with the heads given as encoder partial sums (jh_pmb_fwd_kernel) and
// the head gradients written PACKED, d_g_all [B][8] = (d head0 [A] | d head1 [A] (continuous) | d value | zeros):
// the operand layout of the backward grid (jh_pmb_bwd_kernel).
int jh_ppo_loss_from_partials(jh_ctx* ctx, int continuous, int B, int A, const float* d_hpart, int tiles, int part_rows, int part_ld,
                              const int64_t* d_idx, const float* d_action, const float* d_adv, const float* d_ret,
                              const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef, float ent_coef,
                              float* d_g_all, float* d_stats, float* d_hyper_advance, float* d_defer_dv2, float* d_critic_sums, hipStream_t st) {
  JH_ARG(B > 0 && B <= 1024 && d_hpart && d_g_all);
  if (continuous) {
    PpoArgs<true> a{};
    a.B = B; a.A = A; a.idx = d_idx; a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old;
    a.logp_old = d_logp_old; a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef;
    a.g0 = d_g_all; a.g1 = d_g_all + A; a.gv = d_g_all + 2 * A; a.ldg = 8; a.ldv = 8;
    a.stats = d_stats; a.hpart = d_hpart; a.hp_tiles = tiles; a.hp_rows = part_rows; a.hp_ld = part_ld; a.hyper_advance = d_hyper_advance;
    a.defer_dv2 = d_defer_dv2; a.critic_sums = d_critic_sums;
    return ppo_launch<true>(ctx, a, st);
  }
  PpoArgs<false> a{};
  a.B = B; a.A = A; a.idx = d_idx; a.action = d_action; a.adv = d_adv; a.ret = d_ret; a.value_old = d_value_old;
  a.logp_old = d_logp_old; a.eps = eps_clip; a.vf = vf_coef; a.ent = ent_coef;
  a.g0 = d_g_all; a.g1 = nullptr; a.gv = d_g_all + A; a.ldg = 8; a.ldv = 8;
  a.stats = d_stats; a.hpart = d_hpart; a.hp_tiles = tiles; a.hp_rows = part_rows; a.hp_ld = part_ld; a.hyper_advance = d_hyper_advance;
  a.defer_dv2 = d_defer_dv2; a.critic_sums = d_critic_sums;
  return ppo_launch<false>(ctx, a, st);
}
