// Native policy-value MLP encoder for the PPO hot path on gfx950 (a23 in SURVEY.md §8a):
//   S -> H (relu) -> H (relu) -> {A logits | A mu, A log_std} + 1 value
// (core/network/head.py:6-18 + policy_value.py:8-57), forward, backward, global-norm clip and Adam,
// all on flat fp32 parameter / gradient / moment buckets laid out in the reference's state_dict
// order so one RCCL all-reduce covers the whole gradient.
//
// The two H x H contractions per direction are the only dense work of the path; they run on the
// fp32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32, 157 TFLOP/s chip peak).  At the BASELINE
// shape (minibatch 256, H = 512: 0.13 GFLOP per GEMM) a GEMM is ~1 us of math, so the design goal is
// launch count and latency, not tile efficiency: every wave owns one 16x(16*TN) output tile and
// streams its operands straight from L2 into MFMA fragments (no LDS round trip, no inter-wave
// sync), 4 independent waves per workgroup, M/16 * N/(16*TN) waves in flight.
//
// K-permutation trick: a lane loads 4 consecutive k of its A row / B column as one 16-byte load
// and feeds element j of both to MFMA step j.  Step j therefore contracts k = 16t + 4*(lane>>4) + j
// -- a permutation of the K index that is identical for A and B, so the sum is unchanged and every
// global load is 16 B wide.
#include "jh_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct jh_pponet {
  jh_ctx* ctx = nullptr;
  int S = 0, H = 0, A = 0, cont = 0, max_rows = 0;
  int64_t n_params = 0;
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;  // borrowed flat buckets
  // offsets into the flat buckets (state_dict order)
  int64_t o_w1, o_b1, o_w2, o_b2, o_wh0, o_bh0, o_wh1, o_bh1, o_wv, o_bv;
  // owned workspaces
  float *h1 = nullptr, *h2 = nullptr, *dh1 = nullptr, *dh2 = nullptr;
  float* norm_partial = nullptr;  // [kNormBlocks]
  float* hyper = nullptr;         // device: {lr, beta1, beta2, eps, step, bc1, bc2_sqrt, _}
  unsigned long long* rng = nullptr;  // device: acting RNG counter
};

namespace {
constexpr int kNormBlocks = 256;
}

// ============================================================================ layer 1 (K = S, tiny)
// h1[b][j] = relu(sum_s x[r(b)][s] * W1[j][s] + b1[j]);  one lane per (b, j), j fastest.
__global__ void __launch_bounds__(256) jh_mlp_l1_kernel(int B, int S, int H, const float* __restrict__ x,
                                                        const int64_t* __restrict__ idx, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, float* __restrict__ h1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), j = (int)(i - (int64_t)b * H);
  const int64_t r = idx ? idx[b] : (int64_t)b;
  const float* xr = x + r * S;
  const float* w = W1 + (size_t)j * S;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc = fmaf(xr[s], w[s], acc);
  acc += b1[j];
  h1[i] = acc > 0.f ? acc : 0.f;
}

// ============================================================================ MFMA GEMM
// C[M][N] = A(m,k) * B(k,n), epilogue EPI.  Layout flags:
//   A_KCONT: A stored [M][K] (k contiguous)  else stored [K][M] (m contiguous)
//   B_KCONT: B stored [N][K] (k contiguous)  else stored [K][N] (n contiguous)
// One wave per 16 x (16*TN) tile, 4 waves per workgroup laid out along N.
enum { EPI_BIAS_RELU = 0, EPI_MASK = 1, EPI_NONE = 2 };

template <bool A_KCONT, bool B_KCONT, int TN, int EPI>
__global__ void __launch_bounds__(256) jh_gemm16_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                        const float* __restrict__ Bm, int ldb, float* __restrict__ C,
                                                        int ldc, const float* __restrict__ aux, int ldaux) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tiles_n = (N + 16 * TN - 1) / (16 * TN);
  const int tile = blockIdx.x * 4 + wid;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 16, n0 = tn * 16 * TN;
  if (m0 >= M) return;
  const int r = lane & 15, kq = lane >> 4;
  f32x4 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool m_ok = (m0 + r) < M;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int kb = k0 + 4 * kq;
    float a[4];
    if (A_KCONT) {
      if (m_ok && kb + 3 < K) {
        const float4 v = *reinterpret_cast<const float4*>(A + (size_t)(m0 + r) * lda + kb);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = (m_ok && kb + j < K) ? A[(size_t)(m0 + r) * lda + kb + j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = (m_ok && kb + j < K) ? A[(size_t)(kb + j) * lda + m0 + r] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int n = n0 + 16 * t + r;
      const bool n_ok = n < N;
      float b[4];
      if (B_KCONT) {
        if (n_ok && kb + 3 < K) {
          const float4 v = *reinterpret_cast<const float4*>(Bm + (size_t)n * ldb + kb);
          b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = (n_ok && kb + j < K) ? Bm[(size_t)n * ldb + kb + j] : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (n_ok && kb + j < K) ? Bm[(size_t)(kb + j) * ldb + n] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[t], 0, 0, 0);
    }
  }
  // C/D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = n0 + 16 * t + r;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + kq * 4 + i;
      if (m >= M) continue;
      float v = acc[t][i];
      if (EPI == EPI_BIAS_RELU) {
        v += aux[n];
        v = v > 0.f ? v : 0.f;
      } else if (EPI == EPI_MASK) {
        v = aux[(size_t)m * ldaux + n] > 0.f ? v : 0.f;  // relu'(h) of the forward activation
      }
      C[(size_t)m * ldc + n] = v;
    }
  }
}

template <bool A_KCONT, bool B_KCONT, int TN, int EPI>
static int launch_gemm(const char* name, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                       const float* aux, int ldaux, hipStream_t st) {
  const int tiles = ((M + 15) / 16) * ((N + 16 * TN - 1) / (16 * TN));
  JH_LAUNCH_NAMED(name, (jh_gemm16_kernel<A_KCONT, B_KCONT, TN, EPI>), dim3((tiles + 3) / 4), dim3(256), 0, st, M, N, K, A,
                     lda, B, ldb, C, ldc, aux, ldaux);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// ============================================================================ heads (N = A+1 or 2A+1, tiny)
struct HeadPtrs {
  const float* w[3];  // weight rows [n_i][H]
  const float* b[3];
  float* out[3];      // [B][n_i]
  const float* g[3];  // upstream grads (backward)
  float* dw[3];
  float* db[3];
  int n[3];
  int groups;
  // flattened view over all head outputs (<= 8) for kernels that keep one accumulator per output
  const float* fg[8];  // upstream grad column base (element [b] at fg[o][b * fld[o]])
  int fld[8];
  float* fdw[8];       // weight-grad row of output o
  float* fdb[8];
  int n_out;
};

// forward: one wave per row; lanes split H, one shuffle reduction per output
__global__ void __launch_bounds__(256) jh_mlp_heads_fwd_kernel(int B, int H, const float* __restrict__ h2, HeadPtrs hp) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* hr = h2 + (size_t)b * H;
  for (int g = 0; g < hp.groups; ++g) {
    for (int o = 0; o < hp.n[g]; ++o) {
      const float* w = hp.w[g] + (size_t)o * H;
      float acc = 0.f;
      for (int k = lane * 4; k < H; k += 256) {
        const float4 hv = *reinterpret_cast<const float4*>(hr + k);
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
        acc = fmaf(hv.x, wv.x, acc);
        acc = fmaf(hv.y, wv.y, acc);
        acc = fmaf(hv.z, wv.z, acc);
        acc = fmaf(hv.w, wv.w, acc);
      }
      acc = jh_wave_sum(acc);
      if (lane == 0) hp.out[g][(size_t)b * hp.n[g] + o] = acc + hp.b[g][o];
    }
  }
}

// backward part 1: dh2[b][k] = relu'(h2[b][k]) * sum_o g[b][o] * Wh[o][k]
__global__ void __launch_bounds__(256) jh_mlp_heads_bwd_dh_kernel(int B, int H, const float* __restrict__ h2,
                                                                  float* __restrict__ dh2, HeadPtrs hp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), k = (int)(i - (int64_t)b * H);
  float acc = 0.f;
  for (int g = 0; g < hp.groups; ++g)
    for (int o = 0; o < hp.n[g]; ++o) acc = fmaf(hp.g[g][(size_t)b * hp.n[g] + o], hp.w[g][(size_t)o * H + k], acc);
  dh2[i] = h2[i] > 0.f ? acc : 0.f;
}

// backward part 2: head weight/bias grads + the column sums that are the bias grads of layers 1, 2:
//   dWh[o][k] = sum_b g[b][o] h2[b][k] ; dbh[o] = sum_b g[b][o] ; db2[k] = sum_b dh2[b][k]
// grid: one workgroup per 64 columns k; 256 threads = 64 columns x 4 batch slices, LDS combine.
__global__ void __launch_bounds__(256) jh_mlp_heads_bwd_dw_kernel(int B, int H, const float* __restrict__ h2,
                                                                  const float* __restrict__ dh2, float* __restrict__ db2,
                                                                  HeadPtrs hp) {
  __shared__ float s_acc[4][64][9];  // up to 8 head outputs + db2
  const int kc = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kc;
  float acc[9];
#pragma unroll
  for (int o = 0; o < 9; ++o) acc[o] = 0.f;
  if (k < H) {
    for (int b = sl; b < B; b += 4) {
      const float hv = h2[(size_t)b * H + k];
#pragma unroll
      for (int o = 0; o < 8; ++o)  // static indices: acc[] stays in registers
        if (o < hp.n_out) acc[o] = fmaf(hp.fg[o][(size_t)b * hp.fld[o]], hv, acc[o]);
      acc[8] += dh2[(size_t)b * H + k];
    }
  }
#pragma unroll
  for (int o = 0; o < 9; ++o) s_acc[sl][kc][o] = acc[o];
  __syncthreads();
  if (sl == 0 && k < H) {
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < hp.n_out) hp.fdw[o][k] = s_acc[0][kc][o] + s_acc[1][kc][o] + s_acc[2][kc][o] + s_acc[3][kc][o];
    db2[k] = s_acc[0][kc][8] + s_acc[1][kc][8] + s_acc[2][kc][8] + s_acc[3][kc][8];
  }
  // bias grads of the heads: workgroup 0, first wave
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    for (int o = 0; o < hp.n_out; ++o) {
      float sum = 0.f;
      for (int b = threadIdx.x; b < B; b += 64) sum += hp.fg[o][(size_t)b * hp.fld[o]];
      sum = jh_wave_sum(sum);
      if (threadIdx.x == 0) *hp.fdb[o] = sum;
    }
  }
}

// layer-1 backward: dW1[j][s] = sum_b dh1[b][j] x[r(b)][s] ; db1[j] = sum_b dh1[b][j]
// one workgroup per 64 hidden units j; 4 batch slices combined through LDS.
__global__ void __launch_bounds__(256) jh_mlp_l1_bwd_kernel(int B, int S, int H, const float* __restrict__ x,
                                                            const int64_t* __restrict__ idx,
                                                            const float* __restrict__ dh1, float* __restrict__ dW1,
                                                            float* __restrict__ db1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4][64][S+1]
  const int jc = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jc;
  float* mine = smem + ((size_t)sl * 64 + jc) * (S + 1);
  for (int s = 0; s <= S; ++s) mine[s] = 0.f;
  if (j < H) {
    for (int b = sl; b < B; b += 4) {
      const float d = dh1[(size_t)b * H + j];
      const int64_t r = idx ? idx[b] : (int64_t)b;
      const float* xr = x + r * S;
      for (int s = 0; s < S; ++s) mine[s] = fmaf(d, xr[s], mine[s]);
      mine[S] += d;
    }
  }
  __syncthreads();
  if (sl == 0 && j < H) {
    for (int s = 0; s <= S; ++s) {
      const float t = smem[((size_t)0 * 64 + jc) * (S + 1) + s] + smem[((size_t)1 * 64 + jc) * (S + 1) + s] +
                      smem[((size_t)2 * 64 + jc) * (S + 1) + s] + smem[((size_t)3 * 64 + jc) * (S + 1) + s];
      if (s < S) dW1[(size_t)j * S + s] = t;
      else db1[j] = t;
    }
  }
}

// ============================================================================ clip_grad_norm_ + Adam
// hyper (device): [0] lr [1] beta1 [2] beta2 [3] eps [4] step [5] 1-beta1^t [6] sqrt(1-beta2^t)
__global__ void __launch_bounds__(256) jh_gradnorm_kernel(int64_t n, const float* __restrict__ g,
                                                          float* __restrict__ partial, float* __restrict__ hyper) {
  __shared__ float s_red[16];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
  acc = jh_block_reduce(acc, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // advance Adam's step (nobody reads hyper in this kernel)
    const float t = hyper[4] + 1.f;
    hyper[4] = t;
    hyper[5] = 1.f - powf(hyper[1], t);
    hyper[6] = sqrtf(1.f - powf(hyper[2], t));
  }
}

__global__ void __launch_bounds__(256) jh_adam_kernel(int64_t n, float* __restrict__ p, float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ partial, int n_partial,
                                                      const float* __restrict__ hyper, float max_norm,
                                                      float* __restrict__ norm_out) {
  __shared__ float s_red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
  const float total = sqrtf(jh_block_reduce(acc, s_red, JhAdd(), 0.f));
  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total + 1e-6), clamped to 1
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[5], bc2s = hyper[6];
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    g[i] = gi;                                    // clip is in place, like the reference
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);  // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

// ============================================================================ acting
// softmax + inverse-CDF multinomial on the logits of W rows (W small): one lane per env.
// Counter-based RNG (splitmix64 of (seed, counter, env)); the counter lives in device memory so a
// captured graph can be replayed.
__device__ __forceinline__ float u01_from(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x = x ^ (x >> 31);
  return (float)(x >> 40) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(1024) jh_sample_discrete_kernel(int W, int A, const float* __restrict__ logits,
                                                                  unsigned long long* __restrict__ rng,
                                                                  int64_t* __restrict__ action, int greedy) {
  // ONE workgroup: every lane reads the counter, then lane 0 advances it after the barrier
  const unsigned long long ctr = rng[0], seed = rng[1];
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    const float* z = logits + (size_t)w * A;
    float mx = z[0];
    int arg = 0;
    for (int k = 1; k < A; ++k)
      if (z[k] > mx) { mx = z[k]; arg = k; }
    int a = arg;
    if (!greedy) {
      float se = 0.f;
      for (int k = 0; k < A; ++k) se += expf(z[k] - mx);
      const float u = u01_from(seed * 0x100000001B3ull + ctr * 0x9E3779B97F4A7C15ull + (unsigned long long)w) * se;
      float c = 0.f;
      a = A - 1;
      for (int k = 0; k < A; ++k) {
        c += expf(z[k] - mx);
        if (u < c) { a = k; break; }
      }
    }
    action[w] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) rng[0] = ctr + 1;
}

// ============================================================================ host API
static int64_t pponet_layout(jh_pponet* n) {
  const int64_t S = n->S, H = n->H, A = n->A;
  int64_t o = 0;
  n->o_w1 = o; o += H * S;
  n->o_b1 = o; o += H;
  n->o_w2 = o; o += H * H;
  n->o_b2 = o; o += H;
  n->o_wh0 = o; o += A * H;  // pi.weight | mu.weight
  n->o_bh0 = o; o += A;
  if (n->cont) {
    n->o_wh1 = o; o += A * H;  // log_std.weight
    n->o_bh1 = o; o += A;
  } else {
    n->o_wh1 = n->o_bh1 = -1;
  }
  n->o_wv = o; o += H;
  n->o_bv = o; o += 1;
  return o;
}

JH_EXPORT int64_t jh_pponet_param_count(int32_t S, int32_t H, int32_t A, int32_t continuous) {
  jh_pponet t;
  t.S = S; t.H = H; t.A = A; t.cont = continuous;
  return pponet_layout(&t);
}

JH_EXPORT int jh_pponet_create(jh_ctx* ctx, int32_t S, int32_t H, int32_t A, int32_t continuous, int32_t max_rows,
                               float* d_params, float* d_grads, float* d_m, float* d_v, uint64_t seed,
                               jh_pponet** out) {
  JH_ARG(ctx && out && d_params && d_grads && d_m && d_v);
  JH_ARG(S > 0 && A > 0 && max_rows > 0);
  JH_ARG(H >= 16 && H % 16 == 0);
  JH_ARG((continuous ? 2 * A + 1 : A + 1) <= 8);
  JH_HIP(hipSetDevice(ctx->device));
  jh_pponet* n = new jh_pponet();
  n->ctx = ctx; n->S = S; n->H = H; n->A = A; n->cont = continuous ? 1 : 0; n->max_rows = max_rows;
  n->n_params = pponet_layout(n);
  n->params = d_params; n->grads = d_grads; n->m = d_m; n->v = d_v;
  const size_t act = sizeof(float) * (size_t)max_rows * (size_t)H;
  JH_HIP(hipMalloc((void**)&n->h1, act));
  JH_HIP(hipMalloc((void**)&n->h2, act));
  JH_HIP(hipMalloc((void**)&n->dh1, act));
  JH_HIP(hipMalloc((void**)&n->dh2, act));
  JH_HIP(hipMalloc((void**)&n->norm_partial, sizeof(float) * kNormBlocks));
  JH_HIP(hipMalloc((void**)&n->hyper, sizeof(float) * 8));
  JH_HIP(hipMalloc((void**)&n->rng, sizeof(unsigned long long) * 2));
  const float hy[8] = {1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 0.f, 0.f, 0.f};
  JH_HIP(hipMemcpy(n->hyper, hy, sizeof(hy), hipMemcpyHostToDevice));
  const unsigned long long r[2] = {0ull, (unsigned long long)seed};
  JH_HIP(hipMemcpy(n->rng, r, sizeof(r), hipMemcpyHostToDevice));
  *out = n;
  return JH_OK;
}

JH_EXPORT void jh_pponet_destroy(jh_pponet* n) {
  if (!n) return;
  (void)hipSetDevice(n->ctx->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(n->h1); (void)hipFree(n->h2); (void)hipFree(n->dh1); (void)hipFree(n->dh2);
  (void)hipFree(n->norm_partial); (void)hipFree(n->hyper); (void)hipFree(n->rng);
  delete n;
}

JH_EXPORT int jh_pponet_set_hyper(jh_pponet* n, float lr, float beta1, float beta2, float eps, float step,
                                  jh_stream stream) {
  JH_ARG(n != nullptr);
  // staged through a pinned slab so the copy is a true async H2D that a later graph launch sees
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(n->ctx, 64, &slab);
  if (rc) return rc;
  float* h = (float*)slab->host;
  h[0] = lr; h[1] = beta1; h[2] = beta2; h[3] = eps; h[4] = step;
  JH_HIP(hipMemcpyAsync(n->hyper, h, sizeof(float) * (step >= 0.f ? 5 : 4), hipMemcpyHostToDevice, jh_s(stream)));
  return jh_ctx_slab_release(n->ctx, slab, jh_s(stream));
}

JH_EXPORT int jh_pponet_set_lr(jh_pponet* n, float lr, jh_stream stream) {
  JH_ARG(n != nullptr);
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(n->ctx, 64, &slab);
  if (rc) return rc;
  *(float*)slab->host = lr;
  JH_HIP(hipMemcpyAsync(n->hyper, slab->host, sizeof(float), hipMemcpyHostToDevice, jh_s(stream)));
  return jh_ctx_slab_release(n->ctx, slab, jh_s(stream));
}

static HeadPtrs head_ptrs(jh_pponet* n, float* out0, float* out1, float* outv, const float* g0, const float* g1,
                          const float* gv) {
  HeadPtrs hp{};
  int g = 0;
  hp.w[g] = n->params + n->o_wh0; hp.b[g] = n->params + n->o_bh0; hp.out[g] = out0; hp.g[g] = g0;
  hp.dw[g] = n->grads + n->o_wh0; hp.db[g] = n->grads + n->o_bh0; hp.n[g] = n->A; ++g;
  if (n->cont) {
    hp.w[g] = n->params + n->o_wh1; hp.b[g] = n->params + n->o_bh1; hp.out[g] = out1; hp.g[g] = g1;
    hp.dw[g] = n->grads + n->o_wh1; hp.db[g] = n->grads + n->o_bh1; hp.n[g] = n->A; ++g;
  }
  hp.w[g] = n->params + n->o_wv; hp.b[g] = n->params + n->o_bv; hp.out[g] = outv; hp.g[g] = gv;
  hp.dw[g] = n->grads + n->o_wv; hp.db[g] = n->grads + n->o_bv; hp.n[g] = 1; ++g;
  hp.groups = g;
  int o = 0;
  for (int q = 0; q < g; ++q)
    for (int c = 0; c < hp.n[q]; ++c, ++o) {
      hp.fg[o] = hp.g[q] ? hp.g[q] + c : nullptr;
      hp.fld[o] = hp.n[q];
      hp.fdw[o] = hp.dw[q] + (size_t)c * n->H;
      hp.fdb[o] = hp.db[q] + c;
    }
  hp.n_out = o;
  return hp;
}

// x: [*, S] rows (device, or pinned host memory mapped into the device address space), gathered
// through d_idx when given.  Activations h1/h2 stay in the net's workspace for a following backward.
JH_EXPORT int jh_pponet_forward(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, float* d_head0,
                                float* d_head1, float* d_value, jh_stream stream) {
  JH_ARG(n && d_x && d_head0 && d_value);
  JH_ARG(B > 0 && B <= n->max_rows);
  JH_ARG(!n->cont || d_head1);
  hipStream_t st = jh_s(stream);
  const int H = n->H;
  const int64_t bh = (int64_t)B * H;
  JH_LAUNCH(jh_mlp_l1_kernel, dim3((unsigned)((bh + 255) / 256)), dim3(256), 0, st, B, n->S, H, d_x, d_idx,
                     n->params + n->o_w1, n->params + n->o_b1, n->h1);
  JH_LAUNCH_CHECK();
  int rc = launch_gemm<true, true, 1, EPI_BIAS_RELU>("jh_gemm16_fwd_h2", B, H, H, n->h1, H, n->params + n->o_w2, H, n->h2, H,
                                                     n->params + n->o_b2, 0, st);
  if (rc) return rc;
  HeadPtrs hp = head_ptrs(n, d_head0, d_head1, d_value, nullptr, nullptr, nullptr);
  JH_LAUNCH(jh_mlp_heads_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, B, H, n->h2, hp);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// Backward of the LAST forward (same B, x, idx): overwrites the flat gradient bucket.
JH_EXPORT int jh_pponet_backward(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx,
                                 const float* d_g_head0, const float* d_g_head1, const float* d_g_value,
                                 jh_stream stream) {
  JH_ARG(n && d_x && d_g_head0 && d_g_value);
  JH_ARG(B > 0 && B <= n->max_rows);
  JH_ARG(!n->cont || d_g_head1);
  hipStream_t st = jh_s(stream);
  const int H = n->H, S = n->S;
  const int64_t bh = (int64_t)B * H;
  HeadPtrs hp = head_ptrs(n, nullptr, nullptr, nullptr, d_g_head0, d_g_head1, d_g_value);
  JH_LAUNCH(jh_mlp_heads_bwd_dh_kernel, dim3((unsigned)((bh + 255) / 256)), dim3(256), 0, st, B, H, n->h2,
                     n->dh2, hp);
  JH_LAUNCH_CHECK();
  JH_LAUNCH(jh_mlp_heads_bwd_dw_kernel, dim3((H + 63) / 64), dim3(256), 0, st, B, H, n->h2, n->dh2,
                     n->grads + n->o_b2, hp);
  JH_LAUNCH_CHECK();
  // dW2[o][i] = sum_b dh2[b][o] * h1[b][i]     (A = dh2^T: stored [K=B][M=H]; B = h1: stored [K=B][N=H])
  int rc = launch_gemm<false, false, 2, EPI_NONE>("jh_gemm16_bwd_dW2", H, H, B, n->dh2, H, n->h1, H, n->grads + n->o_w2, H, nullptr, 0, st);
  if (rc) return rc;
  // dh1[b][i] = relu'(h1) * sum_o dh2[b][o] * W2[o][i]   (A = dh2 [M=B][K=H]; B = W2 stored [K=H_out][N=H_in])
  rc = launch_gemm<true, false, 1, EPI_MASK>("jh_gemm16_bwd_dh1", B, H, H, n->dh2, H, n->params + n->o_w2, H, n->dh1, H, n->h1, H, st);
  if (rc) return rc;
  JH_LAUNCH(jh_mlp_l1_bwd_kernel, dim3((H + 63) / 64), dim3(256), sizeof(float) * 4 * 64 * (size_t)(S + 1), st,
                     B, S, H, d_x, d_idx, n->dh1, n->grads + n->o_w1, n->grads + n->o_b1);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// clip_grad_norm_(max_norm) (skipped when max_norm <= 0) + Adam step on the flat buckets.
// d_norm_out (optional, device float) receives the pre-clip global norm.
JH_EXPORT int jh_pponet_adam_step(jh_pponet* n, float max_norm, float* d_norm_out, jh_stream stream) {
  JH_ARG(n != nullptr);
  hipStream_t st = jh_s(stream);
  JH_LAUNCH(jh_gradnorm_kernel, dim3(kNormBlocks), dim3(256), 0, st, n->n_params, n->grads, n->norm_partial,
                     n->hyper);
  JH_LAUNCH_CHECK();
  JH_LAUNCH(jh_adam_kernel, dim3(kNormBlocks), dim3(256), 0, st, n->n_params, n->params, n->grads, n->m, n->v,
                     n->norm_partial, kNormBlocks, n->hyper, max_norm, d_norm_out);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// Batched acting for W envs: forward + softmax + multinomial (greedy when training == 0).
// d_obs / d_action may be pinned host memory mapped into the device address space.
JH_EXPORT int jh_pponet_act_discrete(jh_pponet* n, int32_t W, const float* d_obs, int64_t* d_action,
                                     float* d_logits_ws, float* d_value_ws, int32_t training, jh_stream stream) {
  JH_ARG(n && d_obs && d_action && d_logits_ws && d_value_ws);
  JH_ARG(!n->cont);
  int rc = jh_pponet_forward(n, W, d_obs, nullptr, d_logits_ws, nullptr, d_value_ws, stream);
  if (rc) return rc;
  const int threads = W >= 1024 ? 1024 : ((W + 63) / 64) * 64;
  JH_LAUNCH(jh_sample_discrete_kernel, dim3(1), dim3(threads), 0, jh_s(stream), W, n->A, d_logits_ws, n->rng,
                     d_action, training ? 0 : 1);
  JH_LAUNCH_CHECK();
  return JH_OK;
}
