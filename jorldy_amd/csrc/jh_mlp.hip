// Native policy-value MLP encoder for the PPO hot path on gfx950 (a23 in SURVEY.md §8a):
//   S -> H (relu) -> H (relu) -> {A logits | A mu, A log_std} + 1 value
// (core/network/head.py:6-18 + policy_value.py:8-57): forward, backward, global-norm clip, Adam and
// batched acting, on flat fp32 parameter / gradient / moment buckets laid out in the reference's
// state_dict order so one RCCL all-reduce covers the whole gradient.
//
// Every contraction with a dimension >= 16 runs on the fp32-input MFMA (v_mfma_f32_16x16x4_f32:
// exact fp32, 157 TFLOP/s chip peak).  At the BASELINE shape (minibatch 256, H = 512) one GEMM is
// 0.13 GFLOP = ~1 us of math, and its operands were just rewritten by Adam (L2-cold: every access
// pays ~0.4 us), so the kernel is built for LATENCY, not tile efficiency:
//   * one workgroup per 16x16 output tile, its 4 waves split K (in-workgroup split-K, LDS combine
//     in fixed wave order -> deterministic): M/16 * N/16 * 4 waves in flight;
//   * a wave issues ALL loads of a batch of U k-chunks (2*U 16-byte loads per lane) before its
//     first MFMA: one memory round trip per batch instead of one per chunk;
//   * K-permutation: a lane loads 4 consecutive k of its A row / B column with one 16-byte load and
//     feeds element j of both to MFMA step j, i.e. step j contracts k = 16t + 4*(lane>>4) + j -- the
//     same permutation of K for A and B, so the sum is unchanged and every load is 16 B wide;
//   * bias gradients (column sums of the upstream gradient) fall out of the A fragments as row
//     sums, so they cost no extra pass.
#include "jh_ppo_mb.h"
#include "jh_tgemm.h"
#include "jh_ppo_finish.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kNormBlocks = 256;
constexpr int kNormSlots = kNormBlocks + 256;  // norm_partial: the norm kernel's 256 sums, or the folded form's (jh_ppo_dw1_partial_kernel: NormJob)
constexpr int kMaxHeadOutputs = 40;
}

// ============================================================================ layer 1 (K = S, tiny)
// h1[b][j] = relu(sum_s x[r(b)][s] * W1[j][s] + b1[j]).  A lane owns one hidden unit j for kL1Rows rows: its W1 row (S <= 16 floats,
// 44-byte stride between lanes at S = 11) is fetched ONCE into registers; the workgroup's kL1Rows observation rows are gathered into
// LDS first (index -> row: two dependent round trips per workgroup instead of per row), the store is coalesced.  One lane per (b, j)
// re-read the strided W1 row for every output: 13.5-19 us at B = 2048, S = 11 (4 MB of h1 written; a tenth of that is the store).
constexpr int kL1Rows = 16;
__global__ void __launch_bounds__(256) jh_mlp_l1_kernel(int B, int S, int H, const float* __restrict__ x,
                                                        const int64_t* __restrict__ idx, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, float* __restrict__ h1) {
  __shared__ __attribute__((aligned(16))) float sx[kL1Rows][16];
  const int jb = (H + 255) / 256;  // column blocks of 256 hidden units
  const int rb = blockIdx.x / jb, j = (blockIdx.x - rb * jb) * 256 + threadIdx.x;
  const int b0 = rb * kL1Rows, nb = b0 + kL1Rows < B ? kL1Rows : B - b0;
  const bool lds_rows = S <= 16;
  if (lds_rows) {
    const int rr = threadIdx.x >> 4, sc = threadIdx.x & 15;  // 256 lanes = 16 rows x 16 columns
    float v = 0.f;
    if (rr < nb && sc < S) {
      const int64_t r = idx ? idx[b0 + rr] : (int64_t)(b0 + rr);
      v = x[r * S + sc];
    }
    sx[rr][sc] = v;
  }
  const int jc = j < H ? j : H - 1;
  float w[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) w[s] = s < S ? W1[(size_t)jc * S + s] : 0.f;
  const float bias = b1[jc];
  __syncthreads();
  if (j >= H) return;
  // Every fetch above has LANDED before the row loop (round 5, tools/isa_chain.py): left pending, hipcc's wait for the bias sat
  // INSIDE the loop as s_waitcnt vmcnt(0) -- which also waits for the previous row's STORE: sixteen stores, each acknowledged before
  // the next was issued (14.5 us per launch for 4 MB of h1 at B = 2048; the builtin clears the compiler's own scoreboard).
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
  if (lds_rows) {  // (its own loop: with the S > 16 branch's fetches in the same loop body hipcc keeps a vmcnt(0) in front of every store)
    for (int rr = 0; rr < nb; ++rr) {
      // the row as four 16-byte LDS reads, all 16 taps unconditionally: columns >= S hold 0 in sx AND in w, and fmaf(0, 0, acc) == acc
      // (acc is never -0: it starts at +0), so the chain over s < S is unchanged
      const float4* xr = reinterpret_cast<const float4*>(&sx[rr][0]);
      const float4 x0 = xr[0], x1 = xr[1], x2 = xr[2], x3 = xr[3];
      const float xs[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = fmaf(xs[s], w[s], acc);
      acc += bias;
      h1[(size_t)(b0 + rr) * H + j] = acc > 0.f ? acc : 0.f;
    }
    return;
  }
  for (int rr = 0; rr < nb; ++rr) {
    float acc = 0.f;
    const int64_t r = idx ? idx[b0 + rr] : (int64_t)(b0 + rr);
    for (int s = 0; s < S; ++s) acc = fmaf(x[r * S + s], W1[(size_t)j * S + s], acc);
    acc += bias;
    h1[(size_t)(b0 + rr) * H + j] = acc > 0.f ? acc : 0.f;
  }
}

// ============================================================================ MFMA GEMM
// C[M][N] = A(m,k) * B(k,n).
//   A_MODE 0: A stored [M][K] (k contiguous)           1: A stored [K][M] (m contiguous)
//          2: A(m,k) = relu(b1[k] + sum_s x[r(m)][s] * W1[k][s])  (layer 1 generated on the fly)
//   B_KCONT : B stored [N][K] (k contiguous)  else stored [K][N] (n contiguous, optional row gather)
// Workgroup tile = (16*TM) x (16*TN): operands that are contiguous along m / n are read in
// 64*TM / 64*TN byte segments, so TM = TN = 2 uses whole 128-byte lines.
enum { EPI_BIAS_RELU = 0, EPI_MASK = 1, EPI_NONE = 2, EPI_ROWPTR = 3, EPI_HEADPART = 4 };

struct GemmArgs {
  int M, N, K;
  const float* A;
  int lda;
  const float* B;
  int ldb;
  const int64_t* b_rows;  // !B_KCONT: B(k,n) = B[b_rows[k]*ldb + n] when non-null
  float* C;
  int ldc;
  const float* aux;  // bias[n] (BIAS_RELU / HEADPART) | forward activation for the relu mask (MASK)
  int ldaux;
  float* rowptr[8];  // ROWPTR: row m of C lives at rowptr[m] (+ n); M <= 8
  float* rowsum;     // ROWSUM: rowsum[m] = sum_k A(m,k)   (bias gradients)
  float* rowsum_ptr[8];
  // A_MODE 2
  const float* x;
  const int64_t* x_rows;
  const float* W1;
  const float* b1;
  int S;
  // HEADPART: per-tile partial head outputs part[tile_n][row][8] = sum_{n in tile} h2[row][n]*Wh[o][n]
  const float* wh[8];
  int n_out;
  float* part;           // may be device-mapped pinned HOST memory (acting: the host finishes the heads)
  int part_rows;
  const float* hbias[8];  // bias of head output o, added by column-tile 0
  unsigned* tile_flag;   // optional [tiles] (device-mapped pinned host words): set to flag_seq per tile
  unsigned flag_seq;
};

template <int A_MODE, bool B_KCONT, int EPI, bool ROWSUM, int U, int TM, int TN>
__global__ void __launch_bounds__(256) jh_gemm16_kernel(GemmArgs g) {
  __shared__ float s_acc[4][TM * TN][64][4];
  __shared__ float s_rs[4][TM][64];
  // A_MODE 2: the 16 x K slice of layer 1 this tile needs, generated once per workgroup
  // (dynamic LDS: 16*S floats of observations + 16*(K+4) floats of h1; +4 keeps rows 16 B aligned
  // and breaks the power-of-two row stride)
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tiles_n = (g.N + 16 * TN - 1) / (16 * TN);
  const int tm_blk = blockIdx.x / tiles_n, tn_blk = blockIdx.x - tm_blk * tiles_n;
  const int m0 = tm_blk * 16 * TM, n0 = tn_blk * 16 * TN;
  const int r = lane & 15, kq = lane >> 4;
  // this wave's K range (multiple of 16 long)
  const int kper = ((g.K + 63) / 64) * 16;
  const int kbeg = wid * kper;
  const int kend = kbeg + kper < g.K ? kbeg + kper : g.K;
  bool m_ok[TM], n_ok[TN];
  int mc[TM], nc[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = m0 + 16 * t + r;
    m_ok[t] = m < g.M;
    mc[t] = m_ok[t] ? m : g.M - 1;  // clamped: loads stay in bounds
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = n0 + 16 * t + r;
    n_ok[t] = n < g.N;
    nc[t] = n_ok[t] ? n : g.N - 1;
  }
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = 0.f;
  float a[U][TM][4], b[U][TN][4];

  auto load_b = [&](int k0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = k0 + 16 * u + 4 * kq;
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        if (B_KCONT) {
          const int kc = kb + 3 < g.K ? kb : (g.K >= 4 ? g.K - 4 : 0);
          const float4 v = *reinterpret_cast<const float4*>(g.B + (size_t)nc[t] * g.ldb + kc);
          const bool ok = n_ok[t] && kb < kend;  // K % 4 == 0 for k-contiguous operands
          b[u][t][0] = ok ? v.x : 0.f; b[u][t][1] = ok ? v.y : 0.f; b[u][t][2] = ok ? v.z : 0.f; b[u][t][3] = ok ? v.w : 0.f;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = kb + j;
            const int kc = k < g.K ? k : g.K - 1;
            const int64_t row = g.b_rows ? g.b_rows[kc] : (int64_t)kc;
            const float v = g.B[(size_t)row * g.ldb + nc[t]];
            b[u][t][j] = (n_ok[t] && k < kend) ? v : 0.f;
          }
        }
      }
    }
  };
  auto load_a = [&](int k0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = k0 + 16 * u + 4 * kq;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (A_MODE == 0 || A_MODE == 2) {
          const int kc = kb + 3 < g.K ? kb : (g.K >= 4 ? g.K - 4 : 0);
          const float4 v = A_MODE == 0 ? *reinterpret_cast<const float4*>(g.A + (size_t)mc[t] * g.lda + kc)
                                       : *reinterpret_cast<const float4*>(s_dyn + 16 * g.S + r * (g.K + 4) + kc);
          const bool ok = m_ok[t] && kb < kend;
          a[u][t][0] = ok ? v.x : 0.f; a[u][t][1] = ok ? v.y : 0.f; a[u][t][2] = ok ? v.z : 0.f; a[u][t][3] = ok ? v.w : 0.f;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = kb + j;
            const int kc = k < g.K ? k : g.K - 1;
            const float v = g.A[(size_t)kc * g.lda + mc[t]];
            a[u][t][j] = (m_ok[t] && k < kend) ? v : 0.f;
          }
        }
      }
    }
  };

  if (A_MODE == 2) {
    // The weight loads of the first (normally only) batch do not depend on the observations: put
    // them in flight before the PCIe round trip below.
    load_b(kbeg);
    float* xs = s_dyn;              // [16][S]
    float* h1s = s_dyn + 16 * g.S;  // [16][K + 4]
    const int ldh = g.K + 4;
    // observations may live in device-mapped HOST memory (uncached, a PCIe round trip per access):
    // read every element exactly once per workgroup
    for (int i = threadIdx.x; i < 16 * g.S; i += 256) {
      const int rr = i / g.S, q = i - rr * g.S;
      const int mr = m0 + rr < g.M ? m0 + rr : g.M - 1;
      const int64_t xrow = g.x_rows ? g.x_rows[mr] : (int64_t)mr;
      xs[i] = g.x[xrow * g.S + q];
    }
    __syncthreads();
    // layer 1 ON THE MFMA: h1[16 x K] = x[16 x S] * W1^T[S x K]; wave w owns hidden units
    // [w*K/4, (w+1)*K/4) in tiles of 16 (one 16x16x4 MFMA per tile when S <= 4).  The VALU form of this
    // (16 accumulators per lane, every observation re-read from LDS per unit) cost ~3 us per workgroup.
    {
      const int upw = (((g.K + 3) / 4) + 15) / 16 * 16;  // units per wave, multiple of 16
      for (int u0 = wid * upw; u0 < (wid + 1) * upw && u0 < g.K; u0 += 16) {
        const int unit = u0 + r;
        const int uc = unit < g.K ? unit : g.K - 1;
        f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < g.S; q0 += 4) {
          const int q = q0 + kq;
          const float av = q < g.S ? xs[r * g.S + q] : 0.f;                      // A[row r][k = q]
          const float bv = (q < g.S && unit < g.K) ? g.W1[(size_t)uc * g.S + q] : 0.f;  // B[k = q][unit]
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c1, 0, 0, 0);
        }
        const float bb = g.b1[uc];
        if (unit < g.K) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {  // C: col = lane & 15 -> unit, row = kq * 4 + i -> tile row
            const float v = c1[i] + bb;
            h1s[(kq * 4 + i) * ldh + unit] = v > 0.f ? v : 0.f;
          }
        }
      }
    }
    __syncthreads();
  }

  for (int k0 = kbeg; k0 < kend; k0 += 16 * U) {
    // ---- issue every load of the batch first
    if (!(A_MODE == 2 && k0 == kbeg)) load_b(k0);
    load_a(k0);
    // ---- then the MFMAs
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][tm][j], b[u][tn][j], acc[tm][tn], 0, 0, 0);
          if (ROWSUM) rs[tm] += a[u][tm][j];
        }
      }
    }
  }
  // ---- in-workgroup split-K combine (fixed order)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) s_acc[wid][tm * TN + tn][lane][i] = acc[tm][tn][i];
    if (ROWSUM) s_rs[wid][tm][lane] = rs[tm];
  }
  __syncthreads();
  if (wid != 0) return;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (ROWSUM && tn_blk == 0) {
      float t = ((s_rs[0][tm][lane] + s_rs[1][tm][lane]) + s_rs[2][tm][lane]) + s_rs[3][tm][lane];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (kq == 0 && m_ok[tm]) {
        const int m = m0 + 16 * tm + r;
        if (EPI == EPI_ROWPTR) *g.rowsum_ptr[m] = t;
        else g.rowsum[m] = t;
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int q = tm * TN + tn;
      const int n = n0 + 16 * tn + r;
      // C/D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg
      float hv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mm = m0 + 16 * tm + kq * 4 + i;
        float v = ((s_acc[0][q][lane][i] + s_acc[1][q][lane][i]) + s_acc[2][q][lane][i]) + s_acc[3][q][lane][i];
        if (EPI == EPI_BIAS_RELU || EPI == EPI_HEADPART) {
          v += g.aux[nc[tn]];
          v = v > 0.f ? v : 0.f;
        } else if (EPI == EPI_MASK) {
          const int mmc = mm < g.M ? mm : g.M - 1;
          v = g.aux[(size_t)mmc * g.ldaux + nc[tn]] > 0.f ? v : 0.f;  // relu'(h) of the forward activation
        }
        hv[i] = (n_ok[tn] && mm < g.M) ? v : 0.f;
        if (!n_ok[tn] || mm >= g.M) continue;
        if (EPI == EPI_ROWPTR) g.rowptr[mm][n] = v;
        else if (EPI != EPI_HEADPART || g.C) g.C[(size_t)mm * g.ldc + n] = v;
      }
      if (EPI == EPI_HEADPART) {
        // partial head outputs of this 16-column tile: reduce over the 16 lanes that share kq
        for (int o = 0; o < g.n_out; ++o) {
          const float w = n_ok[tn] ? g.wh[o][n] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float p = hv[i] * w;
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            p += __shfl_xor(p, 4, 64);
            p += __shfl_xor(p, 8, 64);
            const int mm = m0 + 16 * tm + kq * 4 + i;
            if (r == 0 && mm < g.M) {
              if (tn_blk * TN + tn == 0) p += *g.hbias[o];
              g.part[((size_t)(tn_blk * TN + tn) * g.part_rows + mm) * 8 + o] = p;
            }
          }
        }
      }
    }
  }
  if (EPI == EPI_HEADPART && g.tile_flag) {
    // Acting hand-off to the HOST: the partial head outputs went to device-mapped pinned memory;
    // make them visible system-wide, then publish this tile's sequence word.  The host polls the
    // words and finishes the heads (sum over tiles, softmax, sampling): no second launch, no
    // cross-workgroup synchronisation on the device.
    // (one system-scope release: its s_waitcnt vmcnt(0) is wave-wide, so it also covers the partial
    // stores issued by lanes 16/32/48 of this wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_store(g.tile_flag + blockIdx.x, g.flag_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int A_MODE, bool B_KCONT, int EPI, bool ROWSUM, int TM, int TN>
static int launch_gemm(const char* name, const GemmArgs& g, hipStream_t st) {
  const int tiles = ((g.M + 16 * TM - 1) / (16 * TM)) * ((g.N + 16 * TN - 1) / (16 * TN));
  const int kper = ((g.K + 63) / 64) * 16;  // per-wave K range
  const size_t lds = A_MODE == 2 ? sizeof(float) * (16 * (size_t)g.S + 16 * (size_t)(g.K + 4)) : 0;
  constexpr int UMAX = (TM * TN >= 4) ? 4 : 8;  // keep the operand registers of a batch <= 64 + 64
  if (kper > 64 && UMAX == 8) {
    JH_LAUNCH_NAMED(name, (jh_gemm16_kernel<A_MODE, B_KCONT, EPI, ROWSUM, UMAX, TM, TN>), dim3(tiles), dim3(256), lds, st, g);
  } else if (kper > 32) {
    JH_LAUNCH_NAMED(name, (jh_gemm16_kernel<A_MODE, B_KCONT, EPI, ROWSUM, 4, TM, TN>), dim3(tiles), dim3(256), lds, st, g);
  } else {
    JH_LAUNCH_NAMED(name, (jh_gemm16_kernel<A_MODE, B_KCONT, EPI, ROWSUM, 2, TM, TN>), dim3(tiles), dim3(256), lds, st, g);
  }
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// ============================================================================ heads (N = A+1 or 2A+1, tiny)
struct HeadPtrs {
  const float* w[3];  // weight rows [n_i][H]
  const float* b[3];
  float* out[3];      // [B][n_i]
  const float* g[3];  // upstream grads (backward)
  int n[3];
  int groups;
};

// Up to 8 head outputs of one launch as FLAT arrays with static indices (round 5): built on the host.  The grouped form (HeadPtrs walked
// by a run-time loop inside the kernel) put the pointers into a dynamically indexed array, and hipcc fetched weight row after weight row
// behind its own s_waitcnt vmcnt(0): 14 serial L2 round trips per row at H = 512 (11.3 us per launch at B = 2048).
struct HeadFlat {
  const float* w[8];   // weight row of output o ([H]); outputs >= n: a valid row (fetched, ignored)
  const float* b[8];   // its bias
  float* out[8];       // element of row 0 of its output tensor
  const float* g[8];   // backward: element of row 0 of its upstream gradient
  int ld[8];           // row stride of that tensor (A, A, 1)
  int n;
};
// forward: one wave per row; lanes split H.  All (<= 8) outputs' lane-partial dot products first, then their shuffle reductions
// INTERLEAVED (the same tree per output as one reduction after the other: identical bits).  Nets with more than 8 head outputs
// (config.ppo.mujoco on HalfCheetah / Walker / Ant: 2 A + 1 = 13 / 13 / 17) take one launch per 8 outputs.
__global__ void __launch_bounds__(256) jh_mlp_heads_fwd_kernel(int B, int H, const float* __restrict__ h2, HeadFlat hf) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* hr = h2 + (size_t)b * H;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bias[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) bias[o] = *hf.b[o];  // with the weight rows, not one by one between the stores
  for (int k = lane * 4; k < H; k += 256) {
    const float4 hv = *reinterpret_cast<const float4*>(hr + k);
    float4 wv[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) wv[o] = *reinterpret_cast<const float4*>(hf.w[o] + k);  // all eight in flight (rows >= n: row 0 again)
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      acc[o] = fmaf(hv.x, wv[o].x, acc[o]);
      acc[o] = fmaf(hv.y, wv[o].y, acc[o]);
      acc[o] = fmaf(hv.z, wv[o].z, acc[o]);
      acc[o] = fmaf(hv.w, wv[o].w, acc[o]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] += __shfl_xor(acc[o], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < hf.n) hf.out[o][(size_t)b * hf.ld[o]] = acc[o] + bias[o];
  }
}

// backward: dh2[b][k] = relu'(h2[b][k]) * sum_o g[b][o] * Wh[o][k]; the first gld / 4 threads of each row also pack the head gradients
// into g_all[b][gld] (the A operand of the head-weight-gradient GEMM).  Four consecutive hidden units per thread (H % 4 == 0); the sum over
// the outputs runs in output order per element (an fmaf chain, carried through `acc_io` from one launch of 8 outputs to the next).
// first: this launch holds outputs o0 .. o0 + n - 1 and is the first (the chain starts at 0); last: it applies relu' and stores dh2
// (earlier launches store the raw partial chain).  Round 5: flat descriptors with static indices -- the grouped form fetched
// (gradient, weight row) pair after pair behind s_waitcnt vmcnt(0) (7 serial round trips per thread at Hopper's 7 outputs).
__global__ void __launch_bounds__(256) jh_mlp_heads_bwd_dh_kernel(int B, int H, const float* __restrict__ h2,
                                                                  float* __restrict__ dh2, float* __restrict__ g_all,
                                                                  HeadFlat hf, int gld, int o0, int first, int last,
                                                                  const float* __restrict__ dv2, const float* __restrict__ mix, int ov, PpoFinish fin) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int h4 = H >> 2;
  if (i4 >= (int64_t)B * h4) return;
  const int b = (int)(i4 / h4), k = 4 * (int)(i4 - (int64_t)b * h4);
  float gv[8];
  float4 w[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) {  // all fetches in flight (outputs >= n: output 0 again, ignored)
    gv[o] = hf.g[o][(size_t)b * hf.ld[o]];
    w[o] = *reinterpret_cast<const float4*>(hf.w[o] + k);
  }
  if (dv2) {  // the one-launch loss (jh_ppo_onepass_kernel) left both critic branches' value gradients: slot `ov` is the value head
    float w1, w2;
    const float g2 = dv2[b];
    if (fin.partial) {  // ... and its workgroups' partials (nb <= 64): every wave reduces them in the two-pass order (the same bits everywhere), workgroup 0 writes the statistics
      float t[6];
      ppo_reduce_partials_wave(fin.partial, fin.nb, t);
      ppo_finish_stats(t[0], t[1], t[2], t[3], t[4], t[5], fin.B, fin.ent_count, fin.vf, fin.ent, w1, w2, (blockIdx.x == 0 && threadIdx.x == 0) ? fin.stats : nullptr);
    } else {
      w1 = mix[0]; w2 = mix[1];
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o == ov) gv[o] = w1 * gv[o] + w2 * g2;  // jh_ppo_critic_select_kernel's expression
  }
  float4 acc = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(dh2 + (size_t)b * H + k);
  const float4 hv = *reinterpret_cast<const float4*>(h2 + (size_t)b * H + k);
  float mine[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (o < hf.n) {
      acc.x = fmaf(gv[o], w[o].x, acc.x); acc.y = fmaf(gv[o], w[o].y, acc.y); acc.z = fmaf(gv[o], w[o].z, acc.z); acc.w = fmaf(gv[o], w[o].w, acc.w);
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[j] = (o0 + o == k + j) ? gv[o] : mine[j];
    }
  }
  if (last) acc = make_float4(hv.x > 0.f ? acc.x : 0.f, hv.y > 0.f ? acc.y : 0.f, hv.z > 0.f ? acc.z : 0.f, hv.w > 0.f ? acc.w : 0.f);
  *reinterpret_cast<float4*>(dh2 + (size_t)b * H + k) = acc;
  // this launch's outputs that fall into the thread's four packed columns (the columns beyond the head outputs stay 0: the first launch
  // writes all of the row's gld columns, later launches only the ones they own)
  if (k < gld) {
    float* dst = g_all + (size_t)b * gld + k;
    if (first) {
      *reinterpret_cast<float4*>(dst) = make_float4(mine[0], mine[1], mine[2], mine[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j >= o0 && k + j < o0 + hf.n) dst[j] = mine[j];
    }
  }
  // a row narrower than the packed gradient (hidden 16 / 32 under Ant's 20 or Humanoid's 36 columns: ADVICE r5): the row's last thread also
  // packs the columns [H, gld) -- nobody's k reaches them, and they stayed at their memset 0 = silently zero head gradients
  if (H < gld && k == H - 4) {
    for (int kk = H; kk < gld; kk += 4) {
      float* dst = g_all + (size_t)b * gld + kk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) v = (o < hf.n && o0 + o == kk + j) ? gv[o] : v;
        if (first || (kk + j >= o0 && kk + j < o0 + hf.n)) dst[j] = v;
      }
    }
  }
}

// ============================================================================ clip_grad_norm_ + Adam
// hyper (device): [0] lr [1] beta1 [2] beta2 [3] eps [4] step [5] 1-beta1^t [6] sqrt(1-beta2^t)
__global__ void __launch_bounds__(256) jh_gradnorm_kernel(int64_t n, const float* __restrict__ g,
                                                          float* __restrict__ partial, float* __restrict__ hyper) {
  __shared__ float s_red[16];
  float acc = 0.f;
  // eight grid-stride elements per round, all fetched before the first fmaf (round 5: one fetch per s_waitcnt vmcnt(0) before --
  // five serial round trips for Hopper's 272 391 parameters); the order of the chain is unchanged
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 8 * stride) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * stride;
      v[u] = g[i < n ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * stride < n) acc = fmaf(v[u], v[u], acc);
  }
  acc = jh_block_reduce(acc, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) jh_adam_advance(hyper);  // nobody reads hyper in this kernel
}

// FUSED (the four-launch minibatch update, jh_ppo_mb.hip): there is no norm kernel in front.  `partial` holds the sums
// of squares of the gradient tiles the backward's workgroups wrote (dW2 | db2 | head weights and biases), and the
// (dW1 | db1) head of the bucket still is `tiles_m` per-row-tile slabs in `part`: EVERY workgroup sums all slabs (in
// row-tile order: the same bits everywhere, so all workgroups derive the same clip coefficient) for the norm, and keeps
// the elements it is about to update.  164 KB of L2 reads per workgroup for CartPole against a 7 us launch.
template <bool FUSED>
__global__ void __launch_bounds__(256) jh_adam_kernel(int64_t n, float* __restrict__ p, float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ partial, int n_partial,
                                                      const float* __restrict__ hyper, float max_norm,
                                                      float* __restrict__ norm_out, const float* __restrict__ part,
                                                      int tiles_m, int64_t n_head) {
  __shared__ float s_red[16];
  // round 5: the first pass's parameter / moment / gradient rows are fetched BEFORE the norm prologue (they do not depend on it):
  // the prologue's partial sums, slab sums and block reduction used to stand in front of these fetches as one more dependent
  // round trip per launch (tools/isa_chain.py).  The grid (kNormBlocks x 256 threads x 4 floats) covers all but a few rows in the
  // first pass; later passes fetch in the loop as before.  Same arithmetic, same bits.
  const int64_t n4 = n >> 2;  // the buckets are 16-byte aligned: 16-byte accesses + <= 3 trailing elements
  float4 *p4 = reinterpret_cast<float4*>(p), *g4 = reinterpret_cast<float4*>(g), *m4 = reinterpret_cast<float4*>(m), *v4 = reinterpret_cast<float4*>(v);
  const int64_t i_first = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i_fc = i_first < n4 ? i_first : 0;
  const float4 pp0 = p4[i_fc], mm0 = m4[i_fc], vv0 = v4[i_fc], gg0 = g4[i_fc];
  float acc = 0.f;
  bool partials_added = false;
  // the partial sums of squares, called once BEHIND the slab fetches below: issued in front of them, the first use of a partial is a
  // wait for everything older in the queue, i.e. a round trip before the 48 slab fetches even start.  Two per thread together
  // (a plain loop fetched, waited and added once per pass); same order of additions
  auto add_partials = [&]() {
    const int ia = threadIdx.x, ib = threadIdx.x + 256;
    const float pa = partial[ia < n_partial ? ia : 0], pb = partial[ib < n_partial ? ib : 0];
    acc += ia < n_partial ? pa : 0.f;
    acc += ib < n_partial ? pb : 0.f;
    for (int i = threadIdx.x + 512; i < n_partial; i += 256) acc += partial[i];
    partials_added = true;
  };
  float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t h4 = n_head >> 2;
  if (FUSED) {
    // tiles_m <= 16 (host).  ALL slab loads of three elements are issued before the first add (48 x 16 bytes in flight
    // per thread; one wave per SIMD, so the registers are there): in batches of 8 this prologue was six dependent L2
    // round trips = 6 us, as much as the norm launch it replaces.
    const float4* part4 = reinterpret_cast<const float4*>(part);
    constexpr int JU = 3, TU = 16;
    for (int64_t j0 = 0; j0 * 256 < h4; j0 += JU) {
      float4 w[JU][TU];
#pragma unroll
      for (int ju = 0; ju < JU; ++ju) {
        const int64_t i = threadIdx.x + 256 * (j0 + ju);
        const int64_t ic = i < h4 ? i : h4 - 1;
#pragma unroll
        for (int tu = 0; tu < TU; ++tu) w[ju][tu] = part4[(size_t)(tu < tiles_m ? tu : tiles_m - 1) * h4 + ic];
      }
      if (!partials_added) add_partials();
#pragma unroll
      for (int ju = 0; ju < JU; ++ju) {
        const int64_t i = threadIdx.x + 256 * (j0 + ju);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tu = 0; tu < TU; ++tu)
          if (tu < tiles_m) { q.x += w[ju][tu].x; q.y += w[ju][tu].y; q.z += w[ju][tu].z; q.w += w[ju][tu].w; }  // row-tile order
        if (i < h4) {
          acc = fmaf(q.x, q.x, acc); acc = fmaf(q.y, q.y, acc); acc = fmaf(q.z, q.z, acc); acc = fmaf(q.w, q.w, acc);
          if (j0 + ju == blockIdx.x) mine = q;  // element blockIdx.x * 256 + threadIdx.x of the update loop below
        }
      }
    }
  }
  if (!partials_added) add_partials();
  const float total = sqrtf(jh_block_reduce(acc, s_red, JhAdd(), 0.f));
  const float bc1 = hyper[5], bc2s = hyper[6];
  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total + 1e-6), clamped to 1
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
  const float lr = hyper[JH_HY_LR], b2 = hyper[JH_HY_B2], eps = hyper[JH_HY_EPS];
  const float omb1 = hyper[JH_HY_OMB1], omb2 = hyper[JH_HY_OMB2];  // (float)(1.0 - beta): torch derives them in double
  const float step_size = lr / bc1;
  auto upd = [&](float& pi, float& gi_, float& mi_, float& vi_) {
    const float gi = gi_ * coef;
    gi_ = gi;                                   // clip is in place, like the reference
    const float mi = mi_ + omb1 * (gi - mi_);  // exp_avg.lerp_(grad, 1-beta1)
    const float vi = vi_ * b2 + omb2 * gi * gi;
    mi_ = mi;
    vi_ = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    pi = pi - step_size * (mi / denom);
  };
  for (int64_t i = i_first; i < n4; i += (int64_t)gridDim.x * 256) {
    const bool first = i == i_first;
    float4 pp = first ? pp0 : p4[i], mm = first ? mm0 : m4[i], vv = first ? vv0 : v4[i];
    float4 gg = (FUSED && i < h4) ? mine : (first ? gg0 : g4[i]);  // h4 <= gridDim.x * 256 (host): a head element is met in the first pass only
    upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
    p4[i] = pp; g4[i] = gg; m4[i] = mm; v4[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - (n4 << 2))) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
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
  // <= 8 head outputs: every path (the 4-launch minibatch update, the persistent acting kernel).  9 .. 40 (continuous A <= 19: Humanoid's
  // 17; discrete A <= 39): the separate forward / backward calls and the tiled engine -- wider action spaces than Hopper's run, slower
  JH_ARG((continuous ? 2 * A + 1 : A + 1) <= kMaxHeadOutputs);
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
  n->n_out = continuous ? 2 * A + 1 : A + 1;
  n->gld = n->n_out <= 8 ? 8 : (n->n_out + 3) / 4 * 4;
  JH_HIP(hipMalloc((void**)&n->g_all, sizeof(float) * (size_t)n->gld * (size_t)max_rows));
  n->max_act_rows = max_rows < 1024 ? max_rows : 1024;
  {
    const size_t tiles = (size_t)((n->max_act_rows + 15) / 16) * (size_t)(H / 16);
    JH_HIP(hipHostMalloc((void**)&n->obs_pin_h, sizeof(float) * (size_t)n->max_act_rows * (size_t)S, hipHostMallocMapped));
    JH_HIP(hipHostGetDevicePointer((void**)&n->obs_pin_d, n->obs_pin_h, 0));
    JH_HIP(hipHostMalloc((void**)&n->part_pin_h, sizeof(float) * 8 * (size_t)n->max_act_rows * (size_t)(H / 16), hipHostMallocMapped));
    JH_HIP(hipHostGetDevicePointer((void**)&n->part_pin_d, n->part_pin_h, 0));
    JH_HIP(hipHostMalloc((void**)&n->flag_pin_h, sizeof(unsigned) * tiles, hipHostMallocMapped));
    JH_HIP(hipHostGetDevicePointer((void**)&n->flag_pin_d, n->flag_pin_h, 0));
    memset(n->flag_pin_h, 0, sizeof(unsigned) * tiles);
    n->act_seed = seed;
    if (n->n_out > 8) {
      JH_HIP(hipHostMalloc((void**)&n->act_out_h, sizeof(float) * (size_t)n->max_act_rows * (size_t)n->n_out, hipHostMallocMapped));
      JH_HIP(hipHostGetDevicePointer((void**)&n->act_out_d, n->act_out_h, 0));
    }
  }
  n->tg_ws_floats = (size_t)4 << 20;
  n->tg_cnt_slots = 4096;
  JH_HIP(hipMalloc((void**)&n->xg, sizeof(float) * (size_t)max_rows * (size_t)S));
  JH_HIP(hipMalloc((void**)&n->tg_ws, sizeof(float) * n->tg_ws_floats));
  JH_HIP(hipMalloc((void**)&n->tg_cnt, sizeof(unsigned) * (size_t)n->tg_cnt_slots * kTgemmCntStride));
  JH_HIP(hipMemset(n->tg_cnt, 0, sizeof(unsigned) * (size_t)n->tg_cnt_slots * kTgemmCntStride));
  {
    const size_t part_bytes = sizeof(float) * 8 * (size_t)max_rows * (size_t)(H / 16);
    JH_HIP(hipMalloc((void**)&n->fwd_part, part_bytes));
    JH_HIP(hipMemset(n->fwd_part, 0, part_bytes));  // head slots >= n_out are never written: they must read as 0
    JH_HIP(hipMemset(n->g_all, 0, sizeof(float) * (size_t)n->gld * (size_t)max_rows));
    JH_HIP(hipMalloc((void**)&n->dv2, sizeof(float) * ((size_t)(max_rows < 1024 ? max_rows : 1024) + 8)));
    n->stats_tmp = n->dv2 + (max_rows < 1024 ? max_rows : 1024);
    const size_t slabs = (size_t)(((max_rows < 1024 ? max_rows : 1024) + 15) / 16);
    JH_HIP(hipMalloc((void**)&n->part_w1, sizeof(float) * slabs * ((size_t)H * S + H + 8 * (size_t)H)));
    JH_HIP(hipMalloc((void**)&n->ssq_part, sizeof(float) * ((size_t)(H / 32) * (H / 32) + H / 32 + 1)));
  }
  {  // jh_pponet_ppo_update_rows: raw heads + their gradients as separate arrays ([max_rows][A] x 2 x 2, [max_rows] x 3), the loss's partials, {w1, w2}, a ticket
    const size_t rows = (size_t)max_rows, per = 4 * (size_t)A + 3;
    n->upd_floats = rows * per + 8 * ((rows + 255) / 256) + 16;
    JH_HIP(hipMalloc((void**)&n->upd_ws, sizeof(float) * n->upd_floats));
    JH_HIP(hipMemset(n->upd_ws, 0, sizeof(float) * n->upd_floats));
  }
  JH_HIP(hipMalloc((void**)&n->norm_partial, sizeof(float) * kNormSlots));
  JH_HIP(hipMalloc((void**)&n->hyper, sizeof(float) * JH_HY_FLOATS));
  float hy[JH_HY_FLOATS];
  jh_hyper_fill(hy, 1e-3, 0.9, 0.999, 1e-8, 0.0);
  JH_HIP(hipMemcpy(n->hyper, hy, sizeof(hy), hipMemcpyHostToDevice));
  *out = n;
  return JH_OK;
}

JH_EXPORT void jh_pponet_destroy(jh_pponet* n) {
  if (!n) return;
  (void)hipSetDevice(n->ctx->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(n->h1); (void)hipFree(n->h2); (void)hipFree(n->dh1); (void)hipFree(n->dh2);
  (void)hipFree(n->g_all);
  (void)hipHostFree(n->obs_pin_h); (void)hipHostFree(n->part_pin_h); (void)hipHostFree(n->flag_pin_h);
  if (n->act_out_h) (void)hipHostFree(n->act_out_h);
  (void)hipFree(n->norm_partial); (void)hipFree(n->hyper); (void)hipFree(n->dv2);
  (void)hipFree(n->fwd_part); (void)hipFree(n->part_w1); (void)hipFree(n->ssq_part);
  (void)hipFree(n->tg_ws); (void)hipFree(n->tg_cnt); (void)hipFree(n->xg); (void)hipFree(n->upd_ws);
  delete n;
}

// Device address of the optimizer's hyper block {lr, beta1, beta2, eps, step, ...} (8 floats): lets a caller deliver the next learning
// rate with a copy it already makes (jh_collector_set_ride_along) instead of jh_pponet_set_lr's own H2D.
JH_EXPORT void* jh_pponet_hyper_ptr(jh_pponet* n) { return n ? (void*)n->hyper : nullptr; }

// The host-side sampling stream of jh_pponet_act_* / the collectors (counter-based: action of env row w at acting step c =
// f(seed, c, w)): read or (set != 0) restore {seed, counter} -- part of a complete checkpoint.
JH_EXPORT int jh_pponet_act_rng(jh_pponet* n, uint64_t* seed, uint64_t* counter, int32_t set) {
  JH_ARG(n && seed && counter);
  if (set) { n->act_seed = *seed; n->act_ctr = *counter; }
  else { *seed = n->act_seed; *counter = n->act_ctr; }
  return JH_OK;
}

JH_EXPORT int jh_pponet_set_hyper(jh_pponet* n, double lr, double beta1, double beta2, double eps, double step,
                                  jh_stream stream) {
  JH_ARG(n != nullptr);
  // staged through a pinned slab so the copy is a true async H2D that a later graph launch sees.  step < 0: the step counter (and
  // the bias corrections derived from it) stay as they are -- two copies around them
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(n->ctx, 64, &slab);
  if (rc) return rc;
  float* h = (float*)slab->host;
  jh_hyper_fill(h, lr, beta1, beta2, eps, step < 0 ? 0.0 : step);
  if (step >= 0) {
    JH_HIP(hipMemcpyAsync(n->hyper, h, sizeof(float) * JH_HY_FLOATS, hipMemcpyHostToDevice, jh_s(stream)));
  } else {
    JH_HIP(hipMemcpyAsync(n->hyper, h, sizeof(float) * 4, hipMemcpyHostToDevice, jh_s(stream)));
    JH_HIP(hipMemcpyAsync(n->hyper + JH_HY_OMB1, h + JH_HY_OMB1, sizeof(float) * (JH_HY_FLOATS - JH_HY_OMB1), hipMemcpyHostToDevice, jh_s(stream)));
  }
  return jh_ctx_slab_release(n->ctx, slab, jh_s(stream));
}

JH_EXPORT int jh_pponet_set_lr(jh_pponet* n, double lr, jh_stream stream) {
  JH_ARG(n != nullptr);
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(n->ctx, 64, &slab);
  if (rc) return rc;
  *(float*)slab->host = (float)lr;
  JH_HIP(hipMemcpyAsync(n->hyper, slab->host, sizeof(float), hipMemcpyHostToDevice, jh_s(stream)));
  return jh_ctx_slab_release(n->ctx, slab, jh_s(stream));
}

static HeadPtrs head_ptrs(jh_pponet* n, float* out0, float* out1, float* outv, const float* g0, const float* g1,
                          const float* gv) {
  HeadPtrs hp{};
  int g = 0;
  hp.w[g] = n->params + n->o_wh0; hp.b[g] = n->params + n->o_bh0; hp.out[g] = out0; hp.g[g] = g0; hp.n[g] = n->A; ++g;
  if (n->cont) {
    hp.w[g] = n->params + n->o_wh1; hp.b[g] = n->params + n->o_bh1; hp.out[g] = out1; hp.g[g] = g1; hp.n[g] = n->A; ++g;
  }
  hp.w[g] = n->params + n->o_wv; hp.b[g] = n->params + n->o_bv; hp.out[g] = outv; hp.g[g] = gv; hp.n[g] = 1; ++g;
  hp.groups = g;
  return hp;
}

// Outputs o0 .. o0 + 7 of the net's heads (flat output order: head0[A], head1[A] (continuous), value) as the kernels' HeadFlat.
static HeadFlat head_flat(jh_pponet* n, const HeadPtrs& hp, int o0) {
  HeadFlat hf{};
  int o_flat = 0, k = 0;
  for (int g = 0; g < hp.groups; ++g)
    for (int o = 0; o < hp.n[g]; ++o, ++o_flat) {
      if (o_flat < o0 || k >= 8) continue;
      hf.w[k] = hp.w[g] + (size_t)o * n->H;
      hf.b[k] = hp.b[g] + o;
      hf.out[k] = hp.out[g] ? hp.out[g] + o : nullptr;
      hf.g[k] = hp.g[g] ? hp.g[g] + o : nullptr;
      hf.ld[k] = hp.n[g];
      ++k;
    }
  hf.n = k;
  for (; k < 8; ++k) {  // unused slots: valid addresses (fetched unconditionally, ignored)
    hf.w[k] = hf.w[0]; hf.b[k] = hf.b[0]; hf.out[k] = hf.out[0]; hf.g[k] = hf.g[0]; hf.ld[k] = hf.ld[0];
  }
  return hf;
}

// Flat list of head outputs: weight row / weight-grad row / bias / bias-grad of output o.
static int head_rows(jh_pponet* n, const float** w, float** dw, const float** b, float** db) {
  int o = 0;
  for (int a = 0; a < n->A; ++a, ++o) {
    w[o] = n->params + n->o_wh0 + (int64_t)a * n->H; dw[o] = n->grads + n->o_wh0 + (int64_t)a * n->H;
    b[o] = n->params + n->o_bh0 + a; db[o] = n->grads + n->o_bh0 + a;
  }
  if (n->cont)
    for (int a = 0; a < n->A; ++a, ++o) {
      w[o] = n->params + n->o_wh1 + (int64_t)a * n->H; dw[o] = n->grads + n->o_wh1 + (int64_t)a * n->H;
      b[o] = n->params + n->o_bh1 + a; db[o] = n->grads + n->o_bh1 + a;
    }
  w[o] = n->params + n->o_wv; dw[o] = n->grads + n->o_wv; b[o] = n->params + n->o_bv; db[o] = n->grads + n->o_bv;
  return o + 1;
}

// Minibatches of >= kTiledRows rows (config.ppo.mujoco: 2048) run on the LDS-tiled engine (operand reuse across
// a 64 x 64 tile); smaller ones on the latency-oriented kernels of jh_ppo_mb.hip / jh_gemm16.
constexpr int kTiledRows = 1024;
static inline bool pponet_use_tiled(int B) { return B >= kTiledRows; }

static PmbHeads pmb_heads(jh_pponet* n) {
  PmbHeads hd{};
  hd.n_out = head_rows(n, hd.w, hd.dw, hd.b, hd.db);
  return hd;
}

__global__ void __launch_bounds__(256) jh_rowgather_f32_kernel(int B, int S, const float* __restrict__ x, const int64_t* __restrict__ idx, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * S) return;
  const int b = (int)(i / S), s = (int)(i - (int64_t)b * S);
  out[i] = x[(idx ? idx[b] : (int64_t)b) * S + s];
}

// dW1[h][s] = sum_b dh1[b][h] x[idx[b]][s], db1[h] = sum_b dh1[b][h] for minibatches on the tiled path (B >= 1024): K = B is long, but
// N = S (11 for Hopper) is a sliver -- on the 32 x 32 MFMA tile engine that was 0.2-0.6 % of the fp32 matrix peak (25 us at B = 2048) plus
// a row-gather launch.  It is a column reduction of dh1 weighted by S broadcast values per row: HBM-bound on reading dh1 once.
// The head weight gradients dWh[o][h] = sum_b g_all[b][o] h2[b][h] are the same shape (a column reduction of h2 weighted by the <= 8
// packed head gradients of the row; on the tile engine: an unaligned [B][8] operand on the element-wise fetch path, 19 us at B = 2048)
// and ride in the same two launches when h2 is given.
//   pass 1  grid (H / 64, slabs): 256 threads = 64 h x 4 row lanes; a slab's observation rows (gathered through idx) and head-gradient
//           rows sit in LDS; every thread walks its rows with one coalesced dh1 (+ h2) load + S (+ 8) FMAs, the 4 row lanes combine
//           through LDS -> partial[z][h][S + 1 (+ 8)]
//   pass 2  out[h][s] = sum_z partial[z][h][s] in slab order (deterministic); head bias gradients = column sums of g_all (last workgroup)
// Round 6: the launch also carries the sum of squares of the part of the gradient bucket that is ALREADY complete when it starts (dW2 | db2 [| head weights]
// from the grouped GEMM before it) as kNormFoldBlocks more workgroups (blockIdx.y >= y0): the global norm of clip_grad_norm_ then needs no launch of its
// own -- the rest of the bucket's squares come out of the combine kernel below, which forms those elements anyway.  Same loop as jh_gradnorm_kernel.
constexpr int kNormFoldY = 8;  // x gridDim.x (= H / 64) workgroups
struct NormJob {
  const float* g;   // null: no norm job in this launch
  int64_t n;
  float* partial;   // [kNormFoldY * gridDim.x]
  int y0;
};
template <int SP>
__global__ void __launch_bounds__(256) jh_ppo_dw1_partial_kernel(int B, int H, int S, int rows_per, const float* __restrict__ dh1, const float* __restrict__ x,
                                                                 const int64_t* __restrict__ idx, float* __restrict__ partial, const float* __restrict__ h2,
                                                                 const float* __restrict__ g8, NormJob nj) {
  extern __shared__ float s_dyn[];
  if (nj.g && (int)blockIdx.y >= nj.y0) {
    const int id = ((int)blockIdx.y - nj.y0) * (int)gridDim.x + (int)blockIdx.x, nblk = kNormFoldY * (int)gridDim.x;
    float acc = 0.f;
    const int64_t stride = (int64_t)nblk * 256;
    for (int64_t i0 = (int64_t)id * 256 + threadIdx.x; i0 < nj.n; i0 += 8 * stride) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = i0 + u * stride;
        v[u] = nj.g[i < nj.n ? i : i0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * stride < nj.n) acc = fmaf(v[u], v[u], acc);
    }
    acc = jh_block_reduce(acc, s_dyn, JhAdd(), 0.f);
    if (threadIdx.x == 0) nj.partial[id] = acc;  // (idempotent: the profiler may repeat this launch; the step counter advances in the combine kernel)
    return;
  }
  const bool heads = h2 != nullptr;
  constexpr int NA = SP + 1 + 8;                               // accumulators per thread: S taps | row sum | 8 head columns
  float* s_x = s_dyn;                                          // [rows_per][SP]
  float* s_g = s_dyn + (size_t)rows_per * SP;                  // [rows_per][8]
  float* s_acc = s_g + (size_t)rows_per * 8;                   // [4][64][NA]
  const int hl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int h = blockIdx.x * 64 + hl;
  const int z = blockIdx.y, b0 = z * rows_per;
  int nb = B - b0;
  if (nb > rows_per) nb = rows_per;
  // Fetch order (round 5; the launch was a chain of ~12 dependent round trips: three staging passes of idx -> x, two of g8, then one
  // per round of four rows): this thread's first SIXTEEN rows of dh1 / h2 (two rounds of eight: every row of a 64-row slab) are
  // requested first -- they depend on nothing --, then the slab's row indices, its observation rows and its head-gradient rows, each as one
  // batch.  Addresses past the end are clamped and the values dropped (a conditional load is sunk behind its use = serial again).
  constexpr int RU = 8;
  const int hc = h < H ? h : H - 1;
  const float* h2c = heads ? h2 : dh1;
  auto fetch = [&](int r0, float (&g)[RU], float (&a2)[RU]) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = r0 + 4 * u;
      const int rc = r < nb ? r : nb - 1;
      g[u] = dh1[(size_t)(b0 + rc) * H + hc];
      a2[u] = h2c[(size_t)(b0 + rc) * H + hc];
    }
  };
  float gA[RU], aA[RU], gB[RU], aB[RU];
  fetch(rl, gA, aA);
  fetch(rl + 4 * RU, gB, aB);
  float* dummy = s_acc + threadIdx.x;  // not in use before the barrier
  const float4* g4 = (const float4*)((heads ? g8 : dh1) + (size_t)b0 * 8);  // [nb][8] contiguous, 32-byte rows
  const int ng = heads ? nb * 2 : 0;
  const float4 g_first = g4[(int)threadIdx.x < ng ? threadIdx.x : 0];
  __builtin_amdgcn_sched_barrier(0);
  {
    constexpr int SU = 4;
    const int nx = nb * SP;
    for (int i0 = threadIdx.x; i0 < nx; i0 += 256 * SU) {
      int64_t row[SU];
      int qq[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + 256 * u;
        const int ic = i < nx ? i : nx - 1;
        const int r = ic / SP;
        qq[u] = ic - r * SP;
        row[u] = idx ? idx[b0 + r] : (int64_t)(b0 + r);
      }
      float v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = x[row[u] * S + (qq[u] < S ? qq[u] : S - 1)];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + 256 * u;
        *(i < nx ? s_x + i : dummy) = qq[u] < S ? v[u] : 0.f;
      }
    }
    *((int)threadIdx.x < ng ? (float4*)s_g + threadIdx.x : (float4*)s_acc + threadIdx.x) = g_first;
    for (int i = threadIdx.x + 256; i < ng; i += 256) ((float4*)s_g)[i] = g4[i];
  }
  __syncthreads();
  float acc[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) acc[q] = 0.f;
  // the FMA order per accumulator is this thread's rows in ascending order, as before
  auto consume = [&](int r0, const float (&g)[RU], const float (&a2)[RU]) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = r0 + 4 * u;
      if (r >= nb) break;
      const float* xr = s_x + (size_t)r * SP;
#pragma unroll
      for (int q = 0; q < SP; ++q) acc[q] = fmaf(g[u], xr[q], acc[q]);
      acc[SP] += g[u];
      if (heads) {
        const float* gr = s_g + (size_t)r * 8;
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[SP + 1 + o] = fmaf(a2[u], gr[o], acc[SP + 1 + o]);
      }
    }
  };
  if (h < H) {
    for (int r0 = rl; r0 < nb; r0 += 8 * RU) {
      consume(r0, gA, aA);
      if (r0 + 8 * RU < nb) fetch(r0 + 8 * RU, gA, aA);
      consume(r0 + 4 * RU, gB, aB);
      if (r0 + 12 * RU < nb) fetch(r0 + 12 * RU, gB, aB);
    }
  }
  float* mine = s_acc + ((size_t)rl * 64 + hl) * NA;
#pragma unroll
  for (int q = 0; q < NA; ++q) mine[q] = acc[q];
  __syncthreads();
  if (rl == 0 && h < H) {
    const int PS = S + 1 + (heads ? 8 : 0);
    float* out = partial + ((size_t)z * H + h) * PS;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      if (q < S || q == SP || (heads && q > SP)) {
        const float v = ((s_acc[((size_t)0 * 64 + hl) * NA + q] + s_acc[((size_t)1 * 64 + hl) * NA + q]) + s_acc[((size_t)2 * 64 + hl) * NA + q]) +
                        s_acc[((size_t)3 * 64 + hl) * NA + q];
        out[q < S ? q : (q == SP ? S : S + 1 + (q - SP - 1))] = v;
      }
    }
  }
}

struct HeadGradOut {
  float* dw[8];   // row o of the head weight gradients ([H] each), nullptr beyond n_out
  float* db[8];
  const float* g8;  // [B][8] packed head gradients (bias gradients = its column sums)
  int n_out, B;
};
// ssq (optional) [gridDim.x]: the sum of squares of what this workgroup wrote (see NormJob above)
__global__ void __launch_bounds__(256) jh_ppo_dw1_combine_kernel(int H, int S, int slabs, const float* __restrict__ partial, float* __restrict__ dW1,
                                                                 float* __restrict__ db1, HeadGradOut hg, float* __restrict__ ssq, float* __restrict__ hyper_advance) {
  __shared__ float s_ssq[16];
  if (hyper_advance && blockIdx.x == 0 && threadIdx.x == 0) jh_adam_advance(hyper_advance);  // what jh_gradnorm_kernel does when it runs: nobody reads hyper in this launch
  const int PS = S + 1 + (hg.n_out > 0 ? 8 : 0);
  const int n_main = (H * PS + 255) / 256;
  if ((int)blockIdx.x >= n_main) {  // head bias gradients: 8 columns x 32 row lanes, combined in lane order
    __shared__ float s_b[32][8];
    const int o = threadIdx.x & 7, rl = threadIdx.x >> 3;
    float v = 0.f;
    for (int b0 = rl; b0 < hg.B; b0 += 32 * 8) {  // eight fetches in flight (one per wait before: 64 serial round trips at B = 2048 -- this workgroup WAS the launch's 9.6 us); same order of additions
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 32 * u;
        q[u] = hg.g8[(size_t)(b < hg.B ? b : b0) * 8 + o];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b0 + 32 * u < hg.B) v += q[u];
    }
    s_b[rl][o] = v;
    __syncthreads();
    float sq = 0.f;
    if (threadIdx.x < 8 && threadIdx.x < hg.n_out) {
      float t = s_b[0][threadIdx.x];
      for (int k = 1; k < 32; ++k) t += s_b[k][threadIdx.x];
      *hg.db[threadIdx.x] = t;
      sq = t * t;
    }
    if (ssq) {
      sq = jh_block_reduce(sq, s_ssq, JhAdd(), 0.f);
      if (threadIdx.x == 0) ssq[blockIdx.x] = sq;
    }
    return;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  bool wrote = false;
  if (i < H * PS) {
#pragma unroll 8
    for (int z = 0; z < slabs; ++z) v += partial[(size_t)z * H * PS + i];
    const int h = i / PS, q = i - h * PS;
    wrote = true;
    if (q < S) dW1[(size_t)h * S + q] = v;
    else if (q == S) db1[h] = v;
    else if (q - S - 1 < hg.n_out) hg.dw[q - S - 1][h] = v;
    else wrote = false;
  }
  if (ssq) {
    const float sq = jh_block_reduce(wrote ? v * v : 0.f, s_ssq, JhAdd(), 0.f);
    if (threadIdx.x == 0) ssq[blockIdx.x] = sq;
  }
}

// slabs / rows per slab / dynamic LDS bytes of the column reduction for B rows (part_w1 holds 64 slabs of H * (S + 1 + 8) floats)
static size_t pponet_dw1_plan(int B, int S, int* slabs_out, int* rows_per_out) {
  int slabs = (B + 63) / 64;
  if (slabs > 64) slabs = 64;
  const int rows_per = (B + slabs - 1) / slabs;
  slabs = (B + rows_per - 1) / rows_per;
  const int sp = (S + 3) / 4 * 4;
  if (slabs_out) *slabs_out = slabs;
  if (rows_per_out) *rows_per_out = rows_per;
  return sizeof(float) * ((size_t)rows_per * (sp + 8) + 4 * 64 * (size_t)(sp + 1 + 8));
}
// the reduction keeps a slab's observation and head-gradient rows in LDS: beyond ~24 k rows the tile engine takes over again
static bool pponet_dw1_reduce_fits(int B, int S) { return S <= 16 && pponet_dw1_plan(B, S, nullptr, nullptr) <= 60 * 1024; }

// with_norm: the two launches also leave the global norm's sums of squares in n->norm_partial[0 .. n->norm_slots) (NormJob above): the caller goes straight
// to pponet_adam.  Only for a bucket nobody reduces in between (no data-parallel hook).
static int pponet_dw1_reduce(jh_pponet* n, int B, const float* d_x, const int64_t* d_idx, bool heads, hipStream_t st, bool with_norm = false) {
  const int H = n->H, S = n->S;
  int slabs, rows_per;
  const size_t lds = pponet_dw1_plan(B, S, &slabs, &rows_per);
  const dim3 grid((unsigned)((H + 63) / 64), (unsigned)(slabs + (with_norm ? kNormFoldY : 0)));
  NormJob nj{};
  const int n_rest = kNormFoldY * (int)grid.x;
  if (with_norm) {  // what the grouped GEMM wrote: dW2 | db2, and the head weights / biases too when the column reduction does not carry them
    nj.g = n->grads + n->o_w2; nj.n = (heads ? n->o_wh0 : n->n_params) - n->o_w2; nj.partial = n->norm_partial; nj.y0 = slabs;
  }
  const int sp = (S + 3) / 4 * 4;
  if (lds > 60 * 1024) return jh_fail(JH_ERR_ARG, "dW1 reduction: %zu bytes of LDS for %d rows per slab", lds, rows_per);
  HeadGradOut hg{};
  const float* h2 = nullptr;
  if (heads) {
    const float* w[8]; const float* b[8];
    hg.n_out = head_rows(n, w, hg.dw, b, hg.db);
    hg.g8 = n->g_all; hg.B = B;
    h2 = n->h2;
  }
  const double flops = 2.0 * B * (double)H * (S + 1 + (heads ? hg.n_out : 0));
  if (sp == 4) JH_LAUNCH_IDEM("jh_ppo_dw1_partial", flops, jh_ppo_dw1_partial_kernel<4>, grid, dim3(256), lds, st, B, H, S, rows_per, (const float*)n->dh1, d_x, d_idx, n->part_w1, h2, hg.g8, nj);
  else if (sp == 8) JH_LAUNCH_IDEM("jh_ppo_dw1_partial", flops, jh_ppo_dw1_partial_kernel<8>, grid, dim3(256), lds, st, B, H, S, rows_per, (const float*)n->dh1, d_x, d_idx, n->part_w1, h2, hg.g8, nj);
  else if (sp == 12) JH_LAUNCH_IDEM("jh_ppo_dw1_partial", flops, jh_ppo_dw1_partial_kernel<12>, grid, dim3(256), lds, st, B, H, S, rows_per, (const float*)n->dh1, d_x, d_idx, n->part_w1, h2, hg.g8, nj);
  else if (sp == 16) JH_LAUNCH_IDEM("jh_ppo_dw1_partial", flops, jh_ppo_dw1_partial_kernel<16>, grid, dim3(256), lds, st, B, H, S, rows_per, (const float*)n->dh1, d_x, d_idx, n->part_w1, h2, hg.g8, nj);
  else return jh_fail(JH_ERR_ARG, "dW1 reduction: observation width %d", S);
  JH_LAUNCH_CHECK();
  const int PS = S + 1 + (heads ? 8 : 0);
  const int n_comb = (H * PS + 255) / 256 + (heads ? 1 : 0);
  if (with_norm && n_rest + n_comb > kNormSlots) return jh_fail(JH_ERR_ARG, "dW1 reduction: %d norm slots", n_rest + n_comb);
  JH_LAUNCH(jh_ppo_dw1_combine_kernel, dim3((unsigned)n_comb), dim3(256), 0, st, H, S, slabs, (const float*)n->part_w1, n->grads + n->o_w1,
            n->grads + n->o_b1, hg, with_norm ? n->norm_partial + n_rest : nullptr, with_norm ? n->hyper : nullptr);
  JH_LAUNCH_CHECK();
  n->norm_slots = with_norm ? n_rest + n_comb : 0;
  return JH_OK;
}

static int pponet_l1(jh_pponet* n, int B, const float* d_x, const int64_t* d_idx, hipStream_t st) {
  const unsigned grid = (unsigned)(((B + kL1Rows - 1) / kL1Rows) * ((n->H + 255) / 256));
  JH_LAUNCH(jh_mlp_l1_kernel, dim3(grid), dim3(256), 0, st, B, n->S, n->H, d_x, d_idx,
            n->params + n->o_w1, n->params + n->o_b1, n->h1);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// Forward into the per-column-tile partial heads (n->fwd_part) + activations: the first launch of the
// minibatch update and of every no-grad pass.  Layer 1 is generated in registers for S <= 8.
static int pponet_forward_partials(jh_pponet* n, int B, const float* d_x, const int64_t* d_idx, hipStream_t st) {
  const PmbHeads hd = pmb_heads(n);
  // layer 1 generated inside the forward kernel up to 8 observations.  JH_PMB_GEN16=1 (round 6, tried): up to 16, through scalar LDS reads of the W1 rows -- correct
  // (the parity suites pass) and SLOWER at Hopper's S = 11: the forward launch goes from 15.5 to 31.4 us to save a 6.5 us layer-1 launch (36.8 vs 33.3 ms per iteration
  // of configs[4]'s per-GPU share), so the separate launch stays
  static const int gen_max = (getenv("JH_PMB_GEN16") && atoi(getenv("JH_PMB_GEN16")) == 1) ? 16 : 8;
  if (n->S <= gen_max) return jh_pmb_forward(n, B, d_x, d_idx, hd, nullptr, true, st);
  int rc = pponet_l1(n, B, d_x, d_idx, st);
  if (rc) return rc;
  return jh_pmb_forward(n, B, d_x, d_idx, hd, n->h1, true, st);
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
  // below kFwdTiledRows rows the fused forward (layer 2 + head partials in one latency-oriented launch) wins; from there on layer 2
  // belongs on the tile engine's LDS-DMA kernel (B = 2048: 41 + 6 us fused + finish vs GEMM + heads kernel, tools/bench_hopper.py)
  static const int kFwdTiledRows = getenv("JH_PPO_FWD_TILED_ROWS") ? atoi(getenv("JH_PPO_FWD_TILED_ROWS")) : 2048;
  if (jh_pmb_eligible(n, B) && B < kFwdTiledRows) {
    int rc = pponet_forward_partials(n, B, d_x, d_idx, st);
    if (rc) return rc;
    return jh_pmb_heads_finish(n, B, d_head0, d_head1, d_value, st);
  }
  int rc = pponet_l1(n, B, d_x, d_idx, st);
  if (rc) return rc;
  if (pponet_use_tiled(B)) {
    TGemm tg = mk_gemm(B, H, H, op_dense(OP_KCONT, n->h1, H), op_dense(OP_KCONT, n->params + n->o_w2, H), n->h2, H, TEPI_BIAS_RELU, n->params + n->o_b2);
    TGemmWorkspace tw;
    tw.ws = n->tg_ws; tw.ws_floats = n->tg_ws_floats; tw.cnt = n->tg_cnt; tw.cnt_slots = n->tg_cnt_slots;
    rc = jh_tgemm_launch(tw, "jh_tgemm_ppo_fwd_h2", &tg, 1, st);
  } else {
    GemmArgs g{};
    g.M = B; g.N = H; g.K = H; g.A = n->h1; g.lda = H; g.B = n->params + n->o_w2; g.ldb = H; g.C = n->h2; g.ldc = H;
    g.aux = n->params + n->o_b2;
    rc = launch_gemm<0, true, EPI_BIAS_RELU, false, 1, 1>("jh_gemm16_fwd_h2", g, st);
  }
  if (rc) return rc;
  HeadPtrs hp = head_ptrs(n, d_head0, d_head1, d_value, nullptr, nullptr, nullptr);
  for (int o0 = 0; o0 < n->n_out; o0 += 8) {  // one launch per 8 head outputs (Hopper / CartPole: one)
    JH_LAUNCH(jh_mlp_heads_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, B, H, n->h2, head_flat(n, hp, o0));
    JH_LAUNCH_CHECK();
  }
  return JH_OK;
}

// Backward of the LAST forward (same B, x, idx) given d(loss)/d(raw heads) as separate arrays: overwrites the
// flat gradient bucket.  (The PPO agent's minibatches of < kTiledRows rows go through jh_pponet_ppo_update
// instead, where the head gradients never leave the packed [B][8] form.)
static int pponet_backward(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_g_head0, const float* d_g_head1,
                           const float* d_g_value, const float* d_dv2, const float* d_mix, hipStream_t st, bool with_norm = false, const PpoFinish* fin = nullptr);
JH_EXPORT int jh_pponet_backward(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx,
                                 const float* d_g_head0, const float* d_g_head1, const float* d_g_value,
                                 jh_stream stream) {
  JH_ARG(n && d_x && d_g_head0 && d_g_value);
  JH_ARG(B > 0 && B <= n->max_rows);
  JH_ARG(!n->cont || d_g_head1);
  return pponet_backward(n, B, d_x, d_idx, d_g_head0, d_g_head1, d_g_value, nullptr, nullptr, jh_s(stream));
}
// d_dv2 / d_mix (both or neither): the value gradient is w1 d_g_value + w2 d_dv2 with {w1, w2} = d_mix, formed by the first kernel
static int pponet_backward(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_g_head0, const float* d_g_head1,
                           const float* d_g_value, const float* d_dv2, const float* d_mix, hipStream_t st, bool with_norm, const PpoFinish* fin) {
  const int H = n->H, S = n->S;
  n->norm_slots = 0;
  const int64_t bh = (int64_t)B * H;
  HeadPtrs hp = head_ptrs(n, nullptr, nullptr, nullptr, d_g_head0, d_g_head1, d_g_value);
  for (int o0 = 0; o0 < n->n_out; o0 += 8) {  // the chain over the outputs continues from launch to launch (dh2 holds it in between)
    const int ov = n->n_out - 1 - o0;  // the value head is the last output
    JH_LAUNCH(jh_mlp_heads_bwd_dh_kernel, dim3((unsigned)((bh / 4 + 255) / 256)), dim3(256), 0, st, B, H, n->h2, n->dh2,
              n->g_all, head_flat(n, hp, o0), n->gld, o0, o0 == 0 ? 1 : 0, o0 + 8 >= n->n_out ? 1 : 0,
              (d_dv2 && ov >= 0 && ov < 8) ? d_dv2 : nullptr, d_mix, ov, fin ? *fin : PpoFinish{});
    JH_LAUNCH_CHECK();
  }
  const float* w[kMaxHeadOutputs]; float* dw[kMaxHeadOutputs]; const float* b[kMaxHeadOutputs]; float* db[kMaxHeadOutputs];
  const int n_out = head_rows(n, w, dw, b, db);
  const int gld = n->gld;
  int rc;
  if (pponet_use_tiled(B)) {
    // dW2, dh1 and the head weight gradients only need dh2 / g_all: ONE grouped launch of the tiled MFMA GEMM
    // (split-K over the batch / the hidden width), then dW1 over the gathered observation rows (K = B is long)
    const int A = n->A;
    TGemm g[6];
    int ng = 0;
    (void)n_out;
    // dW2[o][i] = sum_b dh2[b][o] h1[b][i], db2 as the A-operand row sum
    g[ng++] = mk_gemm(H, H, B, op_dense(OP_XCONT, n->dh2, H), op_dense(OP_XCONT, n->h1, H), n->grads + n->o_w2, H, TEPI_NONE, nullptr, nullptr, 0, n->grads + n->o_b2);
    // dh1[b][i] = relu'(h1) * sum_o dh2[b][o] W2[o][i]
    g[ng++] = mk_gemm(B, H, H, op_dense(OP_KCONT, n->dh2, H), op_dense(OP_XCONT, n->params + n->o_w2, H), n->dh1, H, TEPI_MASK, nullptr, n->h1, H);
    // head weight gradients (g_all is [B][8] = head0 A cols | head1 A cols (continuous) | value): with the dW1 column reduction below
    // when that runs, else three more problems of this group
    static const bool kDw1Gemm = getenv("JH_PPO_DW1_GEMM") && atoi(getenv("JH_PPO_DW1_GEMM")) != 0;  // A/B: round 2's tile-engine form
    const bool reduce = !kDw1Gemm && pponet_dw1_reduce_fits(B, S);
    const bool reduce_heads = reduce && n->n_out <= 8;  // the column reduction carries 8 head columns; wider heads: three more problems of this group
    if (!reduce_heads) {
      g[ng++] = mk_gemm(A, H, B, op_dense(OP_XCONT, n->g_all, gld), op_dense(OP_XCONT, n->h2, H), n->grads + n->o_wh0, H, TEPI_NONE, nullptr, nullptr, 0, n->grads + n->o_bh0);
      int col = A;
      if (n->cont) {
        g[ng++] = mk_gemm(A, H, B, op_dense(OP_XCONT, n->g_all + col, gld), op_dense(OP_XCONT, n->h2, H), n->grads + n->o_wh1, H, TEPI_NONE, nullptr, nullptr, 0, n->grads + n->o_bh1);
        col += A;
      }
      g[ng++] = mk_gemm(1, H, B, op_dense(OP_XCONT, n->g_all + col, gld), op_dense(OP_XCONT, n->h2, H), n->grads + n->o_wv, H, TEPI_NONE, nullptr, nullptr, 0, n->grads + n->o_bv);
    }
    TGemmWorkspace tw;
    tw.ws = n->tg_ws; tw.ws_floats = n->tg_ws_floats; tw.cnt = n->tg_cnt; tw.cnt_slots = n->tg_cnt_slots;
    rc = jh_tgemm_launch(tw, "jh_tgemm_ppo_bwd", g, ng, st);
    if (rc) return rc;
    if (reduce) return pponet_dw1_reduce(n, B, d_x, d_idx, reduce_heads, st, with_norm);
    JH_LAUNCH(jh_rowgather_f32_kernel, dim3((unsigned)(((int64_t)B * S + 255) / 256)), dim3(256), 0, st, B, S, d_x, d_idx, n->xg);
    JH_LAUNCH_CHECK();
    g[0] = mk_gemm(H, S, B, op_dense(OP_XCONT, n->dh1, H), op_dense(OP_XCONT, n->xg, S), n->grads + n->o_w1, S, TEPI_NONE, nullptr, nullptr, 0, n->grads + n->o_b1);
    return jh_tgemm_launch(tw, "jh_tgemm_ppo_bwd_dW1", g, 1, st);
  }
  for (int o0 = 0; o0 < n_out; o0 += 8) {  // dWh[o][k] = sum_b g[b][o] h2[b][k] ; dbh[o] = sum_b g[b][o]   (A = g_all^T stored [K=B][gld]; 8 output rows per launch)
    GemmArgs g{};
    g.M = n_out - o0 < 8 ? n_out - o0 : 8; g.N = H; g.K = B; g.A = n->g_all + o0; g.lda = gld; g.B = n->h2; g.ldb = H;
    for (int o = 0; o < g.M; ++o) { g.rowptr[o] = dw[o0 + o]; g.rowsum_ptr[o] = db[o0 + o]; }
    rc = launch_gemm<1, false, EPI_ROWPTR, true, 1, 1>("jh_gemm16_bwd_dWheads", g, st);
    if (rc) return rc;
  }
  {  // dW2[o][i] = sum_b dh2[b][o] h1[b][i] ; db2[o] = sum_b dh2[b][o]  (A = dh2^T stored [K=B][M=H])
    GemmArgs g{};
    g.M = H; g.N = H; g.K = B; g.A = n->dh2; g.lda = H; g.B = n->h1; g.ldb = H; g.C = n->grads + n->o_w2; g.ldc = H;
    g.rowsum = n->grads + n->o_b2;
    rc = launch_gemm<1, false, EPI_NONE, true, 1, 1>("jh_gemm16_bwd_dW2", g, st);
    if (rc) return rc;
  }
  {  // dh1[b][i] = relu'(h1) * sum_o dh2[b][o] W2[o][i]                 (B = W2 stored [K=H_out][N=H_in])
    GemmArgs g{};
    g.M = B; g.N = H; g.K = H; g.A = n->dh2; g.lda = H; g.B = n->params + n->o_w2; g.ldb = H; g.C = n->dh1; g.ldc = H;
    g.aux = n->h1; g.ldaux = H;
    rc = launch_gemm<0, false, EPI_MASK, false, 1, 1>("jh_gemm16_bwd_dh1", g, st);
    if (rc) return rc;
  }
  {  // dW1[j][s] = sum_b dh1[b][j] x[r(b)][s] ; db1[j] = sum_b dh1[b][j]  (B = gathered x rows [K=B][N=S])
    GemmArgs g{};
    g.M = H; g.N = S; g.K = B; g.A = n->dh1; g.lda = H; g.B = d_x; g.ldb = S; g.b_rows = d_idx;
    g.C = n->grads + n->o_w1; g.ldc = S; g.rowsum = n->grads + n->o_b1;
    rc = launch_gemm<1, false, EPI_NONE, true, 1, 1>("jh_gemm16_bwd_dW1", g, st);
    if (rc) return rc;
  }
  return JH_OK;
}

static int pponet_adam(jh_pponet* n, float max_norm, float* d_norm_out, hipStream_t st, int n_partial = kNormBlocks) {
  JH_LAUNCH(jh_adam_kernel<false>, dim3(kNormBlocks), dim3(256), 0, st, n->n_params, n->params, n->grads, n->m, n->v,
            n->norm_partial, n_partial, n->hyper, max_norm, d_norm_out, (const float*)nullptr, 0, (int64_t)0);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// The minibatch update's last launch when the backward left sums of squares + (dW1 | db1) slabs behind (see the kernel).
static bool pponet_adam_fused_ok(const jh_pponet* n, int B) {
  static const bool off = getenv("JH_PMB_NO_FUSED_ADAM") != nullptr;
  const int64_t n_head = (int64_t)n->H * n->S + n->H;
  const int64_t tiles_m = (B + 15) / 16;
  // every workgroup re-reads all slabs: worth it while that is a fraction of a launch (CartPole: 2560 x 16 floats)
  return !off && tiles_m <= 16 && n_head * tiles_m <= 65536 && n_head / 4 <= (int64_t)kNormBlocks * 256;
}
static int pponet_adam_fused(jh_pponet* n, int B, float max_norm, float* d_norm_out, hipStream_t st) {
  const int t32 = n->H / 32;
  JH_LAUNCH(jh_adam_kernel<true>, dim3(kNormBlocks), dim3(256), 0, st, n->n_params, n->params, n->grads, n->m, n->v,
            n->ssq_part, t32 * t32 + t32, n->hyper, max_norm, d_norm_out, (const float*)n->part_w1, (B + 15) / 16,
            (int64_t)n->H * n->S + n->H);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// clip_grad_norm_(max_norm) (skipped when max_norm <= 0) + Adam step on the flat buckets.
// d_norm_out (optional, device float) receives the pre-clip global norm.
JH_EXPORT int jh_pponet_adam_step(jh_pponet* n, float max_norm, float* d_norm_out, jh_stream stream) {
  JH_ARG(n != nullptr);
  hipStream_t st = jh_s(stream);
  JH_LAUNCH(jh_gradnorm_kernel, dim3(kNormBlocks), dim3(256), 0, st, n->n_params, n->grads, n->norm_partial, n->hyper);
  JH_LAUNCH_CHECK();
  return pponet_adam(n, max_norm, d_norm_out, st);
}

int jh_ppo_loss_from_partials(jh_ctx* ctx, int continuous, int B, int A, const float* d_hpart, int tiles, int part_rows, int part_ld,
                              const int64_t* d_idx, const float* d_action, const float* d_adv, const float* d_ret,
                              const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef, float ent_coef,
                              float* d_g_all, float* d_stats, float* d_hyper_advance, float* d_defer_dv2, float* d_critic_sums, hipStream_t st);
struct jh_peer;
int jh_ppo_critic_select(int B, float* d_sums, float vf, float ent, float* d_gv, int ldv, const float* d_dv2, const float* d_stats_local,
                         float* d_stats_out, hipStream_t st, jh_peer* peer);

// One PPO minibatch update (ppo.py:122-169) in FOUR launches when the (dW1 | db1) slabs are small enough for Adam's
// prologue to sum them in every workgroup (B <= 256 rows, CartPole / Hopper widths), else five (jh_ppo_mb.hip): forward into partial heads,
// loss fwd+bwd -> packed head gradients, ONE backward grid (dh1 -> dW1/db1 partials | dW2/db2 | head weights),
// partial combine + global norm, clip + Adam.  do_adam == 0 stops after the backward with a complete gradient
// bucket (data-parallel: all-reduce, then jh_pponet_adam_step).
JH_EXPORT int jh_pponet_ppo_update(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_action,
                                   const float* d_adv, const float* d_ret, const float* d_value_old,
                                   const float* d_logp_old, float eps_clip, float vf_coef, float ent_coef, float max_norm,
                                   int32_t do_adam, float* d_stats, jh_stream stream) {
  JH_ARG(n && d_x && d_action && d_adv && d_ret && d_value_old && d_logp_old);
  JH_ARG(B > 0 && B <= 1024 && B <= n->max_rows);
  if (!jh_pmb_eligible(n, B)) return jh_fail(JH_ERR_ARG, "jh_pponet_ppo_update needs hidden_size %% 32 == 0 and at most 8 head outputs (H = %d, %d outputs: A + 1 discrete, 2 A + 1 continuous)", n->H, n->n_out);
  hipStream_t st = jh_s(stream);
  int rc = pponet_forward_partials(n, B, d_x, d_idx, st);
  if (rc) return rc;
  const bool fused = do_adam && pponet_adam_fused_ok(n, B);  // four launches: no combine + norm kernel
  rc = jh_ppo_loss_from_partials(n->ctx, n->cont, B, n->A, n->fwd_part, n->H / 16, n->max_rows, (n->cont ? 2 * n->A + 1 : n->A + 1) <= 4 ? 4 : 8, d_idx, d_action, d_adv, d_ret,
                                 d_value_old, d_logp_old, eps_clip, vf_coef, ent_coef, n->g_all, d_stats, fused ? n->hyper : nullptr, nullptr, nullptr, st);
  if (rc) return rc;
  rc = jh_pmb_backward(n, B, d_x, d_idx, pmb_heads(n), fused, st);
  if (rc) return rc;
  if (fused) return pponet_adam_fused(n, B, max_norm, nullptr, st);
  rc = jh_pmb_finalize(n, B, do_adam != 0, st);
  if (rc) return rc;
  return do_adam ? pponet_adam(n, max_norm, nullptr, st) : JH_OK;
}

// The same update for minibatches of ANY size up to max_rows -- what config.ppo.mujoco's 2048-row minibatches take (round 6) -- in one call:
//   forward (layer 1, layer 2 on the tile engine or the fused latency kernel, heads)            jh_pponet_forward
//   loss, forward AND backward, in ONE launch whatever B (jh_ppo_onepass_kernel: both critic branches' value gradients + {w1, w2})
//   backward (its first kernel forms the value gradient from the two branches), clip + Adam     jh_pponet_backward, jh_pponet_adam_step
// Bit-identical to the separate calls (jh_pponet_forward -> jh_ppo_loss_* -> jh_pponet_backward -> jh_pponet_adam_step): the partials are reduced
// in the same order, and w1 g1 + w2 g2 with w in {0, 1/2, 1} is what the two-pass kernels evaluate (tested).  One launch less per update.
int jh_ppo_loss_onepass(int continuous, int B, int A, const float* d_head0, const float* d_head1, const float* d_value_pred, const int64_t* d_idx,
                        const float* d_action, const float* d_adv, const float* d_ret, const float* d_value_old, const float* d_logp_old, float eps_clip,
                        float vf_coef, float ent_coef, float* d_g0, float* d_g1, float* d_gv, float* d_dv2, float* d_mix, unsigned* d_ticket, float* d_partial,
                        float* d_stats, hipStream_t st);
JH_EXPORT int jh_pponet_ppo_update_rows(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_action, const float* d_adv,
                                        const float* d_ret, const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef, float ent_coef,
                                        float max_norm, int32_t do_adam, float* d_stats, jh_stream stream) {
  JH_ARG(n && d_x && d_action && d_adv && d_ret && d_value_old && d_logp_old);
  JH_ARG(B > 0 && B <= n->max_rows);
  const size_t rows = (size_t)n->max_rows, A = (size_t)n->A;
  float* ws = n->upd_ws;
  float *h0 = ws, *h1 = h0 + rows * A, *hv = h1 + rows * A, *g0 = hv + rows, *g1 = g0 + rows * A, *gv = g1 + rows * A, *dv2 = gv + rows;
  float* partial = dv2 + rows;
  float* mix = partial + 8 * ((rows + 255) / 256);
  unsigned* ticket = reinterpret_cast<unsigned*>(mix + 8);
  int rc = jh_pponet_forward(n, B, d_x, d_idx, h0, n->cont ? h1 : nullptr, hv, stream);
  if (rc) return rc;
  hipStream_t st = jh_s(stream);
  // clip_grad_norm_'s sums of squares ride in the dW1 launches when this call also steps (nobody reduces the bucket in between); JH_PPO_NORM_FOLD=0: the norm kernel
  const bool fold = do_adam && max_norm > 0.f && !(getenv("JH_PPO_NORM_FOLD") && atoi(getenv("JH_PPO_NORM_FOLD")) == 0);
  if (B <= 1024) {  // one workgroup holds the whole minibatch: the fused forward + backward loss kernel IS one launch (and reduces in its own order)
    rc = n->cont ? jh_ppo_loss_continuous(n->ctx, B, n->A, h0, h1, hv, d_idx, d_action, d_adv, d_ret, d_value_old, d_logp_old, eps_clip, vf_coef, ent_coef, g0, g1, gv, d_stats, stream)
                 : jh_ppo_loss_discrete(n->ctx, B, n->A, h0, hv, d_idx, d_action, d_adv, d_ret, d_value_old, d_logp_old, eps_clip, vf_coef, ent_coef, g0, gv, d_stats, stream);
    if (rc) return rc;
    rc = pponet_backward(n, B, d_x, d_idx, g0, n->cont ? g1 : nullptr, gv, nullptr, nullptr, st, fold);
  } else {
    // up to 64 loss workgroups (16 384 rows): the backward's first kernel reduces their partials itself (no ticket, no tail in the loss launch) and writes the
    // statistics; beyond, the loss launch's last workgroup does (every workgroup of the consumer re-reducing thousands of partials would cost more).  The
    // value head sits in the consumer's LAST launch of 8 outputs.
    const int nb = (B + 255) / 256;
    const bool by_consumer = nb <= 64 && !(getenv("JH_PPO_LOSS_TICKET") && atoi(getenv("JH_PPO_LOSS_TICKET")) != 0);
    rc = jh_ppo_loss_onepass(n->cont, B, n->A, h0, n->cont ? h1 : nullptr, hv, d_idx, d_action, d_adv, d_ret, d_value_old, d_logp_old, eps_clip, vf_coef, ent_coef,
                             g0, n->cont ? g1 : nullptr, gv, dv2, mix, by_consumer ? nullptr : ticket, partial, by_consumer ? nullptr : d_stats, st);
    if (rc) return rc;
    PpoFinish fin{};
    if (by_consumer) { fin.partial = partial; fin.nb = nb; fin.B = B; fin.ent_count = n->cont ? B * n->A : B; fin.vf = vf_coef; fin.ent = ent_coef; fin.stats = d_stats; }
    rc = pponet_backward(n, B, d_x, d_idx, g0, n->cont ? g1 : nullptr, gv, dv2, mix, st, fold, by_consumer ? &fin : nullptr);
  }
  if (rc || !do_adam) return rc;
  if (n->norm_slots > 0) return pponet_adam(n, max_norm, nullptr, st, n->norm_slots);  // the backward's last two launches left the norm's sums of squares behind
  return jh_pponet_adam_step(n, max_norm, nullptr, stream);
}

// ---- the same update for DATA-PARALLEL learners with the reference's exact critic.  The critic of ppo.py:147-154 is
// max(mean(e1), mean(e2)) over the whole minibatch; with the minibatch sharded over ranks each rank's own max picks ITS branch and the
// averaged gradient is no longer one learner's once the value clamp binds.  Two halves around an 8-byte all-reduce:
//   jh_pponet_ppo_update_dp_begin   forward + loss: policy / entropy gradients complete, BOTH critic branches' value gradients kept
//                                   (g_all's value column | n->dv2), d_critic_sums[2] = this rank's {sum e1, sum e2}
//   (caller)                        all-reduce MEAN of d_critic_sums over the ranks
//   jh_pponet_ppo_update_dp_end     branch weights from the reduced sums (identical on every rank) -> value gradients -> the backward
//                                   grid + (dW1 | db1) combine: a complete gradient bucket.  d_stats: this rank's actor / entropy terms,
//                                   the GLOBAL critic terms.  Then: all-reduce MEAN of the bucket, jh_pponet_adam_step.
JH_EXPORT int jh_pponet_ppo_update_dp_begin(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_action, const float* d_adv,
                                            const float* d_ret, const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef,
                                            float ent_coef, float* d_critic_sums, jh_stream stream) {
  JH_ARG(n && d_x && d_action && d_adv && d_ret && d_value_old && d_logp_old && d_critic_sums);
  JH_ARG(B > 0 && B <= 1024 && B <= n->max_rows);
  if (!jh_pmb_eligible(n, B)) return jh_fail(JH_ERR_ARG, "jh_pponet_ppo_update_dp_begin needs hidden_size %% 32 == 0 and at most 8 head outputs (H = %d, %d outputs: A + 1 discrete, 2 A + 1 continuous)", n->H, n->n_out);
  hipStream_t st = jh_s(stream);
  int rc = pponet_forward_partials(n, B, d_x, d_idx, st);
  if (rc) return rc;
  return jh_ppo_loss_from_partials(n->ctx, n->cont, B, n->A, n->fwd_part, n->H / 16, n->max_rows, (n->cont ? 2 * n->A + 1 : n->A + 1) <= 4 ? 4 : 8, d_idx, d_action, d_adv,
                                   d_ret, d_value_old, d_logp_old, eps_clip, vf_coef, ent_coef, n->g_all, n->stats_tmp, nullptr, n->dv2, d_critic_sums, st);
}

static int pponet_dp_end(jh_pponet* n, jh_peer* peer, int32_t B, const float* d_x, const int64_t* d_idx, float* d_critic_sums, float vf_coef, float ent_coef,
                         float* d_stats, jh_stream stream, const char* who) {
  JH_ARG(n && d_x && d_critic_sums);
  JH_ARG(B > 0 && B <= 1024 && B <= n->max_rows);
  if (!jh_pmb_eligible(n, B))
    return jh_fail(JH_ERR_ARG, "%s needs hidden_size %% 32 == 0 and at most 8 head outputs (H = %d, %d outputs: A + 1 discrete, 2 A + 1 continuous)", who, n->H, n->n_out);
  hipStream_t st = jh_s(stream);
  const int vcol = n->cont ? 2 * n->A : n->A;  // the value head's column of g_all [B][gld]
  int rc = jh_ppo_critic_select(B, d_critic_sums, vf_coef, ent_coef, n->g_all + vcol, n->gld, n->dv2, n->stats_tmp, d_stats, st, peer);
  if (rc) return rc;
  rc = jh_pmb_backward(n, B, d_x, d_idx, pmb_heads(n), false, st);
  if (rc) return rc;
  return jh_pmb_finalize(n, B, false, st);
}

JH_EXPORT int jh_pponet_ppo_update_dp_end(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_critic_sums, float vf_coef,
                                          float ent_coef, float* d_stats, jh_stream stream) {
  return pponet_dp_end(n, nullptr, B, d_x, d_idx, const_cast<float*>(d_critic_sums), vf_coef, ent_coef, d_stats, stream, "jh_pponet_ppo_update_dp_end");
}

// The same second half with the ranks' exchange of {sum e1, sum e2} INSIDE its first launch (round 6; peer-pointer transport, B <= 256 rows per rank):
// d_critic_sums holds THIS rank's sums on entry and the ranks' mean on exit -- no collective launch between jh_pponet_ppo_update_dp_begin and here.
JH_EXPORT int jh_pponet_ppo_update_dp_end_peer(jh_pponet* n, jh_peer* peer, int32_t B, const float* d_x, const int64_t* d_idx, float* d_critic_sums, float vf_coef,
                                               float ent_coef, float* d_stats, jh_stream stream) {
  JH_ARG(peer != nullptr);
  return pponet_dp_end(n, peer, B, d_x, d_idx, d_critic_sums, vf_coef, ent_coef, d_stats, stream, "jh_pponet_ppo_update_dp_end_peer");
}

// Batched acting for W envs (PPO.act, ppo.py:55-69, discrete): ONE launch + host finish.
//   device: GEMM with layer 1 generated on the fly as the A operand (LDS), h2 kept in registers,
//           epilogue reduces each 16-column tile against the head weights and writes the partial
//           head outputs + a per-tile sequence word straight into device-mapped pinned host memory;
//   host:   polls the sequence words (bounded spin, falls back to hipStreamSynchronize), sums the
//           partials in tile order, softmax + inverse-CDF multinomial (argmax when !training) with a
//           counter-based splitmix64 stream.
// h_obs [W][S] and h_action [W] are ordinary host pointers; the call returns when the actions are
// there (acting is synchronous by nature: the envs need them).  h_logits_out / h_value_out optional.
// raw head outputs z [W][8] (flat output order: head0[A], head1[A] (continuous), value) of W observation rows
// zld: row stride of z_out (8, or n_out beyond 8 outputs)
static inline int pponet_zld(const jh_pponet* n) { return n->n_out <= 8 ? 8 : n->n_out; }
static int pponet_act_raw(jh_pponet* n, int32_t W, const float* h_obs, float* z_out, hipStream_t st) {
  const int H = n->H;
  if (n->n_out > 8) {
    // more than 8 head outputs (round 5): the separate-call forward writes the raw heads straight into device-mapped pinned memory;
    // three launches + a stream sync per timestep instead of one launch + a flag spin -- the wide action spaces run, slower
    memcpy(n->obs_pin_h, h_obs, sizeof(float) * (size_t)W * n->S);
    const int A = n->A;
    float* d0 = n->act_out_d;
    float* d1 = n->cont ? d0 + (size_t)W * A : nullptr;
    float* dv = d0 + (size_t)(n->cont ? 2 : 1) * W * A;
    int rc = jh_pponet_forward(n, W, n->obs_pin_d, nullptr, d0, d1, dv, (jh_stream)st);
    if (rc) return rc;
    JH_HIP(hipStreamSynchronize(st));
    const float *h0 = n->act_out_h, *h1 = h0 + (size_t)W * A, *hv = h0 + (size_t)(n->cont ? 2 : 1) * W * A;
    for (int wq = 0; wq < W; ++wq) {
      float* z = z_out + (size_t)n->n_out * wq;
      memcpy(z, h0 + (size_t)wq * A, sizeof(float) * A);
      if (n->cont) memcpy(z + A, h1 + (size_t)wq * A, sizeof(float) * A);
      z[n->n_out - 1] = hv[wq];
    }
    return JH_OK;
  }
  const int tiles_n = H / 16, tiles = ((W + 15) / 16) * tiles_n;
  // (tried: observations inline in the kernel-argument segment -- the 1 KB larger kernarg made every
  // launch slower than the one PCIe read it saved: 21.9 vs 18.8 us per timestep)
  memcpy(n->obs_pin_h, h_obs, sizeof(float) * (size_t)W * n->S);
  const float* w[8]; float* dw[8]; const float* b[8]; float* db[8];
  const int n_out = head_rows(n, w, dw, b, db);  // (<= 8 here)
  GemmArgs g{};
  g.M = W; g.N = H; g.K = H; g.B = n->params + n->o_w2; g.ldb = H; g.C = nullptr; g.aux = n->params + n->o_b2;
  g.x = n->obs_pin_d; g.x_rows = nullptr; g.W1 = n->params + n->o_w1; g.b1 = n->params + n->o_b1; g.S = n->S;
  for (int o = 0; o < n_out; ++o) { g.wh[o] = w[o]; g.hbias[o] = b[o]; }
  g.n_out = n_out; g.part = n->part_pin_d; g.part_rows = n->max_act_rows;
  const unsigned seq = ++n->act_seq;
  g.tile_flag = n->flag_pin_d; g.flag_seq = seq;
  int rc = launch_gemm<2, true, EPI_HEADPART, false, 1, 1>("jh_gemm16_act_fused", g, st);
  if (rc) return rc;
  // ---- wait for every tile's sequence word
  volatile unsigned* flags = n->flag_pin_h;
  bool all = false;
  for (long spin = 0; spin < 50000000L && !all; ++spin) {
    all = true;
    for (int t = 0; t < tiles; ++t)
      if (flags[t] != seq) { all = false; break; }
    if (!all) __builtin_ia32_pause();
  }
  if (!all) {
    JH_HIP(hipStreamSynchronize(st));
    for (int t = 0; t < tiles; ++t)
      if (flags[t] != seq) return jh_fail(JH_ERR_STATE, "acting kernel finished without publishing tile %d", t);
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  // ---- finish the heads on the host (sum over the column tiles, tile order)
  const float* part = n->part_pin_h;
  for (int wq = 0; wq < W; ++wq) {
    float* z = z_out + 8 * (size_t)wq;
    for (int o = 0; o < 8; ++o) z[o] = 0.f;
    for (int t = 0; t < tiles_n; ++t) {
      const float* p = part + ((size_t)t * n->max_act_rows + wq) * 8;
      for (int o = 0; o < n_out; ++o) z[o] += p[o];
    }
  }
  return JH_OK;
}

JH_EXPORT int jh_pponet_act_discrete(jh_pponet* n, int32_t W, const float* h_obs, int64_t* h_action,
                                     float* h_logits_out, float* h_value_out, int32_t training, jh_stream stream) {
  JH_ARG(n && h_obs && h_action);
  JH_ARG(!n->cont);
  JH_ARG(W > 0 && W <= n->max_act_rows);
  const size_t zld = (size_t)pponet_zld(n);
  std::vector<float> z(zld * (size_t)W);
  int rc = pponet_act_raw(n, W, h_obs, z.data(), jh_s(stream));
  if (rc) return rc;
  for (int wq = 0; wq < W; ++wq) {
    h_action[wq] = jh_sample_discrete(n, z.data() + zld * (size_t)wq, wq, training);
    if (h_logits_out) memcpy(h_logits_out + (size_t)wq * n->A, z.data() + zld * (size_t)wq, sizeof(float) * n->A);
    if (h_value_out) h_value_out[wq] = z[zld * (size_t)wq + n->A];
  }
  n->act_ctr += 1;
  return JH_OK;
}

// PPO.act for a continuous policy (ppo.py:55-63): h_action [W][A] = tanh(Normal(mu, std).sample()) (tanh(mu) when
// !training); h_mu_raw_out / h_log_std_raw_out [W][A] optional raw heads.
JH_EXPORT int jh_pponet_act_continuous(jh_pponet* n, int32_t W, const float* h_obs, float* h_action, float* h_mu_raw_out,
                                       float* h_log_std_raw_out, float* h_value_out, int32_t training, jh_stream stream) {
  JH_ARG(n && h_obs && h_action);
  JH_ARG(n->cont);
  JH_ARG(W > 0 && W <= n->max_act_rows);
  const size_t zld = (size_t)pponet_zld(n);
  std::vector<float> z(zld * (size_t)W);
  int rc = pponet_act_raw(n, W, h_obs, z.data(), jh_s(stream));
  if (rc) return rc;
  const int A = n->A;
  for (int wq = 0; wq < W; ++wq) {
    jh_sample_continuous(n, z.data() + zld * (size_t)wq, wq, training, h_action + (size_t)wq * A);
    if (h_mu_raw_out) memcpy(h_mu_raw_out + (size_t)wq * A, z.data() + zld * (size_t)wq, sizeof(float) * A);
    if (h_log_std_raw_out) memcpy(h_log_std_raw_out + (size_t)wq * A, z.data() + zld * (size_t)wq + A, sizeof(float) * A);
    if (h_value_out) h_value_out[wq] = z[zld * (size_t)wq + 2 * A];
  }
  n->act_ctr += 1;
  return JH_OK;
}
