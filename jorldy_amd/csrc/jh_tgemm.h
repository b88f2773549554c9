// Grouped LDS-tiled fp32 MFMA GEMM engine (jh_tgemm.hip) shared by the value networks (jh_rbnet.hip) and the
// PPO policy-value net's backward (jh_mlp.hip).
#pragma once
#include "jh_common.h"


enum { OP_KCONT = 0, OP_XCONT = 1, OP_NHWC_K = 2, OP_NHWC_X = 3, OP_NCHW_K = 4, OP_NCHW_X = 5 };
enum { TEPI_NONE = 0, TEPI_BIAS = 1, TEPI_BIAS_RELU = 2, TEPI_MASK = 3 };

// One GEMM operand, element (x, k): x = the row (A) / column (B) index of C, k = reduction index.
//   KCONT  p[x * ld + k]                      XCONT  p[k * ld + x]
//   NHWC_K im2col(pixel = x, tap = k)         NHWC_X im2col(pixel = k, tap = x)     tap = (ky, kx, c)
//   NCHW_K / NCHW_X the same on an NCHW image (uint8 or fp32, divided by 255: head.py:46), tap = (c, ky, kx)
struct Opnd {
  const void* p;
  int ld, mode, u8, vec;
  // im2col modes: element (pixel, tap) lives at p[pix_tab[pixel] + tap_tab[tap]] -- two small L2-resident
  // tables built once per layer geometry instead of six integer divisions per fetch
  const int* pix_tab;
  const int* tap_tab;
};

struct TGemm {
  int M, N, K;
  Opnd a, b;
  float* C;
  int ldc, epi;
  const float* bias;   // [N]
  const float* aux;    // MASK: forward activation, same indexing as C
  int ldaux;
  float* rowsum;       // optional [M]: sum_k A(m, k)  (bias gradients)
  // optional second output for the weight gradient of a NoisyNet layer with factorised noise (network/utils.py:60-70: W = mu + sig * f(e_in) f(e_out)^T):
  //   C2[m][n] = C[m][n] * (f(nz_n[n]) * f(nz_m[m]))   = d(sig) beside d(mu),     rowsum2[m] = rowsum[m] * f(nz_m[m])   = d(sig_b) beside d(mu_b)
  // rows m >= nz_split take the second pair of noise vectors (two layers stacked into one matrix: a1 | v1)
  float* C2;
  float* rowsum2;
  const float *nz_m, *nz_n, *nz_m2, *nz_n2;
  int nz_split;
  int splitk, tiles_m, tiles_n;
  int wg_begin;        // first workgroup of this problem in the linear grid (tiles x splits workgroups each)
  float* ws;           // split-K partials [splitk][tiles][BM*BN + BM]
  unsigned* cnt;       // [tiles] arrival counters (zero between launches)
};
constexpr int kMaxGroup = 6;
// A tile's split-K arrival counter has a 128-byte line to itself (round 5): device-scope atomics retire one after the other per line, and
// the counters of a launch's tiles used to sit in two or three lines that ALL of its workgroups hit within the same microsecond.
constexpr int kTgemmCntStride = 32;
struct TGemmBatch {
  TGemm p[kMaxGroup];
  int n;
  int xcd;  // LDS-DMA kernel: 1 = XCD-contiguous order of the linear grid (tgemm_xcd_order)
};

// uint8 / 255 correctly rounded without a division: q = b * (1/255), one Newton correction with exact
// remainders (verified == b / 255.0f for all 256 byte values; tests compare against the fp32-frame path)
__device__ __forceinline__ static float u8_unit(uint32_t b) {
  const float x = (float)b, r = 1.0f / 255.0f;
  const float q = x * r;
  return fmaf(fmaf(-q, 255.0f, x), r, q);
}

__device__ __forceinline__ static float jh_noise_f(float e) { return copysignf(sqrtf(fabsf(e)), e); }  // utils.py:66-67 (sign(0) = 0 either way)

// ---- host-side helpers
inline Opnd op_dense(int mode, const float* p, int ld) {
  Opnd o{};
  o.p = p; o.ld = ld; o.mode = mode;
  o.vec = (ld % 4 == 0) && (((uintptr_t)p & 15) == 0);
  return o;
}
struct ConvGeom {
  int C, H, W, OH, OW, KH, KW, S;
  const int* pix_tab = nullptr;  // device tables, see Opnd
  const int* tap_tab = nullptr;
};
inline Opnd op_conv(int mode, const void* p, int u8, const ConvGeom& c) {
  Opnd o{};
  o.p = p; o.mode = mode; o.u8 = u8; o.pix_tab = c.pix_tab; o.tap_tab = c.tap_tab;
  if (mode <= OP_NHWC_X) o.vec = (c.C % 4 == 0) && (((uintptr_t)p & 15) == 0);
  else o.vec = (c.KW % 4 == 0) && (c.W % 4 == 0) && (c.S % 4 == 0) && (((uintptr_t)p & (u8 ? 3 : 15)) == 0);  // every 4-tap piece aligned
  return o;
}
inline TGemm mk_gemm(int M, int N, int K, const Opnd& a, const Opnd& b, float* C, int ldc, int epi, const float* bias = nullptr,
                     const float* aux = nullptr, int ldaux = 0, float* rowsum = nullptr) {
  TGemm g{};
  g.M = M; g.N = N; g.K = K; g.a = a; g.b = b; g.C = C; g.ldc = ldc; g.epi = epi; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.rowsum = rowsum;
  return g;
}

// Workspace of the split-K hand-off: `ws` floats of partial tiles + zero-initialised arrival counters.
struct TGemmWorkspace {
  float* ws = nullptr;
  size_t ws_floats = 0;
  unsigned* cnt = nullptr;
  int cnt_slots = 0;   // arrival counters, kTgemmCntStride words apart (allocate cnt_slots * kTgemmCntStride words)
};
// The call sites of the engine: launch name "jh_tgemm_<NAME>" <-> kernel symbol jh_tgemm_kernel<TM, TN, ID> (what rocprofv3 sees;
// tools/rocprof_tgemm_names.py maps the IDs back).  A launch under a name that is not listed here is refused.
#define JH_TGEMM_TAGS(X) \
  X(dense, 0) X(conv1_fwd, 1) X(conv2_fwd, 2) X(conv3_fwd, 3) X(head_fwd, 4) X(fc_fwd, 5) X(stream1_fwd, 6) X(stream2_fwd, 7) \
  X(stream2_bwd, 8) X(stream1_bwd, 9) X(fc_bwd, 10) X(head_bwd, 11) X(conv3_bwd, 12) X(conv2_bwd, 13) X(conv1_bwd, 14)      \
  X(ppo_fwd_h2, 15) X(ppo_bwd, 16) X(ppo_bwd_dW1, 17)
// One launch for up to kMaxGroup independent problems (same tile shape, linear grid over (problem, tile, split)).
int jh_tgemm_launch(const TGemmWorkspace& w, const char* name, TGemm* probs, int n, hipStream_t st);
