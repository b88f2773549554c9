// Grouped LDS-tiled fp32 MFMA GEMM (16x16x4 MFMA): every contraction of the value networks (implicit-GEMM
// convolutions, linear layers, data / weight gradients) and the PPO net's backward GEMMs.  See jh_tgemm.h.
#include <stdlib.h>

#include <type_traits>

#include "jh_tgemm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DJH_TGEMM_TRACE (measurement builds, ab/lib_trace.so; never the shipped library): thread 0 of every workgroup of the LDS-DMA kernel
// stores the shader clock at its phase boundaries into a buffer set by jh_tgemm_trace_buffer: slot 0 entry, 1 operands described /
// tables staged, 2 + c chunk c's barrier passed (c < 24), 28 loop left, 29 epilogue operands + split hand-off done, 30 stores issued.
#ifdef JH_TGEMM_TRACE
__device__ unsigned long long* g_tgemm_trace = nullptr;
#define JH_TRACE(slot)                                                                                         \
  do {                                                                                                          \
    if (threadIdx.x == 0 && g_tgemm_trace) g_tgemm_trace[(size_t)blockIdx.x * 32 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define JH_TRACE(slot) do {} while (0)
#endif

namespace {

__device__ __forceinline__ float op_elem(const Opnd& o, int x, int k) {
  if (o.mode == OP_KCONT) return ((const float*)o.p)[(size_t)x * o.ld + k];
  if (o.mode == OP_XCONT) return ((const float*)o.p)[(size_t)k * o.ld + x];
  const bool kfast = !(o.mode & 1);
  const int off = o.pix_tab[kfast ? x : k] + o.tap_tab[kfast ? k : x];
  if (o.mode <= OP_NHWC_X) return ((const float*)o.p)[off];
  return o.u8 ? u8_unit(((const uint8_t*)o.p)[off]) : ((const float*)o.p)[off] / 255.0f;
}

// Rare cases (ragged edges, operands that do not allow 16-byte accesses, fp32 NCHW frames): element-wise, out
// of line so that the hot loop stays small and branch-free.
// k-fast modes: elements (x, k..k+3).  x-fast modes: elements (x..x+3, k).  Zero outside X x K.
__device__ __noinline__ float4 op_fetch4_slow(Opnd o, int x, int k, int X, int K) {
  const bool kfast = !(o.mode & 1);
  const int dx = kfast ? 0 : 1, dk = kfast ? 1 : 0;
  float4 r;
  r.x = (x < X && k < K) ? op_elem(o, x, k) : 0.f;
  r.y = (x + dx < X && k + dk < K) ? op_elem(o, x + dx, k + dk) : 0.f;
  r.z = (x + 2 * dx < X && k + 2 * dk < K) ? op_elem(o, x + 2 * dx, k + 2 * dk) : 0.f;
  r.w = (x + 3 * dx < X && k + 3 * dk < K) ? op_elem(o, x + 3 * dx, k + 3 * dk) : 0.f;
  return r;
}

// Everything behind a tile's MFMA loop, shared by the register-staged and the LDS-DMA kernel: split-K hand-off (partials + ticket,
// the last arriver sums them in split order), epilogue, stores.
struct TileCtx {
  int z, tiles, tile, m0, n0, t, lane, wid, r, kq, wm, wn;
};
// The epilogue's operands of this lane's output elements -- bias, activation mask, NoisyNet noise -- fetched as ONE batch right
// behind the MFMA loop, in front of the split-K hand-off whose own round trips they fly under (round 5).  In the store loop they sat
// inside `if (m < M && n < N)`: fetch, wait, store per element, up to 16 elements x 3 operands in series behind the last MFMA of a
// 13-22 us launch (tools/isa_chain.py).  Every fetch is unconditional from a clamped element; an epilogue that does not use an
// operand reads the output matrix itself (valid memory, value dropped), so that there is no branch in between.  Split-K workgroups
// that turn out not to be the last arriver fetched them for nothing (a few hundred bytes).  (Fetched BEFORE the MFMA loop they cost
// 36 VGPRs across it: the 512-row launches lost a resident workgroup per CU and 5-15 % -- profiles/r05_ab_tgemm_fetch_order.txt.)
// The activation mask (data gradients) and the noise factors (NoisyNet weight gradients) never meet in one problem: they share the
// slots x[] (36 -> 18 registers for a 64 x 64 tile; with both, the LDS-DMA kernel crossed 128 VGPRs = one resident workgroup less per CU).
//   mask:  x[(i TN + j) 4 + q]                                      noise:  x[i 4 + q] = e_out of row m,  x[TM 4 + s TN + j] = e_in of column n, set s
template <int TM, int TN>
struct EpiPre {
  static constexpr int NX = (TM * TN * 4 > TM * 4 + 2 * TN) ? TM * TN * 4 : TM * 4 + 2 * TN;
  float bias[TN], x[NX];
};
// The bias of this lane's TN output columns (BIAS / BIAS_RELU epilogues; else zeros, no fetch -- `has_bias` is uniform).  The LDS-DMA kernel
// calls it BEFORE its k loop (round 6): TN registers across the loop, and the forward launches' epilogue has no memory round trip left.
template <int TN>
__device__ __forceinline__ void tgemm_epi_bias(const TGemm& g, int n0, int wn, int r, float (&bias)[TN]) {
  const bool has_bias = g.epi == TEPI_BIAS || g.epi == TEPI_BIAS_RELU;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * 16 * TN + 16 * j + r;
    bias[j] = has_bias ? g.bias[n < g.N ? n : g.N - 1] : 0.f;
  }
}
template <int TM, int TN>
__device__ __forceinline__ EpiPre<TM, TN> tgemm_epi_prefetch(const TGemm& g, const TileCtx& c, const float* bias_pre) {
  EpiPre<TM, TN> e;
  constexpr int NX = EpiPre<TM, TN>::NX;
  const bool has_mask = g.epi == TEPI_MASK, has_c2 = g.C2 != nullptr;
  if (bias_pre) {
#pragma unroll
    for (int j = 0; j < TN; ++j) e.bias[j] = bias_pre[j];
  } else {
    tgemm_epi_bias<TN>(g, c.n0, c.wn, c.r, e.bias);
  }
  // (round 6: an epilogue without activation mask and without NoisyNet noise -- every forward launch -- fetches nothing here; it used to read NX
  // dummy elements of C so that the two users' fetches stayed branch-free: one more round trip in front of the stores.  Uniform branch.)
  if (!has_mask && !has_c2) {
#pragma unroll
    for (int s = 0; s < NX; ++s) e.x[s] = 0.f;
    return e;
  }
  const float* aux_p = has_mask ? g.aux : g.C;
  const int lda = has_mask ? g.ldaux : g.ldc;
  const float* n1 = has_c2 ? g.nz_n : g.C;
  const float* n2 = (has_c2 && g.nz_n2) ? g.nz_n2 : n1;
  const float* m1 = has_c2 ? g.nz_m : g.C;
  const float* m2 = (has_c2 && g.nz_m2) ? g.nz_m2 : m1;
  int nc[TN], mc[TM][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = c.n0 + c.wn * 16 * TN + 16 * j + c.r;
    nc[j] = n < g.N ? n : g.N - 1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = c.m0 + c.wm * 16 * TM + 16 * i + 4 * c.kq + q;
      mc[i][q] = m < g.M ? m : g.M - 1;
    }
#pragma unroll
  for (int s = 0; s < NX; ++s) {
    // the slot's address under either meaning (static indices after unrolling; a slot without a meaning re-reads slot 0's element)
    const int sa = s < TM * TN * 4 ? s : 0;
    const float* pa = aux_p + (size_t)mc[sa / (4 * TN)][sa % 4] * lda + nc[(sa / 4) % TN];
    const float* pn;
    if (s < TM * 4) pn = (mc[s / 4][s % 4] >= g.nz_split ? m2 : m1) + mc[s / 4][s % 4];
    else if (s < TM * 4 + TN) pn = n1 + nc[s - TM * 4];
    else if (s < TM * 4 + 2 * TN) pn = n2 + nc[s - TM * 4 - TN];
    else pn = m1 + mc[0][0];
    e.x[s] = *(has_c2 ? pn : pa);
  }
  return e;
}

// EPI = false: the kernel variant for groups of plain products (data gradients into the im2col buffer: K = 64, thousands of
// workgroups; the convolutions' weight gradients), chosen by the host per launch: no fetches and none of their address arithmetic.
// Unconditional, they cost conv2 / conv3's backward groups 7-9 % at B = 512; behind a run-time branch, the values join the common
// tail through copies = a wait for the fetches in front of the split-K hand-off (profiles/r05_ab_tgemm_fetch_order.txt).
template <int TM, int TN, bool EPI>
__device__ __forceinline__ void tgemm_finish(const TGemm& g, const TileCtx& c, f32x4 (&acc)[TM][TN], float (&rs)[TM], bool want_rs, int* s_last_p,
                                             const float* bias_pre = nullptr) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  EpiPre<TM, TN> ep;  // (!EPI: never read -- the host launches that variant only for groups without epilogue operands)
  if constexpr (EPI) {
    ep = tgemm_epi_prefetch<TM, TN>(g, c, bias_pre);
    asm volatile("" ::: "memory");  // issued here, not sunk to the stores
  }
  const int z = c.z, tiles = c.tiles, tile = c.tile, m0 = c.m0, n0 = c.n0, t = c.t, lane = c.lane, wid = c.wid, r = c.r, kq = c.kq, wm = c.wm, wn = c.wn;
  int& s_last = *s_last_p;
  auto epilogue = [&](float v, float bias, float aux) -> float {
    if (g.epi == TEPI_BIAS || g.epi == TEPI_BIAS_RELU) v += bias;
    if (g.epi == TEPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
    if (g.epi == TEPI_MASK) v = aux > 0.f ? v : 0.f;
    return v;
  };

  if (g.splitk > 1) {
    // Split-K hand-off without __threadfence(): on a multi-XCD part an agent-scope fence writes back / invalidates
    // the whole per-XCD L2, which costs more than the GEMM.  Partials are written and read with sc1 (agent-coherent)
    // 16-byte accesses instead; s_waitcnt vmcnt(0) makes sure this wave's partial stores have completed before its
    // workgroup takes a ticket.  Partials are stored fragment-major ([wave][tile i][tile j][lane][4]): a lane's
    // accumulator registers are one 16-byte store, and the last workgroup to arrive rebuilds ITS accumulators as
    // the sum over all splits in split order (deterministic) and falls through to the common epilogue.
    constexpr int PSTRIDE = BM * BN + BM;
    float* mine = g.ws + ((size_t)z * tiles + tile) * PSTRIDE;
    const int frag0 = (wid * TM * TN * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float* dst = mine + frag0 + (i * TN + j) * 256;
        // s_nop 1: a 16-byte store reads its data registers over several cycles and the compiler does not know that, so
        // whatever it schedules next (the v_accvgpr_read of the next fragment into the same temporaries) must not land on them.
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[i][j]) : "memory");
      }
      if (want_rs && wn == 0 && kq == 0)
        __hip_atomic_store(mine + BM * BN + wm * 16 * TM + 16 * i + r, rs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      const unsigned old = __hip_atomic_fetch_add(g.cnt + (size_t)tile * kTgemmCntStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)g.splitk - 1;
      if (s_last) __hip_atomic_store(g.cnt + (size_t)tile * kTgemmCntStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    const float* base = g.ws + (size_t)tile * PSTRIDE;
    const size_t zstride = (size_t)tiles * PSTRIDE;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int UNR = 4;  // UNR x G 16-byte loads in flight per lane
    // fragments in groups of G <= 4 (round 6: register-blocked tiles carry 8 fragments per lane; all of them x UNR splits in flight would be
    // 128 registers): the sum over the splits runs in split order within every group, so a fragment's bits do not depend on the grouping
    constexpr int FR = TM * TN, G = FR < 4 ? FR : 4;
    static_assert(FR % G == 0, "fragment groups");
#pragma unroll
    for (int q0 = 0; q0 < FR; q0 += G) {
    for (int sp = 0; sp < g.splitk; sp += UNR) {
      // The loads of one round and their wait are ONE asm statement with early-clobber outputs: an asm load's destination
      // counts as written when the statement ends, so with the wait in a later statement the compiler is free to copy or
      // reuse the registers while the data is still in flight (seen as 32 wrong elements in one fragment, once in a few
      // thousand launches, under the LDS-DMA kernel's register allocation).
      f32x4 part[UNR][G];
      const float* src[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) src[u] = base + (sp + u < g.splitk ? sp + u : g.splitk - 1) * zstride + frag0 + q0 * 256;  // clamped: uniform control flow
      static_assert(UNR == 4, "the asm below names four address registers");
#define JH_LD(d, a, off) "global_load_dwordx4 %" #d ", %" #a ", off offset:" #off " sc1\n\t"
      if constexpr (G == 1) {
        asm volatile(JH_LD(0, 4, 0) JH_LD(1, 5, 0) JH_LD(2, 6, 0) JH_LD(3, 7, 0) "s_waitcnt vmcnt(0)"
                     : "=&v"(part[0][0]), "=&v"(part[1][0]), "=&v"(part[2][0]), "=&v"(part[3][0])
                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3])
                     : "memory");
      } else if constexpr (G == 2) {
        asm volatile(JH_LD(0, 8, 0) JH_LD(1, 8, 1024) JH_LD(2, 9, 0) JH_LD(3, 9, 1024) JH_LD(4, 10, 0) JH_LD(5, 10, 1024) JH_LD(6, 11, 0)
                         JH_LD(7, 11, 1024) "s_waitcnt vmcnt(0)"
                     : "=&v"(part[0][0]), "=&v"(part[0][1]), "=&v"(part[1][0]), "=&v"(part[1][1]), "=&v"(part[2][0]), "=&v"(part[2][1]),
                       "=&v"(part[3][0]), "=&v"(part[3][1])
                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3])
                     : "memory");
      } else {
        static_assert(G == 4, "fragment groups of 1, 2 and 4 are spelled out");
        asm volatile(JH_LD(0, 16, 0) JH_LD(1, 16, 1024) JH_LD(2, 16, 2048) JH_LD(3, 16, 3072) JH_LD(4, 17, 0) JH_LD(5, 17, 1024)
                         JH_LD(6, 17, 2048) JH_LD(7, 17, 3072) JH_LD(8, 18, 0) JH_LD(9, 18, 1024) JH_LD(10, 18, 2048) JH_LD(11, 18, 3072)
                             JH_LD(12, 19, 0) JH_LD(13, 19, 1024) JH_LD(14, 19, 2048) JH_LD(15, 19, 3072) "s_waitcnt vmcnt(0)"
                     : "=&v"(part[0][0]), "=&v"(part[0][1]), "=&v"(part[0][2]), "=&v"(part[0][3]), "=&v"(part[1][0]), "=&v"(part[1][1]),
                       "=&v"(part[1][2]), "=&v"(part[1][3]), "=&v"(part[2][0]), "=&v"(part[2][1]), "=&v"(part[2][2]), "=&v"(part[2][3]),
                       "=&v"(part[3][0]), "=&v"(part[3][1]), "=&v"(part[3][2]), "=&v"(part[3][3])
                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3])
                     : "memory");
      }
#undef JH_LD
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
#pragma unroll
        for (int q = 0; q < G; ++q)
          if (sp + u < g.splitk) acc[(q0 + q) / TN][(q0 + q) % TN] += part[u][q];
      }
    }
    }
    if (want_rs && wn == 0 && kq == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v = 0.f;
        for (int sp = 0; sp < g.splitk; ++sp)
          v += __hip_atomic_load(base + sp * zstride + BM * BN + wm * 16 * TM + 16 * i + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rs[i] = v;
      }
    }
  }

  JH_TRACE(29);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 16 * TN + 16 * j + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // C/D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg
        const int m = m0 + wm * 16 * TM + 16 * i + 4 * kq + q;
        if (m < g.M && n < g.N) {
          const float v = epilogue(acc[i][j][q], ep.bias[j], ep.x[(i * TN + j) * 4 + q]);
          g.C[(size_t)m * g.ldc + n] = v;
          if (g.C2) {  // d(sig) = d(mu) * eps, eps = f(e_in[n]) * f(e_out[m]) (jh_rb_noisy_grad_kernel's expression)
            const bool second = m >= g.nz_split;
            g.C2[(size_t)m * g.ldc + n] = v * (jh_noise_f(second ? ep.x[TM * 4 + TN + j] : ep.x[TM * 4 + j]) * jh_noise_f(ep.x[i * 4 + q]));
          }
        }
      }
    }
    if (want_rs && wn == 0 && kq == 0) {
      const int m = m0 + wm * 16 * TM + 16 * i + r;
      if (m < g.M) {
        g.rowsum[m] = rs[i];
        if (g.rowsum2) g.rowsum2[m] = rs[i] * jh_noise_f((m >= g.nz_split ? g.nz_m2 : g.nz_m)[m]);
      }
    }
  }
}

// C[M][N] = sum_k A(m, k) B(k, n), workgroup tile (32 TM) x (32 TN), BK = 32, 4 waves as 2 x 2.
// LDS tiles are [x][k] with a 36-float row stride: a lane's MFMA operands for 4 consecutive k are ONE
// 16-byte LDS read (the k order inside a 16-wide block is permuted identically for A and B, which a
// sum over k does not see), and the 16 rows a wave reads start in 16 distinct 4-bank groups.
// Operand fetch: every lane owns fixed 4-element pieces of the A and B tiles.  When the operand allows
// 16-byte accesses and the extent that the 4 elements run along is a multiple of 4 (a uniform, per-operand
// test) a piece is either wholly inside or wholly outside the matrix: ONE load from a clamped address plus a
// select, no divergent branches; anything else goes through op_fetch4_slow.
// The grid is linear over (problem, tile, split); splitk > 1: every split writes its partial tile, the last
// workgroup to arrive (per-tile counter) sums the partials in split order -- deterministic, no atomics on
// data -- and runs the epilogue.
constexpr int kTabMax = 1024;  // k-range of one split that an im2col operand can address through its LDS table
// TAG: one kernel SYMBOL per call site (conv1_fwd, stream2_bwd, ...), so that rocprofv3's kernel trace and PMC passes
// attribute time and traffic per layer instead of to three shared `jh_tgemm_kernel<TM,TN>` symbols; the
// body does not depend on it.
template <int TM, int TN, int TAG, bool EPI = true>
__global__ void __launch_bounds__(256, 2) jh_tgemm_kernel(int hn, int hw1, int hw2, int hw3, int hw4, int hw5, int hxcd, int hpad, TGemmBatch batch) {
  constexpr int BM = 32 * TM, BN = 32 * TN, BK = 32, LD = 36;
  __shared__ __attribute__((aligned(16))) float sA[BM * LD];
  __shared__ __attribute__((aligned(16))) float sB[BN * LD];
  __shared__ int sTabA[kTabMax], sTabB[kTabMax];
  __shared__ int s_last;
  // the problem of this workgroup from the launch header: the first eight kernel arguments are scalars that gfx950 preloads into SGPRs at wave
  // launch (-mllvm -amdgpu-kernarg-preload-count=8, see the Makefile), so the descriptor below is the FIRST memory round trip of the workgroup
  // (round 6; through batch.p[i].wg_begin it was the second, behind a fetch of the six begins)
  static_assert(kMaxGroup == 6, "five begins in the header");
  (void)hpad;
  int pi = 0;
  if (1 < hn && (int)blockIdx.x >= hw1) pi = 1;
  if (2 < hn && (int)blockIdx.x >= hw2) pi = 2;
  if (3 < hn && (int)blockIdx.x >= hw3) pi = 3;
  if (4 < hn && (int)blockIdx.x >= hw4) pi = 4;
  if (5 < hn && (int)blockIdx.x >= hw5) pi = 5;
  const TGemm g = batch.p[pi];  // a COPY: the whole descriptor in one batch of scalar loads (through a reference, every field was its own dependent s_load + wait)
  const int tiles = g.tiles_m * g.tiles_n;
  const int local = blockIdx.x - g.wg_begin;
  const int z = local / tiles, tile = local - z * tiles;
  const int tm_blk = tile / g.tiles_n, tn_blk = tile - tm_blk * g.tiles_n;
  const int m0 = tm_blk * BM, n0 = tn_blk * BN;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 15, kq = lane >> 4, wm = wid & 1, wn = wid >> 1;
  const int nchunks = (g.K + BK - 1) / BK, per = (nchunks + g.splitk - 1) / g.splitk;
  const int kbeg = z * per * BK;
  int kend = kbeg + per * BK;
  if (kend > g.K) kend = g.K;
  const bool a_kfast = !(g.a.mode & 1), b_kfast = !(g.b.mode & 1);
  const bool a_conv = g.a.mode >= OP_NHWC_K, b_conv = g.b.mode >= OP_NHWC_K;
  // fast (branch-free) fetch possible?  fp32 NCHW frames always take the slow path (true division by 255)
  const bool a_fast = g.a.vec && !((a_kfast ? kend : g.M) & 3) && (g.a.mode < OP_NCHW_K || g.a.u8);
  const bool b_fast = g.b.vec && !((b_kfast ? kend : g.N) & 3) && (g.b.mode < OP_NCHW_K || g.b.u8);

  // per-slot invariants: position of the piece inside the tile, clamped coordinates, base offsets
  int a_x[TM], a_k[TM], a_off[TM], b_x[TN], b_k[TN], b_off[TN];
  // im2col operands: the offset term that follows k (taps for k-fast, pixels for x-fast) is staged in LDS for
  // this split's whole k range; the term that follows x is fixed per slot and sits in a register.
  // (round 5: one batch of kTabMax / 256 fetches from clamped addresses, unconditional stores -- see jh_tgemm_dma_kernel)
  if (a_conv || b_conv) {  // an operand without a table stages the other one's (never read)
    const int* ta = a_conv ? (a_kfast ? g.a.tap_tab : g.a.pix_tab) : (b_kfast ? g.b.tap_tab : g.b.pix_tab);
    const int* tb = b_conv ? (b_kfast ? g.b.tap_tab : g.b.pix_tab) : ta;
    const int tn = kend - kbeg;
    int tva[kTabMax / 256], tvb[kTabMax / 256];
#pragma unroll
    for (int u = 0; u < kTabMax / 256; ++u) {
      const int i = kbeg + (t + 256 * u < tn ? t + 256 * u : 0);
      tva[u] = ta[i];
      tvb[u] = tb[i];
    }
#pragma unroll
    for (int u = 0; u < kTabMax / 256; ++u) {
      sTabA[t + 256 * u] = tva[u];
      sTabB[t + 256 * u] = tvb[u];
    }
  }
  // the im2col lookups are issued unconditionally, all of them before the first use (a dense operand reads element 0 of its own
  // matrix and drops it): inside `conv ? tab[xc] : ...` each was a fetch + wait in a basic block of its own (round 5)
  int a_xc[TM], b_xc[TN], a_look[TM], b_look[TN];
  const int* a_xt = a_conv ? (a_kfast ? g.a.pix_tab : g.a.tap_tab) : (const int*)g.a.p;
  const int* b_xt = b_conv ? (b_kfast ? g.b.pix_tab : g.b.tap_tab) : (const int*)g.b.p;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int e = t + 256 * i;
    a_x[i] = a_kfast ? m0 + (e >> 3) : m0 + 4 * (e % (BM / 4));
    a_k[i] = a_kfast ? 4 * (e & 7) : e / (BM / 4);
    a_xc[i] = a_kfast ? (a_x[i] < g.M ? a_x[i] : g.M - 1) : (a_x[i] + 3 < g.M ? a_x[i] : (g.M >= 4 ? g.M - 4 : 0));
    a_look[i] = a_xt[a_conv ? a_xc[i] : 0];
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int e = t + 256 * i;
    b_x[i] = b_kfast ? n0 + (e >> 3) : n0 + 4 * (e % (BN / 4));
    b_k[i] = b_kfast ? 4 * (e & 7) : e / (BN / 4);
    b_xc[i] = b_kfast ? (b_x[i] < g.N ? b_x[i] : g.N - 1) : (b_x[i] + 3 < g.N ? b_x[i] : (g.N >= 4 ? g.N - 4 : 0));
    b_look[i] = b_xt[b_conv ? b_xc[i] : 0];
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = a_conv ? a_look[i] : (a_kfast ? a_xc[i] * g.a.ld : a_xc[i]);
#pragma unroll
  for (int i = 0; i < TN; ++i) b_off[i] = b_conv ? b_look[i] : (b_kfast ? b_xc[i] * g.b.ld : b_xc[i]);
  if (a_conv || b_conv) __syncthreads();

  // one piece: (operand, fast?, conv?, kfast?, x, in-tile k, slot offset, LDS table, extent X) at chunk k0
  auto fetch = [&](const Opnd& o, bool fast, bool conv, bool kfast, int x, int kin, int off, const int* tab, int X, int k0, float (&v)[4]) {
    const int k = k0 + kin;
    if (fast) {
      const int last = kfast ? kend - 4 : kend - 1;       // clamped: the load itself is always in bounds
      const int kc = k < last ? k : last;
      const bool ok = (kfast ? x < X : x + 3 < X) && k < kend;
      float4 q;
      if (!conv) {
        q = *reinterpret_cast<const float4*>((const float*)o.p + (kfast ? (size_t)off + kc : (size_t)kc * o.ld + off));
      } else if (o.mode <= OP_NHWC_X) {
        q = *reinterpret_cast<const float4*>((const float*)o.p + off + tab[kc - kbeg]);
      } else {
        const uint32_t w = *reinterpret_cast<const uint32_t*>((const uint8_t*)o.p + off + tab[kc - kbeg]);
        q = make_float4(u8_unit(w & 255u), u8_unit((w >> 8) & 255u), u8_unit((w >> 16) & 255u), u8_unit(w >> 24));
      }
      v[0] = ok ? q.x : 0.f; v[1] = ok ? q.y : 0.f; v[2] = ok ? q.z : 0.f; v[3] = ok ? q.w : 0.f;
    } else {
      const float4 q = op_fetch4_slow(o, x, k, X, kend);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
  };
  float ra[TM][4], rb[TN][4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fetch(g.a, a_fast, a_conv, a_kfast, a_x[i], a_k[i], a_off[i], sTabA, g.M, k0, ra[i]);
#pragma unroll
    for (int i = 0; i < TN; ++i) fetch(g.b, b_fast, b_conv, b_kfast, b_x[i], b_k[i], b_off[i], sTabB, g.N, k0, rb[i]);
  };
  // Round 5, the streamlined fetch for launches whose operands are both fp32 in 16-byte pieces (dense either way, NHWC im2col): the
  // addresses are SELECTED (no branch per mode), the TM + TN loads of a chunk are issued back to back, and the in-range selects are
  // applied when the registers go to LDS one iteration later.  Through fetch() every piece was a basic block of its own that ended
  // in the select on the loaded value: fetch A, wait, fetch B, wait -- in front of the chunk's MFMAs, so two dependent L2 round trips
  // per chunk and nothing in flight under the matrix cores (tools/isa_chain.py).
  const bool all_fast = a_fast && b_fast && g.a.mode <= OP_NHWC_X && g.b.mode <= OP_NHWC_X;
  bool oka[TM], okb[TN];
  auto gload_fast = [&](int k0) {
    auto piece = [&](const Opnd& o, bool conv, bool kfast, int x, int kin, int off, const int* tab, int X, float (&v)[4], bool& ok) {
      const int k = k0 + kin;
      const int last = kfast ? kend - 4 : kend - 1;  // clamped: the load itself is always in bounds
      const int kc = k < last ? k : last;
      ok = (kfast ? x < X : x + 3 < X) && k < kend;
      const int tv = tab[conv ? kc - kbeg : 0];
      const size_t e = conv ? (size_t)((ptrdiff_t)off + tv) : (kfast ? (size_t)off + kc : (size_t)kc * o.ld + off);
      const float4 q = *reinterpret_cast<const float4*>((const float*)o.p + e);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) piece(g.a, a_conv, a_kfast, a_x[i], a_k[i], a_off[i], sTabA, g.M, ra[i], oka[i]);
#pragma unroll
    for (int i = 0; i < TN; ++i) piece(g.b, b_conv, b_kfast, b_x[i], b_k[i], b_off[i], sTabB, g.N, rb[i], okb[i]);
  };
  auto mask_fast = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[i][j] = oka[i] ? ra[i][j] : 0.f;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) rb[i][j] = okb[i] ? rb[i][j] : 0.f;
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int xr = a_x[i] - m0;
      if (a_kfast) {
        *reinterpret_cast<float4*>(&sA[xr * LD + a_k[i]]) = make_float4(ra[i][0], ra[i][1], ra[i][2], ra[i][3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) sA[(xr + j) * LD + a_k[i]] = ra[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int xr = b_x[i] - n0;
      if (b_kfast) {
        *reinterpret_cast<float4*>(&sB[xr * LD + b_k[i]]) = make_float4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) sB[(xr + j) * LD + b_k[i]] = rb[i][j];
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = 0.f;
  const bool want_rs = g.rowsum != nullptr && tn_blk == 0;

  const TileCtx tc{z, tiles, tile, m0, n0, t, lane, wid, r, kq, wm, wn};
  // tile i: registers -> LDS, refill the registers with tile i + 1 (its HBM loads fly under the MFMAs of tile i)
  auto mainloop = [&](auto fast_tag) {
  constexpr bool FAST = decltype(fast_tag)::value;
  if (kbeg < kend) { if constexpr (FAST) gload_fast(kbeg); else gload(kbeg); }
#pragma unroll 1
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if constexpr (FAST) mask_fast();
    sstore();
    __syncthreads();
    if (k0 + BK < kend) { if constexpr (FAST) gload_fast(k0 + BK); else gload(k0 + BK); }
#pragma unroll
    for (int kb = 0; kb < BK; kb += 16) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 q = *reinterpret_cast<const float4*>(&sA[(wm * 16 * TM + 16 * i + r) * LD + kb + 4 * kq]);
        a[i][0] = q.x; a[i][1] = q.y; a[i][2] = q.z; a[i][3] = q.w;
        if (want_rs && wn == 0) rs[i] += (q.x + q.y) + (q.z + q.w);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(&sB[(wn * 16 * TN + 16 * j + r) * LD + kb + 4 * kq]);
        b[j][0] = q.x; b[j][1] = q.y; b[j][2] = q.z; b[j][3] = q.w;
      }
      // consecutive MFMAs go to different accumulators (dependent issue costs 40 cycles instead of 32)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][c], b[j][c], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  };
  if (all_fast) mainloop(std::true_type{}); else mainloop(std::false_type{});
  if (want_rs && wn == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      rs[i] += __shfl_xor(rs[i], 16, 64);
      rs[i] += __shfl_xor(rs[i], 32, 64);
    }
  }

  tgemm_finish<TM, TN, EPI>(g, tc, acc, rs, want_rs, &s_last);
}

// ---- the same 64 x 64 tile with its operands DMA-ed straight from global memory into LDS (global_load_lds_dwordx4: 16 bytes per lane,
// no VGPR destination, no ds_write issue slots) for launches whose operands are all fp32 in 16-byte pieces -- dense k- or x-contiguous,
// NHWC im2col with C % 4 == 0 in either orientation -- and K % 32 == 0: the value networks' forward GEMMs, their data and weight
// gradients from conv2 up, the PPO net's 2048-row forward.
// Round 2's SQ counters (profiles/r02_apex_pmc_tgemm.json): these launches live on occupancy -- waves parked in s_waitcnt / barriers
// 40-57 % of their cycles -- and every variant that took registers or LDS from the co-resident workgroups lost.  This one gives both
// back: 16 staging VGPRs and 4 ds_write_b128 (16 ds_write_b32 for an x-contiguous operand) per chunk and wave are gone, and with them
// the wait for the registers to fill before the LDS store; kDmaBufs 16 KB buffers (A | B, 32 k each) keep kDmaBufs - 1 chunks in flight
// under the MFMAs, with ONE barrier per chunk.
// LDS tiles are UNPADDED and lane-linear (the DMA writes 64 lanes x 16 bytes back to back):
//   k-contiguous operand  [x][32 k]: the 16-byte k-blocks of a row are XOR-swizzled with (x >> 1) & 7 on the GLOBAL side (lane p of an
//     instruction fetches block (p & 7) ^ ((x >> 1) & 7) and lands in block p & 7).  A ds_read_b128 is served 16 lanes = 16 rows at a
//     time from 16 slots of 16 bytes; with 128-byte rows, row x sits in slot 8 (x & 1) + block: the key (x >> 1) & 7 gives the 16 rows of
//     a group 16 distinct slots.  (The first version keyed on x & 7: rows x and x + 8 shared a slot -- every read 2-way conflicted.)
//   x-contiguous operand  [32 k][64 x]: the 16-byte x-blocks of a k row are XOR-swizzled with ((k >> 2) & 3) << 2; MFMA step c of a
//     16-wide k block takes k = 4 kq + c from lane group kq (the same order the k-contiguous float4 gives), so the four lane groups
//     read rows 4 apart, the swizzle is the per-lane constant kq << 2 and the 64 ds_read_b32 of a step hit 64 distinct banks.
// The DMA is inline asm on purpose: with __builtin_amdgcn_global_load_lds hipcc puts s_waitcnt vmcnt(0) in front of every s_barrier
// while a DMA is pending, so the next chunk never flies under this chunk's MFMAs (measured at Ape-X B = 512: stream1_fwd 150 us with
// the builtin, 121 us with the asm; conv2_fwd 111 -> 92, conv3_fwd 84 -> 66).  With asm the compiler does not count these loads at
// all; the kernel waits for them itself (vmcnt(0) before the first issue, then "all but the chunks issued after this one").
__device__ __forceinline__ void tgemm_dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// Measured at Ape-X B = 512 and Rainbow B = 32 (tools/probes/ab_apex_lib.sh, ab_rb_lib.sh): 2 and 3 buffers are the same to the
// noise WHEN three workgroups share a CU either way (3 x (48 + 4) KB still fit); 3 buffers at two workgroups per CU lose 5 %.
#ifndef JH_TGEMM_DMA_BUFS
#define JH_TGEMM_DMA_BUFS 2
#endif
constexpr int kDmaBufs = JH_TGEMM_DMA_BUFS;

// One operand's NP 16-byte pieces per lane and chunk: piece p = (i * 4 + wave) * 64 + lane of the 256 NP that make a (32 NP) x 32 tile
// (NP = 2: the 64-wide tile side; NP = 4: the 128-wide side of the register-blocked tiles of round 6).
template <int NP>
struct DmaOp {
  const float* src[NP];  // address of the piece in chunk 0 (im2col: without the term that follows k)
  int kin[NP];           // im2col: index of that term in the split's LDS table
  size_t step;           // dense: floats from one chunk to the next
  const int* tab;        // im2col: the LDS table (taps for k-contiguous, pixels for x-contiguous), else nullptr
};
// Two phases so that the im2col offset lookups of BOTH operands (and the table staging) are in flight together: phase 1 computes the
// pieces' coordinates and ISSUES the lookups -- unconditionally: a dense operand reads element 0 of its own matrix and drops it; a
// lookup inside `conv ? tab[xc] : ...` is a fetch + wait in its own basic block, four of them in series per launch (round 5) --,
// phase 2 builds the addresses.
template <int NP>
struct DmaLook {
  int xc[NP], kin[NP], look[NP];
};
template <int NP>
__device__ __forceinline__ DmaLook<NP> tgemm_dma_look(const Opnd& o, int X, int x0, int wid, int lane) {  // coordinates only; the lookups: tgemm_dma_lookups
  DmaLook<NP> l;
  constexpr int XB = 8 * NP;  // 16-byte x-blocks per k row of an x-contiguous tile
  const bool xfast = o.mode & 1;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = (i * 4 + wid) * 64 + lane;
    if (!xfast) {
      const int row = p >> 3, kb = 4 * ((p & 7) ^ ((row >> 1) & 7));
      const int x = x0 + row;
      l.xc[i] = x < X ? x : X - 1;  // rows beyond the matrix fetch a valid row: their results are never stored
      l.kin[i] = kb;
    } else {
      const int kk = p / XB, xb = (p % XB) ^ (((kk >> 2) & 3) << 2);
      const int x = x0 + 4 * xb;
      l.xc[i] = x + 3 < X ? x : X - 4;  // X % 4 == 0: a piece is wholly inside or wholly outside
      l.kin[i] = kk;
    }
    l.look[i] = 0;
  }
  return l;
}
// The im2col offset lookups of BOTH operands as one batch under ONE uniform branch (round 6: a problem without an im2col operand -- the
// linear layers, the PPO net -- fetches nothing here; its dummy lookups were a full memory round trip between the descriptor and the first
// DMA).  Inside the branch every fetch is unconditional: the dense operand of a problem with a view reads element 0 of its own matrix and drops it.
template <int NA, int NBP>
__device__ __forceinline__ void tgemm_dma_lookups(const Opnd& a, const Opnd& b, DmaLook<NA>& la, DmaLook<NBP>& lb) {
  const bool a_conv = a.mode >= OP_NHWC_K, b_conv = b.mode >= OP_NHWC_K;
  if (!(a_conv || b_conv)) return;
  const int* xa = a_conv ? ((a.mode & 1) ? a.tap_tab : a.pix_tab) : (const int*)a.p;
  const int* xb = b_conv ? ((b.mode & 1) ? b.tap_tab : b.pix_tab) : (const int*)b.p;
  int va[NA], vb[NBP];
#pragma unroll
  for (int i = 0; i < NA; ++i) va[i] = xa[a_conv ? la.xc[i] : 0];
#pragma unroll
  for (int i = 0; i < NBP; ++i) vb[i] = xb[b_conv ? lb.xc[i] : 0];
#pragma unroll
  for (int i = 0; i < NA; ++i) la.look[i] = va[i];
#pragma unroll
  for (int i = 0; i < NBP; ++i) lb.look[i] = vb[i];
}
template <int NP>
__device__ __forceinline__ DmaOp<NP> tgemm_dma_operand(const Opnd& o, const DmaLook<NP>& l, int kbeg, const int* tab) {
  DmaOp<NP> d;
  const bool xfast = o.mode & 1, conv = o.mode >= OP_NHWC_K;
  d.tab = conv ? tab : nullptr;
  d.step = xfast ? (size_t)32 * o.ld : 32;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    d.kin[i] = l.kin[i];
    if (!xfast) d.src[i] = (const float*)o.p + (conv ? (size_t)l.look[i] : (size_t)l.xc[i] * o.ld + kbeg + l.kin[i]);
    else d.src[i] = (const float*)o.p + (conv ? (size_t)l.look[i] : (size_t)(kbeg + l.kin[i]) * o.ld + l.xc[i]);
  }
  return d;
}
template <int NP>
__device__ __forceinline__ void tgemm_dma_issue(const DmaOp<NP>& d, int c, unsigned lds, bool is_b = false) {  // chunk c -> the wave's NP 1 KB pieces at lds
#ifdef JH_TGEMM_SKIP_B  // measurement builds only (wrong results): what is the loop's rate without operand B's / both operands' DMA traffic?
  if (is_b && c > 0) return;
#endif
#ifdef JH_TGEMM_SKIP_AB
  if (c > 0) return;
#endif
  const float* s[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) s[i] = d.tab ? d.src[i] + d.tab[c * 32 + d.kin[i]] : d.src[i] + c * d.step;
#pragma unroll
  for (int i = 0; i < NP; ++i) tgemm_dma16(s[i], lds + (unsigned)i * 4096u);
}

// The same issue in two halves for the pipelined loop: the im2col table lookups (LDS reads) of chunk c are issued early, among the
// fragment reads of a step, and the DMA instructions later take the offsets from registers.  Unconditional: a dense operand reads
// entry 0 of the (always allocated) table and drops it -- a lookup inside `tab ? ... : ...` is a branch + ds_read + s_waitcnt per piece.
template <int NP>
__device__ __forceinline__ void tgemm_dma_offsets(const DmaOp<NP>& d, const int* any_tab, int c, unsigned (&off)[NP]) {
  const int* t = d.tab ? d.tab : any_tab;
#pragma unroll
  for (int i = 0; i < NP; ++i) off[i] = (unsigned)t[d.tab ? c * 32 + d.kin[i] : 0];  // (table entries are >= 0; unsigned: a sign extension would be folded INTO the load = a wait right behind it)
}
template <int NP>
__device__ __forceinline__ void tgemm_dma_addrs(const DmaOp<NP>& d, int c, const unsigned (&off)[NP], const float* (&s)[NP]) {
  const bool conv = d.tab != nullptr;
#pragma unroll
  for (int i = 0; i < NP; ++i) s[i] = d.src[i] + (conv ? (size_t)off[i] : (size_t)c * d.step);
}
template <int NP>
__device__ __forceinline__ void tgemm_dma_fire(const float* const (&s)[NP], int c, unsigned lds, bool is_b = false) {
#ifdef JH_TGEMM_SKIP_B
  if (is_b && c > 0) return;
#endif
#ifdef JH_TGEMM_SKIP_AB
  if (c > 0) return;
#endif
#pragma unroll
  for (int i = 0; i < NP; ++i) tgemm_dma16(s[i], lds + (unsigned)i * 4096u);
}

// RS: this wave accumulates the row sums of A (bias gradients).  A template parameter, not a run-time flag: as a flag hipcc computes
// the sums in every wave and selects (36 VALU instructions per chunk next to 32 MFMAs: the forward launches lost 8 % to it).
// NB: LDS buffers per operand = chunks in flight + 1
// TM x TN: 16 x 16 fragments per wave.  2 x 2 is the 64 x 64 workgroup tile of rounds 2-5.  Round 6: 4 x 2 / 2 x 4 (128 x 64 / 64 x 128): a wave owns
// 64 x 32 of C, so a chunk's LDS fragments (TM + TN = 6 ds_read_b128 per 16 k) feed 32 MFMAs where the 2 x 2 tile's 4 reads feed 16, the workgroup
// runs 64 MFMAs per wave between two barriers instead of 32, and a chunk moves 24 KB into LDS for 2 x the flops of the 64 x 64 tile's 16 KB.
#ifndef JH_TGEMM_PIPELINED_LOOP
template <int TM, int TN, bool AX, bool BX, bool RS, int NB>
__device__ __forceinline__ void tgemm_dma_mainloop(const float* sA, const float* sB, const DmaOp<TM>& da, const DmaOp<TN>& db, int nc, int wid, int r, int kq,
                                                   int wm, int wn, f32x4 (&acc)[TM][TN], float (&rs)[TM], const TGemm& g, int n0, float (&bias)[TN]) {
  constexpr int BK = 32, XA = 32 * TM, XB = 32 * TN, TILE_A = XA * BK, TILE_B = XB * BK, LOADS = TM + TN;
  const unsigned ldsA = (unsigned)(uintptr_t)sA + (unsigned)wid * 1024u, ldsB = (unsigned)(uintptr_t)sB + (unsigned)wid * 1024u;
  // read offsets (floats) of this lane inside a tile; see the layouts above
  int ao[TM], bo[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int x = wm * 16 * TM + 16 * i + r;
    ao[i] = AX ? 4 * kq * XA + ((((x >> 2) ^ (kq << 2)) << 2) + (x & 3)) : x * BK;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int x = wn * 16 * TN + 16 * j + r;
    bo[j] = BX ? 4 * kq * XB + ((((x >> 2) ^ (kq << 2)) << 2) + (x & 3)) : x * BK;
  }
  // (the compiler's own loads so far -- tables, operand descriptors -- must not be counted by the vmcnt arithmetic below)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int c = 0; c < NB - 1 && c < nc; ++c) {
    tgemm_dma_issue<TM>(da, c, ldsA + (unsigned)c * (TILE_A * 4));
    tgemm_dma_issue<TN>(db, c, ldsB + (unsigned)c * (TILE_B * 4), true);
  }
  // the epilogue's bias, fetched HERE (round 6): behind the first chunks' DMA -- between two asm statements that clobber memory, so the compiler
  // neither hoists it in front of the DMA issue nor sinks it into the epilogue -- and in flight under the whole loop.  The vmcnt arithmetic below
  // stays safe: these TN loads are older than every DMA issued after them and vmcnt retires in order (a wait may cover them early, never late).
  tgemm_epi_bias<TN>(g, n0, wn, r, bias);
  asm volatile("" ::: "memory");
  int buf = 0, nbuf = NB - 1;  // buffer of chunk c, buffer of chunk c + NB - 1 (the one chunk c - 1 just left)
#pragma unroll 1
  for (int c = 0; c < nc; ++c) {
    // this wave's pieces of chunk c have landed once at most the chunks issued after it (LOADS loads each) are outstanding
    const int after = nc - 1 - c < NB - 2 ? nc - 1 - c : NB - 2;
    if (NB >= 4 && after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
    else if (NB >= 3 && after >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // ... and everybody else's; every wave is done with chunk c - 1, so its buffer may be refilled
    if (c < 24) JH_TRACE(2 + c);
    if (c + NB - 1 < nc) {
      tgemm_dma_issue<TM>(da, c + NB - 1, ldsA + (unsigned)nbuf * (TILE_A * 4));
      tgemm_dma_issue<TN>(db, c + NB - 1, ldsB + (unsigned)nbuf * (TILE_B * 4), true);
    }
    const float* A = sA + buf * TILE_A;
    const float* B = sB + buf * TILE_B;
#pragma unroll
    for (int kb = 0; kb < BK; kb += 16) {
      float a[TM][4], b[TN][4];
      const int blk = (((kb >> 2) + kq) ^ ((r >> 1) & 7)) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (AX) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) a[i][cc] = A[ao[i] + (kb + cc) * XA];
        } else {
          const float4 q = *reinterpret_cast<const float4*>(A + ao[i] + blk);
          a[i][0] = q.x; a[i][1] = q.y; a[i][2] = q.z; a[i][3] = q.w;
        }
        if (RS) rs[i] += (a[i][0] + a[i][1]) + (a[i][2] + a[i][3]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (BX) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) b[j][cc] = B[bo[j] + (kb + cc) * XB];
        } else {
          const float4 q = *reinterpret_cast<const float4*>(B + bo[j] + blk);
          b[j][0] = q.x; b[j][1] = q.y; b[j][2] = q.z; b[j][3] = q.w;
        }
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][cc], b[j][cc], acc[i][j], 0, 0, 0);
    }
    buf = buf + 1 == NB ? 0 : buf + 1;
    nbuf = nbuf + 1 == NB ? 0 : nbuf + 1;
  }
}

#else
// ---- round 6: the same loop, software-pipelined by hand.
// What the phase stamps of a trace build showed (tools/probes/tgemm_trace.py, profiles/r06_tgemm_phase_trace.txt): with ONE workgroup per CU
// (the PPO net's 2048-row products) the round-5 loop above takes 2044 cycles per 32-wide chunk for 1024 cycles of MFMA issue, and still
// 1680 with its DMA stream removed altogether: a wave reads a 16-k step's fragments from LDS, waits for them, issues the step's MFMAs,
// reads the next step's, waits again, meets the barrier, issues the next chunk's DMA (address arithmetic + four m0 round trips) and
// only then reads again -- every one of those sits IN FRONT of MFMA issue, and with one wave per SIMD nobody else fills the gap.
// Here the fragments are double-buffered in registers and every non-MFMA instruction of a step is issued in the shadow of MFMAs:
//   even step of chunk c   MFMA step 0 | LDS reads of the odd step                     | MFMA steps 1-3
//   odd step of chunk c    MFMA steps 0-1 | wait for chunk c + 1, barrier, LDS reads of its even step, DMA of chunk c + NB into the
//                          buffer chunk c just left (every wave's reads of it are complete at that barrier) | MFMA steps 2-3
// sched_barrier(0) pins the order (the MFMAs are register-only and would otherwise float across the asm statements).  All NB buffers
// hold chunks (the round-5 loop kept one free), the DMA prefetch distance stays NB - 1 chunks.  Same MFMA sequence per accumulator =
// the same bits.
template <int TM, int TN, bool AX, bool BX>
__device__ __forceinline__ void tgemm_lds_frags(const float* A, const float* B, const int (&ao)[TM], const int (&bo)[TN], int kb, int kq, int r,
                                                float (&a)[TM][4], float (&b)[TN][4]) {
  constexpr int XA = 32 * TM, XB = 32 * TN;
  const int blk = (((kb >> 2) + kq) ^ ((r >> 1) & 7)) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (AX) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) a[i][cc] = A[ao[i] + (kb + cc) * XA];
    } else {
      const float4 q = *reinterpret_cast<const float4*>(A + ao[i] + blk);
      a[i][0] = q.x; a[i][1] = q.y; a[i][2] = q.z; a[i][3] = q.w;
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if (BX) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) b[j][cc] = B[bo[j] + (kb + cc) * XB];
    } else {
      const float4 q = *reinterpret_cast<const float4*>(B + bo[j] + blk);
      b[j][0] = q.x; b[j][1] = q.y; b[j][2] = q.z; b[j][3] = q.w;
    }
  }
}
template <int TM, int TN>
__device__ __forceinline__ void tgemm_mfma_step(int cc, const float (&a)[TM][4], const float (&b)[TN][4], f32x4 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][cc], b[j][cc], acc[i][j], 0, 0, 0);
}

template <int TM, int TN, bool AX, bool BX, bool RS, int NB>
__device__ __forceinline__ void tgemm_dma_mainloop(const float* sA, const float* sB, const int* sTab, const DmaOp<TM>& da, const DmaOp<TN>& db, int nc, int wid, int r,
                                                   int kq, int wm, int wn, f32x4 (&acc)[TM][TN], float (&rs)[TM], const TGemm& g, int n0, float (&bias)[TN]) {
  constexpr int BK = 32, XA = 32 * TM, XB = 32 * TN, TILE_A = XA * BK, TILE_B = XB * BK, LOADS = TM + TN;
  static_assert(NB >= 2 && NB <= 4, "vmcnt immediates below");
  const unsigned ldsA = (unsigned)(uintptr_t)sA + (unsigned)wid * 1024u, ldsB = (unsigned)(uintptr_t)sB + (unsigned)wid * 1024u;
  // read offsets (floats) of this lane inside a tile; see the layouts above
  int ao[TM], bo[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int x = wm * 16 * TM + 16 * i + r;
    ao[i] = AX ? 4 * kq * XA + ((((x >> 2) ^ (kq << 2)) << 2) + (x & 3)) : x * BK;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int x = wn * 16 * TN + 16 * j + r;
    bo[j] = BX ? 4 * kq * XB + ((((x >> 2) ^ (kq << 2)) << 2) + (x & 3)) : x * BK;
  }
  // (the compiler's own loads so far -- tables, operand descriptors -- must not be counted by the vmcnt arithmetic below)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int c = 0; c < NB && c < nc; ++c) {
    tgemm_dma_issue<TM>(da, c, ldsA + (unsigned)c * (TILE_A * 4));
    tgemm_dma_issue<TN>(db, c, ldsB + (unsigned)c * (TILE_B * 4), true);
  }
  tgemm_epi_bias<TN>(g, n0, wn, r, bias);
  asm volatile("" ::: "memory");
  // chunk `want` has landed once at most the chunks issued after it are outstanding: min(issued - 1 - want, NB - 1) of them
  auto wait_landed = [&](int outstanding_chunks) {
    if (NB >= 4 && outstanding_chunks >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LOADS) : "memory");
    else if (NB >= 3 && outstanding_chunks == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
    else if (outstanding_chunks == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  float a0[TM][4], b0[TN][4], a1[TM][4], b1[TN][4];
  {
    const int issued = nc < NB ? nc : NB;
    wait_landed(issued - 1);
    __syncthreads();
    JH_TRACE(2);
    tgemm_lds_frags<TM, TN, AX, BX>(sA, sB, ao, bo, 0, kq, r, a0, b0);
  }
  int buf = 0;
#pragma unroll 1
  for (int c = 0; c < nc; ++c) {
    const float* A = sA + buf * TILE_A;
    const float* B = sB + buf * TILE_B;
    // ---- even step (k 0..15 of the chunk): fragments in a0 / b0
    tgemm_mfma_step<TM, TN>(0, a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    tgemm_lds_frags<TM, TN, AX, BX>(A, B, ao, bo, 16, kq, r, a1, b1);
    unsigned offa[TM], offb[TN];  // im2col offsets of the chunk whose DMA the odd step issues (clamped: the last chunks issue nothing)
    tgemm_dma_offsets<TM>(da, sTab, c + NB < nc ? c + NB : 0, offa);
    tgemm_dma_offsets<TN>(db, sTab, c + NB < nc ? c + NB : 0, offb);
    __builtin_amdgcn_sched_barrier(0);
    tgemm_mfma_step<TM, TN>(1, a0, b0, acc);
    tgemm_mfma_step<TM, TN>(2, a0, b0, acc);
    tgemm_mfma_step<TM, TN>(3, a0, b0, acc);
    if (RS) {
#pragma unroll
      for (int i = 0; i < TM; ++i) rs[i] += (a0[i][0] + a0[i][1]) + (a0[i][2] + a0[i][3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- odd step (k 16..31): fragments in a1 / b1
    tgemm_mfma_step<TM, TN>(0, a1, b1, acc);
    tgemm_mfma_step<TM, TN>(1, a1, b1, acc);
    // the next DMA's addresses on EVERY path (the table offsets are consumed here: left pending on the last chunks' path, their
    // destination registers cost the merged path a wait for ALL LDS reads in front of the MFMAs below)
    const float *pa[TM], *pb[TN];
    tgemm_dma_addrs<TM>(da, c + NB < nc ? c + NB : 0, offa, pa);
    tgemm_dma_addrs<TN>(db, c + NB < nc ? c + NB : 0, offb, pb);
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(pa[i]));  // (a use HERE: LLVM sinks the arithmetic into the branch that issues the DMA otherwise)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(pb[j]));
    __builtin_amdgcn_sched_barrier(0);
    const int nb = buf + 1 == NB ? 0 : buf + 1;
    if (c + 1 < nc) {
      // chunks issued so far: 0 .. min(nc, c + NB) - 1; chunk c + 1 is needed
      const int last_issued = (nc < c + NB ? nc : c + NB) - 1;
      wait_landed(last_issued - (c + 1));
      __syncthreads();  // everybody's pieces of chunk c + 1 have landed, and everybody's reads of chunk c are complete: its buffer may be refilled
      if (c + 1 < 24) JH_TRACE(3 + c);
      tgemm_lds_frags<TM, TN, AX, BX>(sA + nb * TILE_A, sB + nb * TILE_B, ao, bo, 0, kq, r, a0, b0);
      if (c + NB < nc) {
        tgemm_dma_fire<TM>(pa, c + NB, ldsA + (unsigned)buf * (TILE_A * 4));
        tgemm_dma_fire<TN>(pb, c + NB, ldsB + (unsigned)buf * (TILE_B * 4), true);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    tgemm_mfma_step<TM, TN>(2, a1, b1, acc);
    tgemm_mfma_step<TM, TN>(3, a1, b1, acc);
    if (RS) {
#pragma unroll
      for (int i = 0; i < TM; ++i) rs[i] += (a1[i][0] + a1[i][1]) + (a1[i][2] + a1[i][3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    buf = nb;
  }
}

#endif

// XCD-aware order of the linear grid (JH_TGEMM_XCD=1): workgroups are dealt to the 8 XCDs round-robin, and each XCD has its own L2.
// Workgroup b becomes item start(b % 8) + b / 8, where XCD x owns a CONTIGUOUS range of the (problem, tile, split) order -- neighbours in that
// order share A rows (same row tile) or the whole B operand (N = 64 convolutions), so an XCD's L2 fills with what its own CUs re-read.
__device__ __forceinline__ int tgemm_xcd_order(int bid, int total) {
  const int q = total >> 3, rem = total & 7, x = bid & 7, idx = bid >> 3;
  return x * q + (x < rem ? x : rem) + idx;
}

template <int TM, int TN, int TAG, bool EPI = true, int NB = (TM * TN > 4 ? 2 : kDmaBufs)>
__global__ void __launch_bounds__(256, (TM * TN >= 16 ? 2 : 3)) jh_tgemm_dma_kernel(int hn, int hw1, int hw2, int hw3, int hw4, int hw5, int hxcd, int hpad, TGemmBatch batch) {
  constexpr int BM = 32 * TM, BN = 32 * TN, BK = 32;
  __shared__ __attribute__((aligned(16))) float sA[NB * BM * BK];
  __shared__ __attribute__((aligned(16))) float sB[NB * BN * BK];
  __shared__ int sTab[kTabMax];  // at most one operand of a problem is an im2col view; 48 + 4 KB lets three workgroups share a CU
  __shared__ int s_last;
  JH_TRACE(0);
  const int bid = hxcd ? tgemm_xcd_order((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  // (the launch header: see jh_tgemm_kernel)
  (void)hpad;
  int pi = 0;
  if (1 < hn && bid >= hw1) pi = 1;
  if (2 < hn && bid >= hw2) pi = 2;
  if (3 < hn && bid >= hw3) pi = 3;
  if (4 < hn && bid >= hw4) pi = 4;
  if (5 < hn && bid >= hw5) pi = 5;
  const TGemm g = batch.p[pi];  // a COPY: the whole descriptor in one batch of scalar loads (through a reference, every field was its own dependent s_load + wait)
  const int tiles = g.tiles_m * g.tiles_n;
  const int local = bid - g.wg_begin;
  const int z = local / tiles, tile = local - z * tiles;
  const int tm_blk = tile / g.tiles_n, tn_blk = tile - tm_blk * g.tiles_n;
  const int m0 = tm_blk * BM, n0 = tn_blk * BN;
  const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 15, kq = lane >> 4, wm = wid & 1, wn = wid >> 1;
  const int nchunks = g.K / BK, per = (nchunks + g.splitk - 1) / g.splitk;
  const int kbeg = z * per * BK;
  int kend = kbeg + per * BK;
  if (kend > g.K) kend = g.K;
  const bool a_x = g.a.mode & 1, b_x = g.b.mode & 1;
  const bool a_conv = g.a.mode >= OP_NHWC_K, b_conv = g.b.mode >= OP_NHWC_K;
  // im2col operands: the offset term that follows k (taps for k-contiguous, pixels for x-contiguous) is staged in LDS for this
  // split's whole k range, as in the staged kernel
  // (round 5: all kTabMax / 256 passes fetched as ONE batch from clamped addresses and stored unconditionally -- entries past the
  // split's range are never read.  The plain loop compiled to fetch, wait, store per pass: up to four dependent round trips in front
  // of the first operand fetch of a launch that lasts 13-22 us.)
  DmaLook<TM> la = tgemm_dma_look<TM>(g.a, g.M, m0, wid, lane);
  DmaLook<TN> lb = tgemm_dma_look<TN>(g.b, g.N, n0, wid, lane);
  tgemm_dma_lookups<TM, TN>(g.a, g.b, la, lb);
  if (a_conv || b_conv) {
    const int* ktab = a_conv ? (a_x ? g.a.pix_tab : g.a.tap_tab) : (b_x ? g.b.pix_tab : g.b.tap_tab);
    const int tn = kend - kbeg;
    int tv[kTabMax / 256];
#pragma unroll
    for (int u = 0; u < kTabMax / 256; ++u) tv[u] = ktab[kbeg + (t + 256 * u < tn ? t + 256 * u : 0)];
#pragma unroll
    for (int u = 0; u < kTabMax / 256; ++u) sTab[t + 256 * u] = tv[u];
  }
  const DmaOp<TM> da = tgemm_dma_operand<TM>(g.a, la, kbeg, sTab);
  const DmaOp<TN> db = tgemm_dma_operand<TN>(g.b, lb, kbeg, sTab);
  if (a_conv || b_conv) __syncthreads();
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = 0.f;
  const bool want_rs = g.rowsum != nullptr && tn_blk == 0;
  const int nc = (kend - kbeg) / BK;
  const TileCtx tc{z, tiles, tile, m0, n0, t, lane, wid, r, kq, wm, wn};
  float bias_pre[TN];  // fetched inside the loop function, behind the first chunks' DMA issue
  JH_TRACE(1);
#ifndef JH_TGEMM_PIPELINED_LOOP
#define JH_DMA_LOOP(AX, BX, RS) tgemm_dma_mainloop<TM, TN, AX, BX, RS, NB>(sA, sB, da, db, nc, wid, r, kq, wm, wn, acc, rs, g, n0, bias_pre)
#else
#define JH_DMA_LOOP(AX, BX, RS) tgemm_dma_mainloop<TM, TN, AX, BX, RS, NB>(sA, sB, sTab, da, db, nc, wid, r, kq, wm, wn, acc, rs, g, n0, bias_pre)
#endif
  if (want_rs && wn == 0) {  // wave-uniform
    if (a_x) { if (b_x) JH_DMA_LOOP(true, true, true); else JH_DMA_LOOP(true, false, true); }
    else { if (b_x) JH_DMA_LOOP(false, true, true); else JH_DMA_LOOP(false, false, true); }
  } else {
    if (a_x) { if (b_x) JH_DMA_LOOP(true, true, false); else JH_DMA_LOOP(true, false, false); }
    else { if (b_x) JH_DMA_LOOP(false, true, false); else JH_DMA_LOOP(false, false, false); }
  }
#undef JH_DMA_LOOP
  if (want_rs && wn == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      rs[i] += __shfl_xor(rs[i], 16, 64);
      rs[i] += __shfl_xor(rs[i], 32, 64);
    }
  }
  JH_TRACE(28);
  tgemm_finish<TM, TN, EPI>(g, tc, acc, rs, want_rs, &s_last, EPI ? bias_pre : nullptr);
  JH_TRACE(30);
}

}  // namespace

// Can every problem of a launch take the LDS-DMA kernel?  fp32 operands in 16-byte pieces (dense either way, NHWC im2col either way),
// K a multiple of the chunk, x-contiguous operands with an extent that is a multiple of 4.
static bool tgemm_dma_ok(const TGemm* probs, int n) {
  static const bool off = getenv("JH_TGEMM_DMA") && atoi(getenv("JH_TGEMM_DMA")) == 0;
  if (off) return false;
  for (int i = 0; i < n; ++i) {
    const TGemm& g = probs[i];
    auto ok = [](const Opnd& o, int X) {
      if (o.mode > OP_NHWC_X || !o.vec || o.u8) return false;
      if (o.mode <= OP_XCONT && (o.ld & 3)) return false;
      return !(o.mode & 1) || (X >= 4 && !(X & 3));
    };
    if (!ok(g.a, g.M) || !ok(g.b, g.N) || (g.K & 31)) return false;
    if (g.a.mode >= OP_NHWC_K && g.b.mode >= OP_NHWC_K) return false;  // one LDS table
  }
  return true;
}

static int tgemm_tag_of(const char* name) {
#define JH_TGEMM_NAME(NAME, ID) \
  if (strcmp(name, "jh_tgemm_" #NAME) == 0) return ID;
  JH_TGEMM_TAGS(JH_TGEMM_NAME)
#undef JH_TGEMM_NAME
  return -1;
}

// Per-call-site tuning overrides (measurement runs; the defaults are tgemm_default_cfg below):
//   JH_TGEMM_CFG="<tag>:<TM>x<TN>[:s<splits>][:x<0|1>],..."   tag = the call site's ID of JH_TGEMM_TAGS or * for every site
//   e.g. JH_TGEMM_CFG="6:4x2:s4,2:4x2:x1"  (stream1_fwd on 128 x 64 tiles with 4 K splits, conv2_fwd on 128 x 64 tiles in XCD order)
struct TagCfg {
  int tm = 0, tn = 0, split = 0, xcd = -1;  // 0 / -1: not set
};
static TagCfg g_tag_cfg[32];
static bool g_tag_cfg_parsed = false;
static void tgemm_parse_cfg(const char* e) {
  for (int t = 0; t < 32; ++t) g_tag_cfg[t] = TagCfg{};
  g_tag_cfg_parsed = true;
  while (e && *e) {
    int tag = -1;
    if (*e == '*') { tag = -2; ++e; } else { tag = (int)strtol(e, (char**)&e, 10); }
    TagCfg c;
    while (*e == ':') {
      ++e;
      if (*e == 's') c.split = (int)strtol(e + 1, (char**)&e, 10);
      else if (*e == 'x') c.xcd = (int)strtol(e + 1, (char**)&e, 10);
      else { c.tm = (int)strtol(e, (char**)&e, 10); if (*e == 'x') c.tn = (int)strtol(e + 1, (char**)&e, 10); }
    }
    for (int t = 0; t < 32; ++t)
      if (tag == -2 || tag == t) {
        if (c.tm) { g_tag_cfg[t].tm = c.tm; g_tag_cfg[t].tn = c.tn; }
        if (c.split) g_tag_cfg[t].split = c.split;
        if (c.xcd >= 0) g_tag_cfg[t].xcd = c.xcd;
      }
    while (*e && *e != ',') ++e;
    if (*e == ',') ++e;
  }
}
static const TagCfg* tgemm_env_cfg() {
  if (!g_tag_cfg_parsed) tgemm_parse_cfg(getenv("JH_TGEMM_CFG"));
  return g_tag_cfg;
}
#ifdef JH_TGEMM_TRACE
JH_EXPORT int jh_tgemm_trace_buffer(unsigned long long* d_buf) {
  JH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tgemm_trace), &d_buf, sizeof(d_buf)));
  return JH_OK;
}
#endif
// Measurement / test hook: replace the overrides at run time (same grammar as JH_TGEMM_CFG; "" or null: none -- the defaults).  Not
// thread-safe against concurrent launches; launches already captured in a hipGraph keep the tiles they were captured with.
JH_EXPORT int jh_tgemm_set_cfg(const char* cfg) {
  tgemm_parse_cfg(cfg);
  return JH_OK;
}

// The defaults per call site and problem shape (round 6; measured: profiles/r06_tgemm_tile_sweep.txt).
static TagCfg tgemm_default_cfg(int tag, const TGemm* probs, int n) {
  TagCfg c;
  (void)tag; (void)probs; (void)n;
  return c;
}

// The call sites that have 128 x 64 / 64 x 128 instantiations of the LDS-DMA kernel (compile time: one kernel per site, tile and epilogue form)
#define JH_TGEMM_BIG_TAGS(X) X(dense, 0) X(conv2_fwd, 2) X(conv3_fwd, 3) X(fc_fwd, 5) X(stream1_fwd, 6) X(stream1_bwd, 9) X(fc_bwd, 10) \
  X(conv3_bwd, 12) X(conv2_bwd, 13) X(ppo_fwd_h2, 15) X(ppo_bwd, 16)
static bool tgemm_tag_has_big(int tag) {
#define JH_TGEMM_BIGQ(NAME, ID) if (tag == ID) return true;
  JH_TGEMM_BIG_TAGS(JH_TGEMM_BIGQ)
#undef JH_TGEMM_BIGQ
  return false;
}
template <int TM, int TN, int ID, bool EPI>
static void tgemm_dma_go(const char* name, double flops, dim3 grid, hipStream_t st, const TGemmBatch& batch) {
  JH_LAUNCH_IDEM(name, flops, (jh_tgemm_dma_kernel<TM, TN, ID, EPI>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch);
}

int jh_tgemm_launch(const TGemmWorkspace& net_w, const char* name, TGemm* probs, int n, hipStream_t st) {
  if (n < 1 || n > kMaxGroup) return jh_fail(JH_ERR_ARG, "tgemm group of %d", n);
  // A group whose heavy problems could take the LDS-DMA kernel but for some small ones that cannot (the PPO net's backward at 2048
  // rows: dW2 and dh1 at 1.07 GFLOP each next to three head weight gradients with M = 1 / 3 rows of the [B][8] loss gradient) goes as
  // two launches, heavy ones first: the second launch's ramp is cheaper than staging 2 GFLOP of operands through registers.
  if (n > 1) {
    static const double kSplitFlops = getenv("JH_TGEMM_SPLIT_GFLOP") ? atof(getenv("JH_TGEMM_SPLIT_GFLOP")) * 1e9 : 0.5e9;
    TGemm fast[kMaxGroup], rest[kMaxGroup];
    int nf = 0, nr = 0;
    double fast_flops = 0.0;
    for (int i = 0; i < n; ++i) {
      if (probs[i].M > 32 && probs[i].N > 32 && tgemm_dma_ok(&probs[i], 1)) {
        fast[nf++] = probs[i];
        fast_flops += 2.0 * probs[i].M * (double)probs[i].N * (double)probs[i].K;
      } else {
        rest[nr++] = probs[i];
      }
    }
    if (nf > 0 && nr > 0 && kSplitFlops > 0.0 && fast_flops >= kSplitFlops) {
      const int rc = jh_tgemm_launch(net_w, name, fast, nf, st);
      return rc ? rc : jh_tgemm_launch(net_w, name, rest, nr, st);
    }
  }
  int maxM = 0, maxN = 0;
  for (int i = 0; i < n; ++i) {
    if (probs[i].M > maxM) maxM = probs[i].M;
    if (probs[i].N > maxN) maxN = probs[i].N;
  }
  int TM = maxM <= 32 ? 1 : 2, TN = maxN <= 32 ? 1 : 2;
  // long-K, few-tile groups (B = 32: conv1's weight gradient M = 32, N = 256, K = 12 800; the fc layer 96 x 512, K = 3136): the
  // split-K tail is the LAST ARRIVER of each tile summing all splits' partial tiles alone (fc: 16 tiles x 16 splits x 16 KB).  32 x 32
  // tiles give 2-4 x as many tiles -- as many more arrivers, each with a fraction of the bytes to sum, at a lower split count per tile.
  // Measured (tools/probes/ab_tgemm_tiles.sh, Rainbow.learn() at config.rainbow.atari shapes): 0.428 -> 0.416 ms (K >= 4096, <= 8 tiles:
  // conv1 wgrad only) -> 0.4065 ms (K >= 2048, <= 32 tiles); <= 64 tiles / K >= 1024: no further change
  static const int kSmallTileK = getenv("JH_TGEMM_SMALL_TILE_K") ? atoi(getenv("JH_TGEMM_SMALL_TILE_K")) : 2048;
  static const int kSmallTileMax = getenv("JH_TGEMM_SMALL_TILE_MAXTILES") ? atoi(getenv("JH_TGEMM_SMALL_TILE_MAXTILES")) : 32;
  {
    int big_tiles = 0, min_k = 1 << 30;
    for (int i = 0; i < n; ++i) {
      big_tiles += ((probs[i].M + 32 * TM - 1) / (32 * TM)) * ((probs[i].N + 32 * TN - 1) / (32 * TN));
      if (probs[i].K < min_k) min_k = probs[i].K;
    }
    if (kSmallTileK > 0 && min_k >= kSmallTileK && big_tiles <= kSmallTileMax) TM = TN = 1;
  }
  static const long kDmaMask = getenv("JH_TGEMM_DMA_MASK") ? atol(getenv("JH_TGEMM_DMA_MASK")) : -1L;  // bit per call-site tag (debugging)
  const int tag_early = tgemm_tag_of(name);
  const bool use_dma = TM == 2 && TN == 2 && tgemm_dma_ok(probs, n) && tag_early >= 0 && ((kDmaMask >> tag_early) & 1L);
  // round 6: the register-blocked tiles of the LDS-DMA kernel (a wave owns 64 x 32 of C), per call site: the override of a measurement run, else the default
  const TagCfg ecfg = tag_early >= 0 ? tgemm_env_cfg()[tag_early] : TagCfg{};
  const TagCfg dcfg = tgemm_default_cfg(tag_early, probs, n);
  const int want_tm = ecfg.tm ? ecfg.tm : dcfg.tm, want_tn = ecfg.tm ? ecfg.tn : dcfg.tn;
  const int force_split = ecfg.split ? ecfg.split : dcfg.split;
  const int xcd_order = ecfg.xcd >= 0 ? ecfg.xcd : (dcfg.xcd > 0 ? 1 : 0);
  if (use_dma && tgemm_tag_has_big(tag_early) && ((want_tm == 4 && want_tn == 2) || (want_tm == 2 && want_tn == 4))) {
    TM = want_tm;
    TN = want_tn;
  }
  const int BM = 32 * TM, BN = 32 * TN;
  int max_tiles = 0;
  for (int i = 0; i < n; ++i) {
    probs[i].tiles_m = (probs[i].M + BM - 1) / BM;
    probs[i].tiles_n = (probs[i].N + BN - 1) / BN;
    const int t = probs[i].tiles_m * probs[i].tiles_n;
    if (t > max_tiles) max_tiles = t;
  }
  // fill the chip: ~2 workgroups per CU, but every split keeps at least two 32-wide K chunks
  static const int kTargetWgs = getenv("JH_TGEMM_TARGET_WGS") ? atoi(getenv("JH_TGEMM_TARGET_WGS")) : 256;
  static const int kMinChunks = getenv("JH_TGEMM_MIN_CHUNKS") ? atoi(getenv("JH_TGEMM_MIN_CHUNKS")) : 4;
  auto base_split = [&](int tiles, int nchunks) {
    // split K only when the tiles alone leave most of the 256 CUs idle; every split keeps >= kMinChunks chunks of 32
    int s = tiles >= kTargetWgs / 2 ? 1 : kTargetWgs / (tiles > 0 ? tiles : 1);
    if (s > nchunks / kMinChunks) s = nchunks / kMinChunks;
    return s < 1 ? 1 : s;
  };
  // Round 6, the launch's QUANTISATION: workgroups are dealt to 256 CUs, up to three resident on each, and a launch lasts as long as its
  // busiest CU.  Ape-X's stream-1 forward is 384 tiles of 98 chunks: half the CUs run two of them, half run one -- 119 us where three
  // 49-chunk workgroups on every CU take 94.  The launch's time by a model fitted to K sweeps of the engine (profiles/r06_tgemm_kslope.txt:
  // T = fixed + chunks x 0.645 us at one workgroup per CU, 0.55 us per workgroup-chunk at three):
  //   T(m) = 2.9 + rounds x 4.7 + 0.43 x ceil(W / 256) x (mean chunks per workgroup) / eff(resident) [+ the split-K hand-off's bytes], W = sum tiles x base x m
  // is evaluated for a common multiplier m of the base splits (long-K problems only) and the cheapest m taken.  A different split is a
  // different summation order: the parity suites bound what that may cost (1e-5 on every loss); JH_TGEMM_QUANT=0 restores round 5's splits.
  static const bool kQuant = !(getenv("JH_TGEMM_QUANT") && atoi(getenv("JH_TGEMM_QUANT")) == 0);
  int mult = 1;
  if (kQuant && force_split <= 0) {
    const double unit = 0.43 * (TM * TN) / 4.0;
    static const double eff[4] = {1.0, 0.66, 0.72, 0.78};
    double best = 1e30;
    for (int m = 1; m <= 4; ++m) {
      long W = 0;
      double chunk_work = 0.0, handoff_mb = 0.0;
      bool ok = true;
      for (int i = 0; i < n; ++i) {
        const int tiles = probs[i].tiles_m * probs[i].tiles_n, nchunks = (probs[i].K + 31) / 32;
        int sp = base_split(tiles, nchunks);
        if (m > 1 && nchunks >= 8 * kMinChunks) {  // only problems whose splits stay long
          if (sp * m > nchunks / kMinChunks) { ok = false; break; }
          sp *= m;
        }
        const int per = (nchunks + sp - 1) / sp;
        W += (long)tiles * sp;
        chunk_work += (double)tiles * sp * per;
        if (sp > 1) handoff_mb += (double)tiles * sp * (BM * BN * 4.0) * 1e-6;
      }
      if (!ok) break;
      const int per_cu = (int)((W + 255) / 256), rounds = (per_cu + 2) / 3;
      const double t = 2.9 + rounds * 4.7 + unit * per_cu * (chunk_work / (double)W) / eff[per_cu < 3 ? per_cu : 3] + (handoff_mb > 0 ? 1.0 + handoff_mb / 4.0 : 0.0);
      if (t < best * 0.97) {  // a larger split has to win by 3 %
        best = t;
        mult = m;
      }
    }
  }
  size_t ws_used = 0;
  int cnt_used = 0, max_split = 1;
  for (int i = 0; i < n; ++i) {
    const int tiles = probs[i].tiles_m * probs[i].tiles_n;
    const int nchunks = (probs[i].K + 31) / 32;
    int s = base_split(tiles, nchunks);
    if (mult > 1 && nchunks >= 8 * kMinChunks) s *= mult;
    if (force_split > 0) s = force_split < nchunks ? force_split : nchunks;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    const bool conv = probs[i].a.mode >= OP_NHWC_K || probs[i].b.mode >= OP_NHWC_K;
    const int min_s = conv ? (nchunks * 32 + kTabMax - 1) / kTabMax : 1;  // an im2col operand's k range must fit its LDS table
    if (s < min_s) s = min_s;
    const size_t pstride = (size_t)BM * BN + BM;
    while (s > min_s && (ws_used + (size_t)s * tiles * pstride > net_w.ws_floats || cnt_used + tiles > net_w.cnt_slots)) --s;
    if (s > 1 && (ws_used + (size_t)s * tiles * pstride > net_w.ws_floats || cnt_used + tiles > net_w.cnt_slots))
      return jh_fail(JH_ERR_NOMEM, "%s: split-K workspace too small (%d tiles x %d splits)", name, tiles, s);
    // no empty splits: shrink to the number of splits that actually own chunks
    if (s > 1) {
      const int per = (nchunks + s - 1) / s;
      s = (nchunks + per - 1) / per;
      if (conv && per * 32 > kTabMax) return jh_fail(JH_ERR_STATE, "%s: im2col k range %d exceeds the LDS table", name, per * 32);
    }
    probs[i].splitk = s;
    if (s > 1) {
      probs[i].ws = net_w.ws + ws_used;
      probs[i].cnt = net_w.cnt + (size_t)cnt_used * kTgemmCntStride;
      ws_used += (size_t)s * tiles * pstride;
      cnt_used += tiles;
    }
    if (s > max_split) max_split = s;
  }
  TGemmBatch batch{};
  int wgs = 0;
  for (int i = 0; i < n; ++i) {
    probs[i].wg_begin = wgs;
    wgs += probs[i].tiles_m * probs[i].tiles_n * probs[i].splitk;
    batch.p[i] = probs[i];
  }
  batch.n = n;
  batch.xcd = use_dma && xcd_order;
  (void)max_tiles; (void)max_split;
  const dim3 grid(wgs);
  double flops = 0.0;  // profiling: 2 M N K of every problem of the group (the launch is idempotent: partial slabs are
                       // rewritten, arrival counters return to zero)
  for (int i = 0; i < n; ++i) flops += 2.0 * probs[i].M * (double)probs[i].N * (double)probs[i].K;
  const int tag = tgemm_tag_of(name);
  // groups without epilogue operands at the call sites that have such groups: the variant without the operand fetches (tgemm_finish)
  bool plain = true;
  for (int i = 0; i < n; ++i) plain = plain && probs[i].epi == TEPI_NONE && probs[i].C2 == nullptr;
#define JH_TGEMM_PLAIN_TAGS(X) X(dense, 0) X(conv3_bwd, 12) X(conv2_bwd, 13) X(conv1_bwd, 14) X(ppo_bwd_dW1, 17)
  if (plain && use_dma && TM * TN == 8) {  // (the call sites that have both: the intersection of JH_TGEMM_PLAIN_TAGS and JH_TGEMM_BIG_TAGS)
#define JH_TGEMM_DMA_PLAIN_BIG(NAME, ID)                                              \
  case ID:                                                                            \
    if (TM == 4) tgemm_dma_go<4, 2, ID, false>(name, flops, grid, st, batch);         \
    else tgemm_dma_go<2, 4, ID, false>(name, flops, grid, st, batch);                 \
    JH_LAUNCH_CHECK();                                                                \
    return JH_OK;
    switch (tag) {
      JH_TGEMM_DMA_PLAIN_BIG(dense, 0) JH_TGEMM_DMA_PLAIN_BIG(conv3_bwd, 12) JH_TGEMM_DMA_PLAIN_BIG(conv2_bwd, 13)
      default: break;
    }
#undef JH_TGEMM_DMA_PLAIN_BIG
  } else if (plain && use_dma) {
#define JH_TGEMM_DMA_PLAIN(NAME, ID) \
  case ID: tgemm_dma_go<2, 2, ID, false>(name, flops, grid, st, batch); JH_LAUNCH_CHECK(); return JH_OK;
    switch (tag) {
      JH_TGEMM_PLAIN_TAGS(JH_TGEMM_DMA_PLAIN)
      default: break;
    }
#undef JH_TGEMM_DMA_PLAIN
  } else if (plain) {
#define JH_TGEMM_PLAIN(NAME, ID)                                                                                                 \
  case ID:                                                                                                                      \
    if (TM == 2 && TN == 2) JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<2, 2, ID, false>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch);      \
    else if (TM == 1 && TN == 2) JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<1, 2, ID, false>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch); \
    else if (TM == 2 && TN == 1) JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<2, 1, ID, false>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch); \
    else JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<1, 1, ID, false>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch);                         \
    JH_LAUNCH_CHECK();                                                                                                          \
    return JH_OK;
    switch (tag) {
      JH_TGEMM_PLAIN_TAGS(JH_TGEMM_PLAIN)
      default: break;
    }
#undef JH_TGEMM_PLAIN
  }
#undef JH_TGEMM_PLAIN_TAGS
  if (use_dma && TM * TN == 8) {
#define JH_TGEMM_DMA_BIG(NAME, ID)                                                  \
  case ID:                                                                          \
    if (TM == 4) tgemm_dma_go<4, 2, ID, true>(name, flops, grid, st, batch);        \
    else tgemm_dma_go<2, 4, ID, true>(name, flops, grid, st, batch);                \
    break;
    switch (tag) {
      JH_TGEMM_BIG_TAGS(JH_TGEMM_DMA_BIG)
      default: return jh_fail(JH_ERR_ARG, "tgemm launch name %s has no register-blocked kernel (jh_tgemm.hip: JH_TGEMM_BIG_TAGS)", name);
    }
#undef JH_TGEMM_DMA_BIG
    JH_LAUNCH_CHECK();
    return JH_OK;
  }
  if (use_dma) {
#define JH_TGEMM_DMA_CASE(NAME, ID) \
  case ID: tgemm_dma_go<2, 2, ID, true>(name, flops, grid, st, batch); break;
    switch (tag) {
      JH_TGEMM_TAGS(JH_TGEMM_DMA_CASE)
      default: return jh_fail(JH_ERR_ARG, "tgemm launch name %s has no kernel tag (jh_tgemm.h: JH_TGEMM_TAGS)", name);
    }
#undef JH_TGEMM_DMA_CASE
    JH_LAUNCH_CHECK();
    return JH_OK;
  }
#define JH_TGEMM_CASE(NAME, ID)                                                                                          \
  case ID:                                                                                                              \
    if (TM == 2 && TN == 2) JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<2, 2, ID>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch);      \
    else if (TM == 1 && TN == 2) JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<1, 2, ID>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch); \
    else if (TM == 2 && TN == 1) JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<2, 1, ID>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch); \
    else JH_LAUNCH_IDEM(name, flops, (jh_tgemm_kernel<1, 1, ID>), grid, dim3(256), 0, st, batch.n, batch.p[1].wg_begin, batch.p[2].wg_begin, batch.p[3].wg_begin, batch.p[4].wg_begin, batch.p[5].wg_begin, batch.xcd, 0, batch);                         \
    break;
  switch (tag) {
    JH_TGEMM_TAGS(JH_TGEMM_CASE)
    default: return jh_fail(JH_ERR_ARG, "tgemm launch name %s has no kernel tag (jh_tgemm.h: JH_TGEMM_TAGS)", name);
  }
#undef JH_TGEMM_CASE
  JH_LAUNCH_CHECK();
  return JH_OK;
}


// C-ABI entry for the dense modes (parity tests; also usable on its own): C[M][N] = A (.) B with a fused epilogue.
//   a_kcont != 0: A stored [M][K] (lda)   else stored [K][M] (transposed operand of a weight gradient)
//   b_kcont != 0: B stored [N][K] (ldb)   else stored [K][N]
//   epi 0 none | 1 + bias[n] | 2 relu(+ bias[n]) | 3 mask by aux[m][n] > 0;  d_rowsum (optional) [M] = sum_k A(m, k)
struct jh_tgemm_ws_holder {
  TGemmWorkspace w;
};
static thread_local jh_tgemm_ws_holder g_dense_ws[16];  // per calling thread AND per device (one process-global workspace served every device)

JH_EXPORT int jh_tgemm_dense(jh_ctx* ctx, int32_t M, int32_t N, int32_t K, const float* d_a, int32_t lda, int32_t a_kcont, const float* d_b, int32_t ldb,
                             int32_t b_kcont, float* d_c, int32_t ldc, int32_t epi, const float* d_bias, const float* d_aux, int32_t ldaux, float* d_rowsum,
                             jh_stream stream) {
  JH_ARG(ctx && d_a && d_b && d_c);
  JH_ARG(M > 0 && N > 0 && K > 0 && epi >= 0 && epi <= 3);
  JH_ARG((epi != 1 && epi != 2) || d_bias);
  JH_ARG(epi != 3 || d_aux);
  JH_HIP(hipSetDevice(ctx->device));
  JH_ARG(ctx->device >= 0 && ctx->device < 16);
  TGemmWorkspace& w = g_dense_ws[ctx->device].w;
  if (!w.ws) {  // one lazily created workspace per calling thread (split-K partials + arrival counters)
    w.ws_floats = (size_t)4 << 20;
    w.cnt_slots = 4096;
    JH_HIP(hipMalloc((void**)&w.ws, sizeof(float) * w.ws_floats));
    JH_HIP(hipMalloc((void**)&w.cnt, sizeof(unsigned) * (size_t)w.cnt_slots * kTgemmCntStride));
    JH_HIP(hipMemset(w.cnt, 0, sizeof(unsigned) * (size_t)w.cnt_slots * kTgemmCntStride));
  }
  TGemm g = mk_gemm(M, N, K, op_dense(a_kcont ? OP_KCONT : OP_XCONT, d_a, lda), op_dense(b_kcont ? OP_KCONT : OP_XCONT, d_b, ldb), d_c, ldc, epi, d_bias, d_aux, ldaux,
                    d_rowsum);
  return jh_tgemm_launch(w, "jh_tgemm_dense", &g, 1, jh_s(stream));
}

// Test entry: `n` (<= 6) independent k-contiguous problems C_j [M][N] = A_j [M][K] B_j [N][K]^T as ONE grouped launch (the shape of the
// value networks' forward launches: online / target trunks side by side).
JH_EXPORT int jh_tgemm_dense_group(jh_ctx* ctx, int32_t n, int32_t M, int32_t N, int32_t K, const float* const* d_a, const float* const* d_b, float* const* d_c,
                                   jh_stream stream) {
  JH_ARG(ctx && d_a && d_b && d_c && n >= 1 && n <= kMaxGroup && M > 0 && N > 0 && K > 0);
  JH_HIP(hipSetDevice(ctx->device));
  JH_ARG(ctx->device >= 0 && ctx->device < 16);
  TGemmWorkspace& w = g_dense_ws[ctx->device].w;
  if (!w.ws) {
    w.ws_floats = (size_t)4 << 20;
    w.cnt_slots = 4096;
    JH_HIP(hipMalloc((void**)&w.ws, sizeof(float) * w.ws_floats));
    JH_HIP(hipMalloc((void**)&w.cnt, sizeof(unsigned) * (size_t)w.cnt_slots * kTgemmCntStride));
    JH_HIP(hipMemset(w.cnt, 0, sizeof(unsigned) * (size_t)w.cnt_slots * kTgemmCntStride));
  }
  TGemm g[kMaxGroup];
  for (int j = 0; j < n; ++j) g[j] = mk_gemm(M, N, K, op_dense(OP_KCONT, d_a[j], K), op_dense(OP_KCONT, d_b[j], K), d_c[j], N, TEPI_NONE);
  return jh_tgemm_launch(w, "jh_tgemm_stream1_fwd", g, n, jh_s(stream));
}
