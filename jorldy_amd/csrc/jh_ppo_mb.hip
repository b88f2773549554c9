// The PPO minibatch update (core/agent/ppo.py:122-169) of the policy-value MLP in FIVE launches
// for minibatches of <= 1024 rows (config.ppo.cartpole: 256 rows, hidden 512):
//
//   1 jh_pmb_fwd_kernel     h2 = relu(relu(x[idx] W1^T + b1) W2^T + b2); layer 1 is GENERATED IN REGISTERS as the
//                           A operand (K = S <= 8 is 4-8 FMAs per element: no h1 round trip, no layer-1 launch);
//                           the epilogue reduces each 16-column tile against the head weights -> partial heads
//   2 jh_ppo_fused_kernel   (jh_ppo.hip) sums the partials in tile order, clipped loss fwd + bwd -> g_all [B][8]
//   3 jh_pmb_bwd_kernel     ONE grid, three roles: dh1 (+ the dW1 / db1 partial sums of its 16 rows: dh1 itself is
//                           never stored), dW2 / db2, head weight gradients.  dh2 = relu'(h2) * (g Wh) is generated
//                           in the operand fetch of both consumers instead of a dh2 kernel + 2 x 512 KB round trip
//   4 jh_pmb_norm_kernel    sums the dW1 / db1 partials in row-tile order (deterministic) + global-norm partials
//   5 jh_adam_kernel        (jh_mlp.hip) clip + Adam
// Minibatches of <= 256 rows skip launch 4: the backward's dW2 / head-weight workgroups also write the sum of squares of
// their tiles, and EVERY Adam workgroup sums the 16 (dW1 | db1) slabs itself in its prologue (jh_adam_kernel<true>:
// identical bits in all workgroups -> one clip coefficient).  Measured: Adam 6.9 -> 10.4 us, norm kernel 7.7 us gone.
//
// against eleven before (l1, GEMM, heads, loss, dh2, dW_heads, dW2, dh1, dW1, norm, Adam), each of which costs
// >= 4.5 us of launch ramp / drain at these sizes whatever it computes.
//
// Operand layouts on the fp32 MFMA (v_mfma_f32_16x16x4_f32; lane = (r = lane & 15, kq = lane >> 4) supplies
// A[m = r][k = kq], B[k = kq][n = r], holds C[4 kq + i][r]):
//   * k-contiguous operands: one 16-byte load = 4 consecutive k, element j feeds MFMA step j of BOTH operands
//     (the same permutation of K on both sides: the sum is unchanged);
//   * n- (or m-) contiguous operands (W2 in dh1 = dh2 W2; dh2^T, h1 in dW2 = dh2^T h1): the 16 lanes of a k
//     read 2 consecutive columns each (one 8-byte load) and use element t for COLUMN TILE t, i.e. tile t of a
//     32-wide workgroup tile owns the columns n0 + 2 r + t.  The interleave is undone for free in the epilogue
//     (a lane then holds 2 adjacent columns of a row: one 8-byte store).  This replaces the scalar 4-byte
//     operand loads that made dh1 the slowest kernel of round 1 (14 us, 4 x its algorithmic HBM traffic).
#include "jh_ppo_mb.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ============================================================================ 1. forward
struct PmbFwd {
  int M, H, S;
  const float* x;
  const int64_t* x_rows;
  const float *W1, *b1, *W2, *b2;
  const float* h1_in;  // !GEN: layer 1 precomputed [M][H]
  float* h1_out;       // GEN: column-tile 0 stores the generated layer 1 (the backward reads it)
  float* h2_out;       // nullable
  const float* wh[8];
  const float* hb[8];
  int n_out;
  float* part;  // [H/16][part_rows][part_ld]
  int part_rows, part_ld;
};

// SV: S == 4 * SV -> W1 rows / observations as float4 (SV = 1, 2); SV = 0: any S <= 8, scalar loads; SV = 3 (round 6): any S <= 16, scalar loads
// (config.ppo.mujoco's Hopper, S = 11: layer 1 used to be a launch of its own in front of this one)
template <int SV, bool GEN, int U>
__global__ void __launch_bounds__(256) jh_pmb_fwd_kernel(PmbFwd g) {
  constexpr bool VEC = SV == 1 || SV == 2;
  constexpr int XS = SV == 3 ? 16 : 8;
  __shared__ float s_acc[4][64][4];
  // GEN: each wave stages the W1 rows + biases of ITS K quarter in LDS once ([kper][S] | [kper] per wave).  Read straight
  // from global memory, hipcc fetches them chunk by chunk behind an s_waitcnt vmcnt(0) each: eight serial L2 round
  // trips (~5 us of the first version's 12.2 us) in front of the MFMAs
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
  const int K = g.H, tiles_n = K / 16;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * 16, n = tn * 16 + r;
  const int kper = ((K + 63) / 64) * 16, kbeg = wid * kper;
  const int kend = kbeg + kper < K ? kbeg + kper : K;
  const int m = m0 + r;
  const bool m_ok = m < g.M;
  const int mc = m_ok ? m : g.M - 1;
  // ---- fetch order (round 5, read off the ISA with tools/isa_chain.py): as first compiled this prologue was SEVEN dependent
  // L2 round trips in front of the first MFMA (row number -> observation, two W1 staging passes and two b1 passes each behind its
  // own s_waitcnt vmcnt(0), bias / head columns, W2).  Now every independent fetch of the workgroup is in flight before the first
  // wait: staging operands first (they are needed first and vmcnt retires in issue order), then the first batch of W2 rows and the
  // epilogue operands; the staging data is written to LDS while W2 is still on its way.  Same arithmetic, same bits.
  float xr[XS];
  int64_t row = mc;
  if (GEN && g.x_rows) row = g.x_rows[mc];
  // SV == 3: the staged W1 rows are padded to a multiple of four floats (zeros), so that the generation reads them as float4s like the S = 4 / 8 forms
  const int ws = SV == 3 ? ((g.S + 3) & ~3) : g.S;  // row stride of the staged W1 rows
  float* w1s = s_dyn + (size_t)wid * kper * (ws + 1);
  float* b1s = w1s + (size_t)kper * ws;
  const int kn = kend > kbeg ? kend - kbeg : 0;
  constexpr int WB = 4;  // W1 float4s per lane and staging pass (H = 512: S = 4 needs 2, S = 8 needs 4); scalars, not an array: with
                         // the scheduling barrier below an array stayed in scratch memory
  float4 w1t0 = make_float4(0.f, 0.f, 0.f, 0.f), w1t1 = w1t0, w1t2 = w1t0, w1t3 = w1t0, b1t = w1t0;
  const int n_el = kn * g.S;
  const int wi0 = lane * 4 < n_el ? lane * 4 : 0, wi1 = (lane + 64) * 4 < n_el ? (lane + 64) * 4 : 0;
  const int wi2 = (lane + 128) * 4 < n_el ? (lane + 128) * 4 : 0, wi3 = (lane + 192) * 4 < n_el ? (lane + 192) * 4 : 0;
  const int bi0 = 4 * lane < kn ? 4 * lane : 0;
  if (GEN && VEC) {  // (unconditional, clamped addresses: an empty wave -- H < 64 -- re-reads element 0)
    const int kb0 = kn > 0 ? kbeg : 0;
    const float* src = g.W1 + (size_t)kb0 * g.S;
    w1t0 = *reinterpret_cast<const float4*>(src + wi0);
    w1t1 = *reinterpret_cast<const float4*>(src + wi1);
    w1t2 = *reinterpret_cast<const float4*>(src + wi2);
    w1t3 = *reinterpret_cast<const float4*>(src + wi3);
    b1t = *reinterpret_cast<const float4*>(g.b1 + kb0 + bi0);
    __builtin_amdgcn_sched_barrier(0);  // keep them FIRST in the queue (the machine scheduler moved them behind the W2 rows)
  }
  if (GEN) {
    if (VEC) {
#pragma unroll
      for (int q = 0; q < SV; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(g.x + row * g.S + 4 * q);
        xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < XS; ++s) xr[s] = s < g.S ? g.x[row * g.S + s] : 0.f;
    }
  }
  // first batch of weight rows + the epilogue operands (bias, head weight columns): in flight behind the staging fetches
  float4 bw[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int kb = kbeg + 16 * u + 4 * kq;
    const int kc = kb < kend ? kb : (kbeg < K ? kbeg : 0);
    bw[u] = *reinterpret_cast<const float4*>(g.W2 + (size_t)n * K + kc);
  }
  const float b2 = g.b2[n];
  float whv[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) whv[o] = o < g.n_out ? g.wh[o][n] : 0.f;
  if (GEN) {
    if (VEC) {
      // UNCONDITIONAL stores: a lane beyond the slice fetched element 0 and rewrites it with the same bits -- behind an `if` LLVM
      // sinks each fetch into its store's block and the staging is four serial round trips again
      *reinterpret_cast<float4*>(w1s + wi0) = w1t0;
      *reinterpret_cast<float4*>(w1s + wi1) = w1t1;
      *reinterpret_cast<float4*>(w1s + wi2) = w1t2;
      *reinterpret_cast<float4*>(w1s + wi3) = w1t3;
      for (int i = (lane + 64 * WB) * 4; i < n_el; i += 256) *reinterpret_cast<float4*>(w1s + i) = *reinterpret_cast<const float4*>(g.W1 + (size_t)kbeg * g.S + i);  // wider nets
      *reinterpret_cast<float4*>(b1s + bi0) = b1t;
      for (int i = 4 * (lane + 64); i < kn; i += 256) *reinterpret_cast<float4*>(b1s + i) = *reinterpret_cast<const float4*>(g.b1 + kbeg + i);
    } else {
      const float* src = g.W1 + (size_t)kbeg * g.S;
      if (SV == 3) {
        for (int i = lane; i < n_el; i += 64) {
          const int rr = i / g.S, cc = i - rr * g.S;
          w1s[rr * ws + cc] = src[i];
        }
        for (int i = lane; i < kn * (ws - g.S); i += 64) {
          const int rr = i / (ws - g.S), cc = g.S + (i - rr * (ws - g.S));
          w1s[rr * ws + cc] = 0.f;
        }
      } else {
        for (int i = lane; i < n_el; i += 64) w1s[i] = src[i];
      }
      for (int i = lane; i < kn; i += 64) b1s[i] = g.b1[kbeg + i];
    }
    // wave-local hand-off: this wave's own ds_writes are ordered before its ds_reads
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool st_h1 = GEN && g.h1_out && tn == 0;
  for (int k0 = kbeg; k0 < kend; k0 += 16 * U) {
    float4 av[U], bwn[U];
    if (!GEN) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kb = k0 + 16 * u + 4 * kq;
        const int kc = kb < kend ? kb : kbeg;
        av[u] = *reinterpret_cast<const float4*>(g.h1_in + (size_t)mc * K + kc);
      }
    }
    const bool more = k0 + 16 * U < kend;
    if (more) {  // the next batch's weight rows travel while this one is consumed
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kb = k0 + 16 * U + 16 * u + 4 * kq;
        const int kc = kb < kend ? kb : kbeg;
        bwn[u] = *reinterpret_cast<const float4*>(g.W2 + (size_t)n * K + kc);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = k0 + 16 * u + 4 * kq;
      const int kc = kb < kend ? kb : kbeg;
      const bool ok = m_ok && kb < kend;
      if (GEN) {
        // layer 1 exactly as jh_mlp_l1_kernel: fmaf chain over s from 0, + bias, relu
        const float4 bb = *reinterpret_cast<const float4*>(b1s + (kc - kbeg));
        const float bj[4] = {bb.x, bb.y, bb.z, bb.w};
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = 0.f;
          if (VEC) {
#pragma unroll
            for (int q = 0; q < SV; ++q) {
              const float4 w = *reinterpret_cast<const float4*>(w1s + (size_t)(kc - kbeg + j) * g.S + 4 * q);
              t = fmaf(xr[4 * q], w.x, t); t = fmaf(xr[4 * q + 1], w.y, t); t = fmaf(xr[4 * q + 2], w.z, t); t = fmaf(xr[4 * q + 3], w.w, t);
            }
          } else if (SV == 3) {
            // (the pad columns hold 0 in xr AND in the staged row: fmaf(0, 0, t) == t, t is never -0 -- the chain over s < S is jh_mlp_l1_kernel's)
#pragma unroll
            for (int q = 0; q < XS / 4; ++q) {
              if (4 * q < ws) {
                const float4 w = *reinterpret_cast<const float4*>(w1s + (size_t)(kc - kbeg + j) * ws + 4 * q);
                t = fmaf(xr[4 * q], w.x, t); t = fmaf(xr[4 * q + 1], w.y, t); t = fmaf(xr[4 * q + 2], w.z, t); t = fmaf(xr[4 * q + 3], w.w, t);
              }
            }
          } else {
#pragma unroll
            for (int s = 0; s < XS; ++s)
              if (s < g.S) t = fmaf(xr[s], w1s[(size_t)(kc - kbeg + j) * g.S + s], t);
          }
          t += bj[j];
          a[j] = t > 0.f ? t : 0.f;
        }
        av[u] = make_float4(a[0], a[1], a[2], a[3]);
        if (st_h1 && ok) *reinterpret_cast<float4*>(g.h1_out + (size_t)m * K + kb) = av[u];
      }
      if (!ok) av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bw[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bw[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bw[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bw[u].w, acc, 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) bw[u] = bwn[u];
    }
  }
  // in-workgroup split-K combine, fixed wave order
#pragma unroll
  for (int i = 0; i < 4; ++i) s_acc[wid][lane][i] = acc[i];
  __syncthreads();
  if (wid != 0) return;
  float hv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mm = m0 + kq * 4 + i;
    float v = ((s_acc[0][lane][i] + s_acc[1][lane][i]) + s_acc[2][lane][i]) + s_acc[3][lane][i];
    v += b2;
    v = v > 0.f ? v : 0.f;
    hv[i] = mm < g.M ? v : 0.f;
    if (mm < g.M && g.h2_out) g.h2_out[(size_t)mm * K + n] = v;
  }
  // partial head outputs of this 16-column tile (reduced over the 16 lanes that share kq)
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (o >= g.n_out) break;
    const float w = whv[o];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float p = hv[i] * w;
      p += __shfl_xor(p, 1, 64);
      p += __shfl_xor(p, 2, 64);
      p += __shfl_xor(p, 4, 64);
      p += __shfl_xor(p, 8, 64);
      const int mm = m0 + kq * 4 + i;
      if (r == 0 && mm < g.M) {
        if (tn == 0) p += *g.hb[o];
        g.part[((size_t)tn * g.part_rows + mm) * g.part_ld + o] = p;
      }
    }
  }
}

// heads = sum of the per-tile partials in tile order (the order the loss kernel uses: the same bits)
__global__ void __launch_bounds__(256) jh_pmb_heads_finish_kernel(int M, int tiles, int part_rows, int part_ld, const float* __restrict__ part,
                                                                  int A, int cont, float* __restrict__ h0, float* __restrict__ h1,
                                                                  float* __restrict__ v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool wide = part_ld == 8;
  for (int t0 = 0; t0 < tiles; t0 += 8) {
    float4 q0[8], q1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u < tiles ? t0 + u : tiles - 1;
      const float4* q = reinterpret_cast<const float4*>(part + ((size_t)t * part_rows + i) * part_ld);
      q0[u] = q[0];
      if (wide) q1[u] = q[1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (t0 + u < tiles) {
        z[0] += q0[u].x; z[1] += q0[u].y; z[2] += q0[u].z; z[3] += q0[u].w;
        if (wide) { z[4] += q1[u].x; z[5] += q1[u].y; z[6] += q1[u].z; z[7] += q1[u].w; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < A) h0[(size_t)i * A + k] = z[k];
    if (cont && k >= A && k < 2 * A) h1[(size_t)i * A + (k - A)] = z[k];
    if (k == (cont ? 2 * A : A)) v[i] = z[k];
  }
}

// ============================================================================ 3. backward
struct PmbBwd {
  int B, H, S, n_out;
  const float* x;
  const int64_t* x_rows;
  const float *h1, *h2, *g_all, *W2;
  const float* wh[8];
  float* dW2;
  float* db2;
  float* dwh[8];
  float* dbh[8];
  float* part_w1;  // [ceil(B/16)][H*S + H]: per-row-tile partial sums of (dW1 | db1), flat-bucket order
  int n_dh1, n_dw2;  // workgroups of the first two roles (the rest: head weight gradients)
  float* ssq_part;   // nullable: [n_dw2 + H/32] sum of squares of what each dW2 / head-weight workgroup wrote
};

__device__ __forceinline__ float pmb_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- role: dh1 = relu'(h1) * (dh2 W2) for a 16-row x 32-column tile, reduced on the spot against the
// observation rows into partial dW1 / db1 (dh1 never reaches HBM)
template <int NO, int U, bool ROWS>
__device__ __forceinline__ void pmb_role_dh1(const PmbBwd& g, int blk, float (*s_acc)[4][64][4], float* s_wh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
  const int H = g.H, tiles_n = H / 32;
  const int tm = blk / tiles_n, tn = blk - tm * tiles_n;
  const int m0 = tm * 16, i0 = tn * 32;
  // ---- fetch order (round 5, tools/isa_chain.py): the head weight rows used to be staged by a loop that indexed the kernel
  // argument array g.wh[j] DYNAMICALLY -- hipcc fetched the pointer from the kernarg segment with a vector load, waited, fetched
  // the weight, waited, wrote it: sixteen serial round trips in front of the barrier every workgroup of this role sits behind, and
  // the main loop's forty fetches only started after it.  Now: one static 16-byte fetch per head row and thread, the gradient row,
  // wave 0's epilogue operands and the first batch of h2 / W2 rows are ALL issued before the first wait; the head rows (oldest in
  // the queue) go to LDS while the rest travels.
  const int m = m0 + r;
  const bool m_ok = m < g.B;
  const int mc = m_ok ? m : g.B - 1;
  // (4-byte fetches: the head rows sit behind A-element biases in the flat bucket and are not 16-byte aligned in general)
  constexpr int KW = 2;  // columns per thread and head row in the first pass (H <= 512)
  const int wk0 = threadIdx.x < H ? threadIdx.x : 0, wk1 = threadIdx.x + 256 < H ? threadIdx.x + 256 : 0;
  float whs[NO][KW];
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    whs[j][0] = whs[j][1] = 0.f;
    if (j < g.n_out) { whs[j][0] = g.wh[j][wk0]; whs[j][1] = g.wh[j][wk1]; }
  }
  float gj[NO];
  {
    const float4 a = *reinterpret_cast<const float4*>(g.g_all + (size_t)mc * 8);
    gj[0] = a.x; gj[1] = a.y; gj[2] = a.z; gj[3] = a.w;
    if (NO > 4) {
      const float4 b = *reinterpret_cast<const float4*>(g.g_all + (size_t)mc * 8 + 4);
      gj[4] = b.x; gj[5] = b.y; gj[6] = b.z; gj[7] = b.w;
    }
  }
  const int kper = ((H + 63) / 64) * 16, kbeg = wid * kper;
  const int kend = kbeg + kper < H ? kbeg + kper : H;
  // epilogue operands of wave 0 (relu'(h1) mask rows, observation rows) fetched before the main loop: behind
  // the split-K barrier they would be dependent L2 round trips on the critical path
  // (every wave fetches them, not only wave 0: behind `if (wid == 0)` the block ends in copies of the fetched values, i.e. in a
  // wait for the whole queue in front of the batch below)
  float2 hmask[4];
  int64_t xrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mm = m0 + kq * 4 + i;
    const int mmc = mm < g.B ? mm : g.B - 1;
    hmask[i] = *reinterpret_cast<const float2*>(g.h1 + (size_t)mmc * H + i0 + 2 * r);
    xrow[i] = ROWS ? g.x_rows[mmc] : (int64_t)mmc;  // (compile-time: a run-time select ends in copies = a wait for the queue)
  }
  float4 h2v[U];
  float2 bw[U][4];
#pragma unroll
  for (int u = 0; u < U; ++u) {  // first batch (H = 512: the only one)
    const int kb = kbeg + 16 * u + 4 * kq;
    const int kc = kb < kend ? kb : (kbeg < H ? kbeg : 0);
    h2v[u] = *reinterpret_cast<const float4*>(g.h2 + (size_t)mc * H + kc);
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[u][j] = *reinterpret_cast<const float2*>(g.W2 + (size_t)(kc + j) * H + i0 + 2 * r);
  }
  // UNCONDITIONAL stores (a thread beyond the row rewrites column 0 with the same bits): behind an `if` LLVM sinks the fetches into
  // the store's block, i.e. behind the waits again
#pragma unroll
  for (int j = 0; j < NO; ++j) { s_wh[j * H + wk0] = whs[j][0]; s_wh[j * H + wk1] = whs[j][1]; }
  for (int k = threadIdx.x + 256 * KW; k < H; k += 256) {  // H > 512
#pragma unroll
    for (int j = 0; j < NO; ++j) s_wh[j * H + k] = j < g.n_out ? g.wh[j][k] : 0.f;
  }
  __syncthreads();
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  auto consume = [&](int k0, const float4 (&h2b)[U], const float2 (&bwb)[U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = k0 + 16 * u + 4 * kq;
      const int kc = kb < kend ? kb : kbeg;
      const bool ok = m_ok && kb < kend;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < NO; ++jj) {  // dh2[m][k] = sum_o g[m][o] Wh[o][k] (output order, fmaf chain)
        const float4 w = *reinterpret_cast<const float4*>(s_wh + jj * H + kc);
        a[0] = fmaf(gj[jj], w.x, a[0]); a[1] = fmaf(gj[jj], w.y, a[1]); a[2] = fmaf(gj[jj], w.z, a[2]); a[3] = fmaf(gj[jj], w.w, a[3]);
      }
      const float hq[4] = {h2b[u].x, h2b[u].y, h2b[u].z, h2b[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float aj = (ok && hq[j] > 0.f) ? a[j] : 0.f;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, bwb[u][j].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, bwb[u][j].y, acc[1], 0, 0, 0);
      }
    }
  };
  if (kend > kbeg) consume(kbeg, h2v, bw);
  for (int k0 = kbeg + 16 * U; k0 < kend; k0 += 16 * U) {  // H > 512: further batches, fetched at the loop top (their own registers:
    float4 h2n[U];                                          // carrying the first batch's through the loop cost 90 VGPRs)
    float2 bwn[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = k0 + 16 * u + 4 * kq;
      const int kc = kb < kend ? kb : kbeg;
      h2n[u] = *reinterpret_cast<const float4*>(g.h2 + (size_t)mc * H + kc);
#pragma unroll
      for (int j = 0; j < 4; ++j) bwn[u][j] = *reinterpret_cast<const float2*>(g.W2 + (size_t)(kc + j) * H + i0 + 2 * r);
    }
    consume(k0, h2n, bwn);
  }
  // observation values of wave 0's epilogue rows (the first PX features), issued HERE: the batch registers are free again (earlier
  // they lift the kernel over 168 VGPRs = three workgroups per CU = the whole grid resident at once), and the fetch overlaps the
  // split-K exchange instead of following it
  constexpr int PX = 8;
  float xv[PX][4];
  if (wid == 0) {
#pragma unroll
    for (int s = 0; s < PX; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[s][i] = s < g.S ? g.x[xrow[i] * g.S + s] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) s_acc[wid][t][lane][i] = acc[t][i];
  __syncthreads();
  if (wid != 0) return;
  float v[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mm = m0 + kq * 4 + i;
    const float2 hh = hmask[i];
    const float s0 = ((s_acc[0][0][lane][i] + s_acc[1][0][lane][i]) + s_acc[2][0][lane][i]) + s_acc[3][0][lane][i];
    const float s1 = ((s_acc[0][1][lane][i] + s_acc[1][1][lane][i]) + s_acc[2][1][lane][i]) + s_acc[3][1][lane][i];
    v[0][i] = (mm < g.B && hh.x > 0.f) ? s0 : 0.f;  // relu'(h1)
    v[1][i] = (mm < g.B && hh.y > 0.f) ? s1 : 0.f;
  }
  float* slab = g.part_w1 + (size_t)tm * ((size_t)H * g.S + H);
  auto dw1_feature = [&](int s, float x0, float x1, float x2, float x3) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float p = 0.f;
      p = fmaf(v[t][0], x0, p); p = fmaf(v[t][1], x1, p); p = fmaf(v[t][2], x2, p); p = fmaf(v[t][3], x3, p);
      p += __shfl_xor(p, 16, 64);
      p += __shfl_xor(p, 32, 64);
      if (kq == 0) slab[(size_t)(i0 + 2 * r + t) * g.S + s] = p;
    }
  };
#pragma unroll
  for (int s = 0; s < PX; ++s)
    if (s < g.S) dw1_feature(s, xv[s][0], xv[s][1], xv[s][2], xv[s][3]);
  for (int s = PX; s < g.S; ++s)  // wider observations (Hopper: S = 11)
    dw1_feature(s, g.x[xrow[0] * g.S + s], g.x[xrow[1] * g.S + s], g.x[xrow[2] * g.S + s], g.x[xrow[3] * g.S + s]);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float p = ((v[t][0] + v[t][1]) + v[t][2]) + v[t][3];
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    if (kq == 0) slab[(size_t)H * g.S + i0 + 2 * r + t] = p;
  }
}

// ---- role: dW2[o][i] = sum_b dh2[b][o] h1[b][i] (32 x 32 tile, both operands 2-column interleaved), db2 = row sums.
// DWH (the workgroups of column tile 0, one per 32 h2 columns): also the head weight gradients dWh[j][k] = sum_b g[b][j] h2[b][k] of
// those 32 columns (rows j < n_out of a 16 x 32 tile) and, in the first of them, dbh = column sums of g.  Round 4 ran them as a third
// role of H/32 extra workgroups: 528 workgroups need three per CU resident at once (<= 168 VGPRs: no room to prefetch anything), and
// its loop compiled into one fetch at a time (tools/isa_chain.py: 13 serial round trips, the longest chain of the launch).  The h2
// columns and the gradient rows it needs are exactly what this role already holds in registers: two more MFMAs per step in 16 of the
// 256 workgroups, same row split over the waves, same accumulation order -> the same bits as the separate role.
template <int NO, bool DWH>
__device__ __forceinline__ void pmb_role_dw2(const PmbBwd& g, int blk, float (*s_acc)[4][64][4], float (*s_rs)[2][64], float (*s_hacc)[2][64][4], float (*s_hrs)[64]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
  const int H = g.H, tiles_n = H / 32;
  const int tm = blk / tiles_n, tn = blk - tm * tiles_n;
  const int o0 = tm * 32, i0 = tn * 32;
  float whr[NO][2];
#pragma unroll
  for (int jj = 0; jj < NO; ++jj) {
    whr[jj][0] = jj < g.n_out ? g.wh[jj][o0 + 2 * r] : 0.f;
    whr[jj][1] = jj < g.n_out ? g.wh[jj][o0 + 2 * r + 1] : 0.f;
  }
  const int kper = ((g.B + 15) / 16) * 4, kbeg = wid * kper;  // rows per wave, multiple of 4
  const int kend = kbeg + kper < g.B ? kbeg + kper : g.B;
  f32x4 acc[2][2], hacc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    hacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float rs[2] = {0.f, 0.f}, hrs = 0.f;
  constexpr int U = NO <= 4 ? 16 : 8;  // rows x 4 in flight per wave: 256-row minibatches are ONE batch of fetches (NO <= 4)
  for (int k0 = kbeg; k0 < kend; k0 += 4 * U) {
    float4 g0[U], g1[U];
    float2 h2v[U], h1v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = k0 + 4 * u + kq;
      const int bc = b < kend ? b : g.B - 1;
      g0[u] = *reinterpret_cast<const float4*>(g.g_all + (size_t)bc * 8);
      if (NO > 4) g1[u] = *reinterpret_cast<const float4*>(g.g_all + (size_t)bc * 8 + 4);
      h2v[u] = *reinterpret_cast<const float2*>(g.h2 + (size_t)bc * H + o0 + 2 * r);
      h1v[u] = *reinterpret_cast<const float2*>(g.h1 + (size_t)bc * H + i0 + 2 * r);
    }
    // the whole batch in flight before the first MFMA: without this fence the machine scheduler interleaves fetch and use in the
    // DWH variant (register-pressure heuristic), one round trip per step -- what made the separate head-gradient role the longest chain
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = k0 + 4 * u + kq < kend;
      float gq[8] = {g0[u].x, g0[u].y, g0[u].z, g0[u].w, 0.f, 0.f, 0.f, 0.f};
      if (NO > 4) { gq[4] = g1[u].x; gq[5] = g1[u].y; gq[6] = g1[u].z; gq[7] = g1[u].w; }
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int jj = 0; jj < NO; ++jj) {
        a0 = fmaf(gq[jj], whr[jj][0], a0);
        a1 = fmaf(gq[jj], whr[jj][1], a1);
      }
      if (DWH) {  // A[m = r][k = kq] = g[row of this lane][r]: picked BEFORE the relu mask below touches nothing of it
        float gr = gq[0];
#pragma unroll
        for (int jj = 1; jj < NO; ++jj) gr = (r & 7) == jj ? gq[jj] : gr;
        const float ah = (ok && r < g.n_out) ? gr : 0.f;
        hrs += ah;
        hacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, h2v[u].x, hacc[0], 0, 0, 0);
        hacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, h2v[u].y, hacc[1], 0, 0, 0);
      }
      a0 = (ok && h2v[u].x > 0.f) ? a0 : 0.f;  // relu'(h2)
      a1 = (ok && h2v[u].y > 0.f) ? a1 : 0.f;
      rs[0] += a0;
      rs[1] += a1;
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, h1v[u].x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, h1v[u].y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, h1v[u].x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, h1v[u].y, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) s_acc[wid][t * 2 + u][lane][i] = acc[t][u][i];
    s_rs[wid][t][lane] = rs[t];
    if (DWH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s_hacc[wid][t][lane][i] = hacc[t][i];
    }
  }
  if (DWH) s_hrs[wid][lane] = hrs;
  __syncthreads();
  if (wid != 0) return;
  float ssq = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = o0 + 2 * (kq * 4 + i) + t;  // row tile t owns the rows o0 + 2 * row + t
      float2 out;
      out.x = ((s_acc[0][t * 2][lane][i] + s_acc[1][t * 2][lane][i]) + s_acc[2][t * 2][lane][i]) + s_acc[3][t * 2][lane][i];
      out.y = ((s_acc[0][t * 2 + 1][lane][i] + s_acc[1][t * 2 + 1][lane][i]) + s_acc[2][t * 2 + 1][lane][i]) + s_acc[3][t * 2 + 1][lane][i];
      *reinterpret_cast<float2*>(g.dW2 + (size_t)o * H + i0 + 2 * r) = out;
      ssq = fmaf(out.x, out.x, ssq);
      ssq = fmaf(out.y, out.y, ssq);
    }
    if (tn == 0) {
      float s = ((s_rs[0][t][lane] + s_rs[1][t][lane]) + s_rs[2][t][lane]) + s_rs[3][t][lane];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (kq == 0) {
        g.db2[o0 + 2 * r + t] = s;
        ssq = fmaf(s, s, ssq);
      }
    }
  }
  if (g.ssq_part) {
    ssq = pmb_wave_sum(ssq);
    if (lane == 0) g.ssq_part[blk] = ssq;
  }
  if (!DWH) return;
  // ---- head weight gradients of the columns o0 .. o0 + 31 (the former third role's epilogue, its own sum-of-squares slot)
  float hsq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = kq * 4 + i;
    if (j < g.n_out) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float w = ((s_hacc[0][t][lane][i] + s_hacc[1][t][lane][i]) + s_hacc[2][t][lane][i]) + s_hacc[3][t][lane][i];
        g.dwh[j][o0 + 2 * r + t] = w;
        hsq = fmaf(w, w, hsq);
      }
    }
  }
  if (tm == 0) {
    float s = ((s_hrs[0][lane] + s_hrs[1][lane]) + s_hrs[2][lane]) + s_hrs[3][lane];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (kq == 0 && r < g.n_out) {
      *g.dbh[r] = s;
      hsq = fmaf(s, s, hsq);
    }
  }
  if (g.ssq_part) {
    hsq = pmb_wave_sum(hsq);
    if (lane == 0) g.ssq_part[g.n_dw2 + tm] = hsq;
  }
}

template <int NO, int U1>
__global__ void __launch_bounds__(256) jh_pmb_bwd_kernel(PmbBwd g) {
  __shared__ float s_acc[4][4][64][4];
  __shared__ float s_rs[4][2][64];
  __shared__ float s_hacc[4][2][64][4];
  __shared__ float s_hrs[4][64];
  extern __shared__ __attribute__((aligned(16))) float s_wh[];  // [NO][H] (dh1 role)
  const int b = blockIdx.x;
  if (b < g.n_dh1) {  // the longest chains first
    if (g.x_rows) pmb_role_dh1<NO, U1, true>(g, b, s_acc, s_wh);
    else pmb_role_dh1<NO, U1, false>(g, b, s_acc, s_wh);
  } else if ((b - g.n_dh1) % (g.H / 32) == 0) {
    pmb_role_dw2<NO, true>(g, b - g.n_dh1, s_acc, s_rs, s_hacc, s_hrs);
  } else {
    pmb_role_dw2<NO, false>(g, b - g.n_dh1, s_acc, s_rs, s_hacc, s_hrs);
  }
}

// ============================================================================ 4. dW1 / db1 combine + global norm
// hyper (device): [0] lr [1] beta1 [2] beta2 [3] eps [4] step [5] 1-beta1^t [6] sqrt(1-beta2^t)
__global__ void __launch_bounds__(256) jh_pmb_norm_kernel(int64_t n, float* __restrict__ grads, const float* __restrict__ part,
                                                          int tiles_m, int64_t n_head, float* __restrict__ partial,
                                                          float* __restrict__ hyper, int do_norm) {
  __shared__ float s_red[16];
  float acc = 0.f;
  const int64_t n4 = n >> 2, h4 = n_head >> 2;  // n_head % 4 == 0 (H % 32 == 0); the buckets are 16-byte aligned
  const int64_t lim4 = do_norm ? n4 : h4;
  float4* g4 = reinterpret_cast<float4*>(grads);
  const float4* p4 = reinterpret_cast<const float4*>(part);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < lim4; i += (int64_t)gridDim.x * 256) {
    float4 v;
    if (i < h4) {
      v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t0 = 0; t0 < tiles_m; t0 += 8) {  // row-tile order: deterministic; 8 loads in flight
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = p4[(size_t)(t0 + u < tiles_m ? t0 + u : tiles_m - 1) * h4 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (t0 + u < tiles_m) { v.x += q[u].x; v.y += q[u].y; v.z += q[u].z; v.w += q[u].w; }
      }
      g4[i] = v;
    } else {
      v = g4[i];
    }
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  if (!do_norm) return;
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - (n4 << 2))) {  // the <= 3 trailing elements
    const float v = grads[(n4 << 2) + threadIdx.x];
    acc = fmaf(v, v, acc);
  }
  acc = jh_block_reduce(acc, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) jh_adam_advance(hyper);  // nobody reads hyper in this kernel
}

// ============================================================================ host side
bool jh_pmb_eligible(const jh_pponet* n, int B) { return n->H % 32 == 0 && B > 0 && B <= n->max_rows && n->n_out <= 8; }  // (packed [B][8] head gradients, 8 partial-head slots per tile)

template <int SV, bool GEN>
static int pmb_fwd_launch(const PmbFwd& g, hipStream_t st) {
  const int tiles = ((g.M + 15) / 16) * (g.H / 16);
  const int kper = ((g.H + 63) / 64) * 16;
  // flops: layer 1 (generated) + the H x H contraction + the heads
  const double fl = 2.0 * g.M * (double)g.H * ((GEN ? g.S : 0) + g.H + g.n_out);
  const char* nm = g.M > 1024 ? "jh_pmb_fwd_nograd" : "jh_pmb_fwd";  // the no-grad pass over [state; next_state] vs a minibatch
  const size_t lds = GEN ? sizeof(float) * 4 * (size_t)kper * ((SV == 3 ? ((g.S + 3) & ~3) : g.S) + 1) : 0;
  if (kper > 64) JH_LAUNCH_IDEM(nm, fl, (jh_pmb_fwd_kernel<SV, GEN, 8>), dim3(tiles), dim3(256), lds, st, g);
  else if (kper > 32) JH_LAUNCH_IDEM(nm, fl, (jh_pmb_fwd_kernel<SV, GEN, 4>), dim3(tiles), dim3(256), lds, st, g);
  else JH_LAUNCH_IDEM(nm, fl, (jh_pmb_fwd_kernel<SV, GEN, 2>), dim3(tiles), dim3(256), lds, st, g);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

int jh_pmb_forward(jh_pponet* n, int M, const float* d_x, const int64_t* d_idx, const PmbHeads& hd, const float* h1_in,
                   bool store_act, hipStream_t st) {
  PmbFwd g{};
  g.M = M; g.H = n->H; g.S = n->S; g.x = d_x; g.x_rows = d_idx;
  g.W1 = n->params + n->o_w1; g.b1 = n->params + n->o_b1; g.W2 = n->params + n->o_w2; g.b2 = n->params + n->o_b2;
  g.h1_in = h1_in; g.h1_out = store_act ? n->h1 : nullptr; g.h2_out = store_act ? n->h2 : nullptr;
  for (int o = 0; o < hd.n_out; ++o) { g.wh[o] = hd.w[o]; g.hb[o] = hd.b[o]; }
  g.n_out = hd.n_out; g.part = n->fwd_part; g.part_rows = n->max_rows; g.part_ld = hd.n_out <= 4 ? 4 : 8;
  if (h1_in) return pmb_fwd_launch<0, false>(g, st);
  const bool al = (((uintptr_t)d_x) & 15) == 0 && (((uintptr_t)g.W1) & 15) == 0;
  if (n->S == 4 && al) return pmb_fwd_launch<1, true>(g, st);
  if (n->S == 8 && al) return pmb_fwd_launch<2, true>(g, st);
  if (n->S > 8) return pmb_fwd_launch<3, true>(g, st);  // 9 .. 16 observations
  return pmb_fwd_launch<0, true>(g, st);
}

int jh_pmb_heads_finish(jh_pponet* n, int M, float* d_head0, float* d_head1, float* d_value, hipStream_t st) {
  JH_LAUNCH(jh_pmb_heads_finish_kernel, dim3((M + 255) / 256), dim3(256), 0, st, M, n->H / 16, n->max_rows, (n->cont ? 2 * n->A + 1 : n->A + 1) <= 4 ? 4 : 8, n->fwd_part, n->A,
            n->cont, d_head0, d_head1, d_value);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

int jh_pmb_backward(jh_pponet* n, int B, const float* d_x, const int64_t* d_idx, const PmbHeads& hd, bool emit_ssq, hipStream_t st) {
  PmbBwd g{};
  g.B = B; g.H = n->H; g.S = n->S; g.n_out = hd.n_out; g.x = d_x; g.x_rows = d_idx;
  g.h1 = n->h1; g.h2 = n->h2; g.g_all = n->g_all; g.W2 = n->params + n->o_w2;
  g.dW2 = n->grads + n->o_w2; g.db2 = n->grads + n->o_b2;
  for (int o = 0; o < hd.n_out; ++o) { g.wh[o] = hd.w[o]; g.dwh[o] = hd.dw[o]; g.dbh[o] = hd.db[o]; }
  g.part_w1 = n->part_w1;
  g.ssq_part = emit_ssq ? n->ssq_part : nullptr;
  const int t32 = n->H / 32;
  g.n_dh1 = ((B + 15) / 16) * t32;
  g.n_dw2 = t32 * t32;
  const int grid = g.n_dh1 + g.n_dw2;  // 512 workgroups at B = 256, H = 512: two per CU, all resident (<= 256 VGPRs)
  static const int u1 = getenv("JH_PMB_U1") ? atoi(getenv("JH_PMB_U1")) : 8;  // k-chunks of dh1 loaded ahead of the MFMAs
  const bool deep = u1 >= 8 && n->H >= 512;
  // flops: dW2 + dh1 (B x H x H each), head weight gradients, dW1 partials, dh2 generated twice
  const double fl = 2.0 * B * (double)n->H * (2.0 * n->H + hd.n_out + n->S + 2.0 * hd.n_out);
  if (hd.n_out <= 4) {
    if (deep) JH_LAUNCH_IDEM("jh_pmb_bwd", fl, (jh_pmb_bwd_kernel<4, 8>), dim3(grid), dim3(256), sizeof(float) * 4 * (size_t)n->H, st, g);
    else JH_LAUNCH_IDEM("jh_pmb_bwd", fl, (jh_pmb_bwd_kernel<4, 4>), dim3(grid), dim3(256), sizeof(float) * 4 * (size_t)n->H, st, g);
  } else {
    if (deep) JH_LAUNCH_IDEM("jh_pmb_bwd", fl, (jh_pmb_bwd_kernel<8, 8>), dim3(grid), dim3(256), sizeof(float) * 8 * (size_t)n->H, st, g);
    else JH_LAUNCH_IDEM("jh_pmb_bwd", fl, (jh_pmb_bwd_kernel<8, 4>), dim3(grid), dim3(256), sizeof(float) * 8 * (size_t)n->H, st, g);
  }
  JH_LAUNCH_CHECK();
  return JH_OK;
}

int jh_pmb_finalize(jh_pponet* n, int B, bool with_norm, hipStream_t st) {
  const int64_t n_head = (int64_t)n->H * n->S + n->H;
  const int tiles_m = (B + 15) / 16;
  const int blocks = with_norm ? 256 : (int)((n_head / 4 + 255) / 256);
  JH_LAUNCH(jh_pmb_norm_kernel, dim3(blocks), dim3(256), 0, st, n->n_params, n->grads, n->part_w1, tiles_m, n_head, n->norm_partial,
            n->hyper, with_norm ? 1 : 0);
  JH_LAUNCH_CHECK();
  return JH_OK;
}
