// The value networks of the DQN / Rainbow / Ape-X family on the device: Nature-CNN or MLP head, then
//   kind 0 / 3  rainbow  -> Linear -> noisy dueling categorical heads (factorised / independent noise)
//   kind 1      dueling  -> l1_a | l1_v -> l2_a, l2_v -> dueling combine
//   kind 2      q        -> Linear -> q
//   reference: core/network/rainbow.py:8-94, dueling.py:8-35, q_network.py:8-20, head.py:6-61 (MLP / CNN),
//              utils.py:55-107 (noisy linear); the three forwards + backward + optimizer step of one learn()
//              (core/agent/rainbow.py:154-253, dqn.py:128-147, ape_x.py:96-131).
//
// Every contraction (convolutions as implicit GEMMs, linear layers, their data- and weight-gradients) runs
// on ONE grouped LDS-tiled fp32 MFMA kernel (jh_tgemm.hip).  What changes between layers is only how an operand
// tile is fetched from HBM (dense, transposed, im2col of an NHWC activation, im2col of the NCHW uint8
// frames straight out of the replay store) -- the column matrix of a convolution is never materialised
// in the forward pass or the weight-gradient pass.
//
// Private layouts (import/export in jorldy_amd/ops.py permutes to the reference's state_dict):
//   activations  NHWC  [rows = (b, oy, ox)][channels]        (a GEMM's C is the next conv's input as is)
//   conv weights [out][(ky, kx, c)]  (layer 1: [out][(c, ky, kx)] = the reference layout, input is NCHW)
//   linear / noisy weights [out][in] (the reference keeps noisy weights as [in][out])
//   a1 | v1 noisy layers stacked into one [2H][H] matrix (one GEMM feeds both streams)
#include "jh_tgemm.h"
#include "jh_fused.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---------------------------------------------------------------------------------- noisy layers
__device__ __forceinline__ float noise_f(float e) { return copysignf(sqrtf(fabsf(e)), e); }  // utils.py:66-67 (sign(0) = 0 either way)

// Layout of one noise set (standard normal draws, reference draw order utils.py:58-60 per layer a1, v1, a2, v2):
//   [ei_a1 H][ej_a1 H][ei_v1 H][ej_v1 H][ei_a2 H][ej_a2 NA][ei_v2 H][ej_v2 K]
struct NoisyDims {
  int H, NA, K;
  int64_t o_av1, o_bav1, o_a2, o_ba2, o_v2, o_bv2, set_stride;  // inside one effective-weight set
  int64_t p_mu_av1, p_sig_av1, p_mub_av1, p_sigb_av1, p_mu_a2, p_sig_a2, p_mub_a2, p_sigb_a2, p_mu_v2, p_sig_v2, p_mub_v2, p_sigb_v2;
  int64_t noise_len;
  int independent;  // utils.py:73-76: a full eps_w [in][out] + eps_b [out] per layer instead of the outer product
};

// factor pair (f_j of the output unit, f_i of the input unit) for flat index i of the weight part
__device__ __forceinline__ void noisy_locate(const NoisyDims& d, int64_t i, const float* e, int64_t* p_mu, int64_t* p_sig, int64_t* w_off, float* eps) {
  const int H = d.H;
  const int64_t n_av1 = (int64_t)2 * H * H, n_a2 = (int64_t)d.NA * H, n_v2 = (int64_t)d.K * H;
  int64_t n, k;
  const float *ei, *ej;
  // independent noise, one set: [eps_w a1 H x H (in, out)][eps_b a1 H][eps_w v1][eps_b v1][eps_w a2 H x NA][eps_b a2 NA][eps_w v2 H x K][eps_b v2 K]
  const int64_t hh = (int64_t)H * H, i_a1 = 0, i_v1 = hh + H, i_a2 = 2 * (hh + H), i_v2 = i_a2 + (int64_t)H * d.NA + d.NA;
  int64_t ind = 0;  // index of this weight's own draw (independent noise)
  if (i < n_av1) {
    n = i / H; k = i - n * H;
    *p_mu = d.p_mu_av1 + i; *p_sig = d.p_sig_av1 + i; *w_off = d.o_av1 + i;
    if (n < H) { ei = e; ej = e + H; ind = i_a1 + k * H + n; } else { ei = e + 2 * H; ej = e + 3 * H; n -= H; ind = i_v1 + k * H + n; }
  } else if (i < n_av1 + n_a2) {
    i -= n_av1;
    n = i / H; k = i - n * H;
    *p_mu = d.p_mu_a2 + i; *p_sig = d.p_sig_a2 + i; *w_off = d.o_a2 + i;
    ei = e + 4 * H; ej = e + 5 * H;
    ind = i_a2 + k * d.NA + n;
  } else if (i < n_av1 + n_a2 + n_v2) {
    i -= n_av1 + n_a2;
    n = i / H; k = i - n * H;
    *p_mu = d.p_mu_v2 + i; *p_sig = d.p_sig_v2 + i; *w_off = d.o_v2 + i;
    ei = e + 5 * H + d.NA; ej = e + 6 * H + d.NA;
    ind = i_v2 + k * d.K + n;
  } else {  // biases: eps_b = f_j (utils.py:69) / its own draw (utils.py:76)
    i -= n_av1 + n_a2 + n_v2;
    if (i < 2 * H) {
      *p_mu = d.p_mub_av1 + i; *p_sig = d.p_sigb_av1 + i; *w_off = d.o_bav1 + i;
      if (d.independent) *eps = e ? (i < H ? e[i_a1 + hh + i] : e[i_v1 + hh + (i - H)]) : 0.f;
      else *eps = e ? noise_f(i < H ? e[H + i] : e[3 * H + (i - H)]) : 0.f;
    } else if (i < 2 * H + d.NA) {
      i -= 2 * H;
      *p_mu = d.p_mub_a2 + i; *p_sig = d.p_sigb_a2 + i; *w_off = d.o_ba2 + i;
      if (d.independent) *eps = e ? e[i_a2 + (int64_t)H * d.NA + i] : 0.f;
      else *eps = e ? noise_f(e[5 * H + i]) : 0.f;
    } else {
      i -= 2 * H + d.NA;
      *p_mu = d.p_mub_v2 + i; *p_sig = d.p_sigb_v2 + i; *w_off = d.o_bv2 + i;
      if (d.independent) *eps = e ? e[i_v2 + (int64_t)H * d.K + i] : 0.f;
      else *eps = e ? noise_f(e[6 * H + d.NA + i]) : 0.f;
    }
    return;
  }
  if (d.independent) *eps = e ? e[ind] : 0.f;
  else *eps = e ? noise_f(ei[k]) * noise_f(ej[n]) : 0.f;  // torch.outer(f_i, f_j)
}

struct NoiseSets {
  const float* params[3];
  const float* noise[3];  // null: evaluation mode, W = mu (rainbow.py network: is_train False)
  float* weff[3];
  int n_sets;
};

// W_eff = mu + sig * eps for up to three (parameter set, noise draw) pairs in one launch
__global__ void __launch_bounds__(256) jh_rb_noise_kernel(NoisyDims d, NoiseSets s, int64_t n_total) {
  const int set = blockIdx.y;
  const float* P = s.params[set];
  const float* e = s.noise[set];
  float* W = s.weff[set];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_total; i += (int64_t)gridDim.x * 256) {
    int64_t pm, ps, wo;
    float eps;
    noisy_locate(d, i, e, &pm, &ps, &wo, &eps);
    W[wo] = e ? P[pm] + P[ps] * eps : P[pm];
  }
}

// d(sig) = d(W_eff) * eps; d(mu) = d(W_eff) was written in place by the weight-gradient GEMMs
__global__ void __launch_bounds__(256) jh_rb_noisy_grad_kernel(NoisyDims d, const float* __restrict__ e, float* __restrict__ G, int64_t n_total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_total; i += (int64_t)gridDim.x * 256) {
    int64_t pm, ps, wo;
    float eps;
    noisy_locate(d, i, e, &pm, &ps, &wo, &eps);
    G[ps] = G[pm] * eps;
  }
}

// ---------------------------------------------------------------------------------- dueling combine
struct DuelSets {
  const float* xa[3];
  const float* xv[3];
  float* out[3];
};
// logits[b][a][k] = xa[b][a][k] - mean_a xa[b][.][k] + xv[b][k]   (network/rainbow.py:88-93)
__global__ void __launch_bounds__(256) jh_rb_duel_fwd_kernel(DuelSets s, int B, int A, int K, int ld_a, int ld_v) {
  const int set = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  const float* xa = s.xa[set] + (size_t)b * ld_a;
  float sum = 0.f;
  for (int a = 0; a < A; ++a) sum += xa[a * K + k];
  const float mean = sum / (float)A;
  const float v = s.xv[set][(size_t)b * ld_v + k];
  float* o = s.out[set] + (size_t)b * A * K;
  for (int a = 0; a < A; ++a) o[a * K + k] = (xa[a * K + k] - mean) + v;
}
__global__ void __launch_bounds__(256) jh_rb_duel_bwd_kernel(const float* __restrict__ g, int B, int A, int K, float* __restrict__ dxa, int ld_a,
                                                             float* __restrict__ dxv, int ld_v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  const float* gb = g + (size_t)b * A * K;
  float sum = 0.f;
  for (int a = 0; a < A; ++a) sum += gb[a * K + k];
  dxv[(size_t)b * ld_v + k] = sum;
  const float mean = sum / (float)A;
  for (int a = 0; a < A; ++a) dxa[(size_t)b * ld_a + a * K + k] = gb[a * K + k] - mean;
}

// ---------------------------------------------------------------------------------- col2im (+ relu')
// d(act)[(b, y, x)][c] = relu'(act) * sum over the taps (ky, kx) that read this pixel of d(col)[(b, oy, ox)][(ky, kx, c)]
// Round 5: the taps that can reach a pixel are ky = y % S + S t, kx = x % S + S u (all others fail `yy % S`): (KH / S) (KW / S) of
// them -- 4 for conv2's 4 x 4 stride 2, 9 for conv3's 3 x 3 stride 1 --, fetched as ONE batch from clamped addresses and added in
// ascending (ky, kx) order where valid.  The loop over all KH KW taps with `continue` compiled to one fetch per wait: up to nine
// dependent round trips per pixel (tools/isa_chain.py), 7-10 us per launch at B = 32.
template <int KH, int KW, int S>
__global__ void __launch_bounds__(256) jh_rb_col2im_kernel(const float* __restrict__ dcol, const float* __restrict__ act, float* __restrict__ dact,
                                                           int n_pix, int C, int H, int W, int OH, int OW) {
  constexpr int TY = (KH + S - 1) / S, TX = (KW + S - 1) / S;
  const int c4 = C / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)n_pix * c4) return;
  const int pix = (int)(i / c4), c = 4 * (int)(i - (int64_t)pix * c4);
  const int x = pix % W, t = pix / W, y = t % H, b = t / H;
  const int Kp = KH * KW * C;
  const float4 a = *reinterpret_cast<const float4*>(act + (size_t)pix * C + c);
  float4 v[TY][TX];
  bool ok[TY][TX];
#pragma unroll
  for (int ty = 0; ty < TY; ++ty) {
    const int ky = y % S + S * ty, yy = y - ky, oy = yy / S;  // yy % S == 0 by construction
    const bool oky = ky < KH && yy >= 0 && oy < OH;
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) {
      const int kx = x % S + S * tx, xx = x - kx, ox = xx / S;
      ok[ty][tx] = oky && kx < KW && xx >= 0 && ox < OW;
      const int oyc = ok[ty][tx] ? oy : 0, oxc = ok[ty][tx] ? ox : 0, kc = ok[ty][tx] ? ky * KW + kx : 0;
      v[ty][tx] = *reinterpret_cast<const float4*>(dcol + ((size_t)(b * OH + oyc) * OW + oxc) * Kp + kc * C + c);
    }
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ty = 0; ty < TY; ++ty)
#pragma unroll
    for (int tx = 0; tx < TX; ++tx)
      if (ok[ty][tx]) { acc.x += v[ty][tx].x; acc.y += v[ty][tx].y; acc.z += v[ty][tx].z; acc.w += v[ty][tx].w; }
  acc.x = a.x > 0.f ? acc.x : 0.f; acc.y = a.y > 0.f ? acc.y : 0.f; acc.z = a.z > 0.f ? acc.z : 0.f; acc.w = a.w > 0.f ? acc.w : 0.f;
  *reinterpret_cast<float4*>(dact + (size_t)pix * C + c) = acc;
}

// ---------------------------------------------------------------------------------- conv1 forward on uint8 frames
// act1[(b, oy, ox)][oc] = relu(b1[oc] + sum over taps of x[b][c][4 oy + ky][4 ox + kx] / 255 * W1[oc][c][ky][kx]) for the head's first layer.
// As a GEMM: M = pixels, N = 32, K = 64 Cin -- half a tile wide and eight chunks deep on the tile engine, whose uint8 operand goes
// global -> VGPR -> convert -> LDS one dword per lane (170 us for 3 x 512 frames, 38 % of the MFMA peak; 24 us at B = 32).  Here a
// workgroup keeps W1 (32 KB, row stride + 4 floats) in LDS and walks 16-pixel tiles of ITS frame, one tile per wave and turn:
//   A[m = pixel r][k]   lane group kq of step s loads the dword of taps (c, ky, 4 h .. 4 h + 3), (c, ky, h) = bits of 4 s + kq, straight
//                       from the frame: byte j is the lane's k element of MFMA j of that step (the float4 trick of the tile engine)
//   B[k][n = oc r]      the matching float4 of W1 row oc: one ds_read_b128 per step and 16-channel tile
// all 4 Cin dwords of a tile are in flight before its first MFMA; no barriers after the weight staging.
struct C1Job {
  const uint8_t* x;
  const float *W, *bias;
  float* out;
  int wg_begin;
};
struct C1Fwd {
  C1Job j[2];
  int nj, H, W, OW, P, tiles_per_img, tiles_per_wg, wgs_per_img;
  float inv_ow;
};
template <int CIN>
__global__ void __launch_bounds__(256) jh_rb_conv1_fwd_kernel(C1Fwd a) {
  constexpr int K = CIN * 64, LDW = K + 4, NS = CIN * 4;
  __shared__ __attribute__((aligned(16))) float sW[32 * LDW];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 15, kq = lane >> 4;
  const C1Job& J = a.j[(a.nj > 1 && (int)blockIdx.x >= a.j[1].wg_begin) ? 1 : 0];
  const int local = blockIdx.x - J.wg_begin;
  const int img = local / a.wgs_per_img, part = local - img * a.wgs_per_img;
  // Fetch order (round 5, tools/isa_chain.py): the weight staging was 8 passes compiled as 5 + 1 + 1 + 1 fetches per wait, the bias
  // followed the barrier, a tile's 16 frame dwords came as two batches of 8 and the next tile's only after the last MFMA of this one
  // -- about ten dependent round trips for a launch with 0.2 us of matrix work.  Now: the first tile's frame dwords, the bias and all
  // weight passes are ONE batch in front of the barrier, and tile i + 1 is fetched in front of tile i's MFMAs.
  const int t0 = part * a.tiles_per_wg;
  int t1 = t0 + a.tiles_per_wg;
  if (t1 > a.tiles_per_img) t1 = a.tiles_per_img;
  const uint8_t* frame = J.x + (size_t)img * CIN * a.H * a.W;
  float* out = J.out + (size_t)img * a.P * 32;
  auto fetch_tile = [&](int tile, uint32_t (&xw)[NS]) {  // a tile past the end re-reads the last pixel (never used)
    const int p = tile * 16 + r, pc = p < a.P ? p : a.P - 1;
    const int oy = (int)(((float)pc + 0.5f) * a.inv_ow), ox = pc - oy * a.OW;
    const uint8_t* base = frame + (size_t)(4 * oy) * a.W + 4 * ox;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int q = 4 * s + kq, c = q >> 4, ky = (q >> 1) & 7, h = q & 1;
      xw[s] = *reinterpret_cast<const uint32_t*>(base + (size_t)(c * a.H + ky) * a.W + 4 * h);
    }
  };
  uint32_t xw[NS], xn[NS];
  fetch_tile(t0 + wid, xw);
  const float bias0 = J.bias[r], bias1 = J.bias[16 + r];
  {
    constexpr int NP = 32 * (K / 4) / 256;  // 8 passes for 4 input planes
    static_assert(32 * (K / 4) % 256 == 0, "whole passes");
    float4 wv[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = t + 256 * u, oc = i / (K / 4), k4 = i - oc * (K / 4);
      wv[u] = *reinterpret_cast<const float4*>(J.W + (size_t)oc * K + 4 * k4);
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = t + 256 * u, oc = i / (K / 4), k4 = i - oc * (K / 4);
      *reinterpret_cast<float4*>(&sW[oc * LDW + 4 * k4]) = wv[u];
    }
  }
  __syncthreads();
  auto run_tile = [&](int tile, const uint32_t (&xv)[NS]) {
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float af[4] = {u8_unit(xv[s] & 255u), u8_unit((xv[s] >> 8) & 255u), u8_unit((xv[s] >> 16) & 255u), u8_unit(xv[s] >> 24)};
      const float4 w0 = *reinterpret_cast<const float4*>(&sW[r * LDW + 4 * (4 * s + kq)]);
      const float4 w1 = *reinterpret_cast<const float4*>(&sW[(16 + r) * LDW + 4 * (4 * s + kq)]);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], w0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], w1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], w0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], w1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], w0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], w1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], w0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], w1.w, acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // C fragment: rows (pixels) 4 kq + q, column (channel) r
      const int m = tile * 16 + 4 * kq + q;
      if (m < a.P) {
        const float v0 = acc[0][q] + bias0, v1 = acc[1][q] + bias1;
        out[(size_t)m * 32 + r] = v0 > 0.f ? v0 : 0.f;
        out[(size_t)m * 32 + 16 + r] = v1 > 0.f ? v1 : 0.f;
      }
    }
  };
  // two tiles per turn, registers ping-pong (no copies: a copy of a fetched value is a wait)
  for (int tile = t0 + wid; tile < t1; tile += 8) {
    fetch_tile(tile + 4, xn);
    __builtin_amdgcn_sched_barrier(0);
    run_tile(tile, xw);
    if (tile + 4 >= t1) break;
    fetch_tile(tile + 8, xw);
    __builtin_amdgcn_sched_barrier(0);
    run_tile(tile + 4, xn);
  }
}

// ---------------------------------------------------------------------------------- conv1 weight gradient on uint8 frames
// dW1[oc][c][ky][kx] = sum over (b, oy, ox) of dY[(b, oy, ox)][oc] * x[b][c][4 oy + ky][4 ox + kx] / 255 for the head's first layer
// (8 x 8 kernel, stride 4, 32 output channels, uint8 NCHW frames: core/network/head.py:37-47 CNN).  As a GEMM this is M = 32,
// N = 64 Cin, K = B * OH * OW (12 800 at B = 32, 204 800 at B = 512): four 64 x 64 output tiles for the whole chip, so the tile engine
// splits K 64 ways and its last arrivers sum 64 partial tiles each, alone -- 33 us at B = 32, 140 us at B = 512 (15 % of the MFMA peak).
// Here the K range is the parallel dimension: workgroup g takes a contiguous run of 4-pixel groups, wave w the input channels
// w, w + 4, ..., and a wave keeps the whole 32 x 64 block of ITS channel in 8 accumulator tiles:
//   A[m = oc][k = pixel kq]   two dwords of dY per lane and group (oc = r, r + 16)
//   B[k = pixel kq][n = r]    tile j holds the taps 4 r + j, i.e. (ky = r >> 1, kx = 4 (r & 1) + j): ONE aligned dword of the frame per
//                             lane and group is the B operand of all four tiles, unpacked with v_cvt_f32_ubyte0..3
// (the permutation of the taps is undone by the store: a lane ends up with 4 consecutive kx of one (oc, c, ky) -> one 16-byte store).
// UN groups' loads (3 dwords each) are in flight before the first MFMA of a round, and a workgroup is SIXTEEN waves -- four K slices
// of its run x four channels -- so that every SIMD has four independent load -> MFMA chains to interleave (with one wave per SIMD the
// kernel sat at 4 x its MFMA time: 86 us at B = 512); slices 1..3 hand their accumulators to slice 0 through LDS, added in slice order.
// Every workgroup writes its partial [32][64 Cin] (+ 32 bias sums) and jh_rb_parts_sum_kernel adds the partials in workgroup order:
// deterministic.
template <int UN>
__global__ void __launch_bounds__(1024) jh_rb_conv1_wgrad_kernel(const uint8_t* __restrict__ x, const float* __restrict__ dY, float* __restrict__ part, int Cin,
                                                                 int H, int W, int OW, int P, float inv_ow, int groups_per_img, int total_groups,
                                                                 int groups_per_wg) {
  __shared__ float s_acc[3][4][34][64];  // [slice - 1][channel wave][32 accumulator registers + 2 row sums][lane]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, kq = lane >> 4;
  const int cw = wid & 3, slice = wid >> 2;
  const int per_slice = groups_per_wg / 4;  // the host keeps groups_per_wg a multiple of 4 UN
  const int g0 = blockIdx.x * groups_per_wg + slice * per_slice;
  int g1 = g0 + per_slice;
  if (g1 > total_groups) g1 = total_groups;
  const int NW = Cin * 64, n_all = 32 * NW + 32;
  float* mine = part + (size_t)blockIdx.x * n_all;
  const int ky = r >> 1, kx0 = 4 * (r & 1);
  const size_t CHW = (size_t)Cin * H * W;
  for (int c = cw; c < Cin; c += 4) {
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double rs[2] = {0.0, 0.0};  // bias gradient = a 200 000-term sum of dY at Ape-X shapes: double running sums cost two adds next to eight MFMAs
    // The walk over (frame b, 4-pixel group pg) is incremental: one division per wave and channel, then adds and two compares per
    // group (P % 4 == 0 and OW >= 4: the host checks).  Addresses as (b, pixel) -> offsets took 370 VALU instructions per round of 5
    // groups next to 40 MFMAs; the kernel ran at the VALU's pace (62 us at B = 512), not the matrix core's.
    int b = g0 / groups_per_img, pg = g0 - b * groups_per_img;  // uniform
    int oy, ox;                                                 // this lane's pixel (kq-th of the group)
    {
      const int p = pg * 4 + kq;
      oy = (int)(((float)p + 0.5f) * inv_ow);
      ox = p - oy * OW;
    }
    const uint8_t* xl = x + (size_t)c * H * W + (size_t)ky * W + kx0;  // lane-constant part of the frame address
    const float* dyl = dY + (size_t)kq * 32 + r;
    auto fetch = [&](float (&a)[2], uint32_t& xw) {
      const float* dy = dyl + ((size_t)b * P + (size_t)pg * 4) * 32;
      a[0] = dy[0];
      a[1] = dy[16];
      xw = *reinterpret_cast<const uint32_t*>(xl + (size_t)b * CHW + (size_t)(4 * oy) * W + 4 * ox);
      ox += 4;
      if (ox >= OW) { ox -= OW; oy += 1; }
      if (++pg == groups_per_img) { pg = 0; ++b; oy = 0; ox = kq; }
    };
    auto consume = [&](const float (&a)[2], uint32_t xw) {
      const float bf[4] = {u8_unit(xw & 255u), u8_unit((xw >> 8) & 255u), u8_unit((xw >> 16) & 255u), u8_unit(xw >> 24)};
      rs[0] += a[0];
      rs[1] += a[1];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bf[j], acc[i][j], 0, 0, 0);
    };
    int gi = g0;
#pragma unroll 1
    for (; gi + UN <= g1; gi += UN) {
      float a[UN][2];
      uint32_t xw[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) fetch(a[u], xw[u]);
#pragma unroll
      for (int u = 0; u < UN; ++u) consume(a[u], xw[u]);
    }
#pragma unroll 1
    for (; gi < g1; ++gi) {  // ragged end of the last workgroup's run
      float a[2];
      uint32_t xw;
      fetch(a, xw);
      consume(a, xw);
    }
    if (slice > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) s_acc[slice - 1][cw][(i * 4 + j) * 4 + q][lane] = acc[i][j][q];
        s_acc[slice - 1][cw][32 + i][lane] = (float)rs[i];
      }
    }
    __syncthreads();
    if (slice == 0) {
#pragma unroll
      for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] += s_acc[sl][cw][(i * 4 + j) * 4 + q][lane];
          rs[i] += s_acc[sl][cw][32 + i][lane];
        }
      // C fragment: a lane holds rows 4 kq + q, column r of every tile: tile j is tap 4 r + j
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(mine + (size_t)(16 * i + 4 * kq + q) * NW + c * 64 + 4 * r) =
              make_float4(acc[i][0][q], acc[i][1][q], acc[i][2][q], acc[i][3][q]);
      if (c == 0) {  // bias gradient = sum over this workgroup's pixels of dY
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v = (float)rs[i];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if (kq == 0) mine[32 * NW + 16 * i + r] = v;
        }
      }
    }
    __syncthreads();  // s_acc is reused by the next channel round
  }
}

// out[e] = sum over g < G of part[g][e] in g order (8 interleaved runs, then those in order): 32 elements x 8 runs per workgroup
__global__ void __launch_bounds__(256) jh_rb_parts_sum_kernel(const float* __restrict__ part, float* __restrict__ out_w, float* __restrict__ out_b, int n_w,
                                                              int n_all, int G) {
  __shared__ float s[8][32];
  const int el = threadIdx.x & 31, run = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float acc = 0.f;
  if (e < n_all) {
    int g = run;
    for (; g + 24 < G; g += 32) {
      const float v0 = part[(size_t)g * n_all + e], v1 = part[(size_t)(g + 8) * n_all + e], v2 = part[(size_t)(g + 16) * n_all + e],
                  v3 = part[(size_t)(g + 24) * n_all + e];
      acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; g < G; g += 8) acc += part[(size_t)g * n_all + e];
  }
  s[run][el] = acc;
  __syncthreads();
  if (run == 0 && e < n_all) {
    float v = s[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += s[k][el];
    if (e < n_w) out_w[e] = v;
    else out_b[e - n_w] = v;
  }
}

// ---------------------------------------------------------------------------------- optimizers
// hyper (device): {lr, beta1 | alpha, beta2, eps, step, centered}.  The step counter advances inside: every
// workgroup derives what it needs from step + 1 at its start; the last one to finish stores the new step.
// Optional global-norm clip (torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6)), applied
// to the gradient in place like the reference): `partial` holds per-workgroup sums of squares.
__global__ void __launch_bounds__(256) jh_rb_gradnorm_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ partial) {
  __shared__ float s_red[16];
  // 16-byte loads, four of them in flight per lane and round (13 MB of gradients at Ape-X: 18 us with one dword per lane and round)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int64_t n4 = n / 4, stride = (int64_t)gridDim.x * 256;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
    a0 = fmaf(v0.x, v0.x, a0); a0 = fmaf(v0.y, v0.y, a0); a0 = fmaf(v0.z, v0.z, a0); a0 = fmaf(v0.w, v0.w, a0);
    a1 = fmaf(v1.x, v1.x, a1); a1 = fmaf(v1.y, v1.y, a1); a1 = fmaf(v1.z, v1.z, a1); a1 = fmaf(v1.w, v1.w, a1);
    a2 = fmaf(v2.x, v2.x, a2); a2 = fmaf(v2.y, v2.y, a2); a2 = fmaf(v2.z, v2.z, a2); a2 = fmaf(v2.w, v2.w, a2);
    a3 = fmaf(v3.x, v3.x, a3); a3 = fmaf(v3.y, v3.y, a3); a3 = fmaf(v3.z, v3.z, a3); a3 = fmaf(v3.w, v3.w, a3);
  }
  for (; i < n4; i += stride) {
    const float4 v = g4[i];
    a0 = fmaf(v.x, v.x, a0); a0 = fmaf(v.y, v.y, a0); a0 = fmaf(v.z, v.z, a0); a0 = fmaf(v.w, v.w, a0);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
    const float v = g[4 * n4 + threadIdx.x];
    a0 = fmaf(v, v, a0);
  }
  float acc = (a0 + a1) + (a2 + a3);
  acc = jh_block_reduce(acc, s_red, JhAdd(), 0.f);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// A tail of the backward pass that rides in the optimizer's launch (jh_rbnet_backward_deferred; only without clipping: the global norm would
// need the finished gradient first): the sum of conv1's weight-gradient partials.  The bucket still ends up holding the complete gradient.
struct OptFuse {
  int n_c1;              // > 0: elements [0, n_c1) = conv1's weight + bias gradient, still `parts` partial sums in c1_part (n_c1 % 32 == 0)
  int parts;
  const float* c1_part;
  int extra_wgs;         // workgroups at the end of the grid that only do the conv1 part (beside the others' pass, not in front of it)
};
template <int OPT>  // 0 torch.optim.Adam, 1 torch.optim.RMSprop (momentum 0, optionally centered); no weight decay
__global__ void __launch_bounds__(256) jh_rb_optim_kernel(int64_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, float* __restrict__ hyper, unsigned* __restrict__ ticket,
                                                          const float* __restrict__ partial, int n_partial, float max_norm, OptFuse f) {
  __shared__ float s_red[16];
  __shared__ float s_bc[2];
  const float t_new = hyper[JH_HY_STEP] + 1.f;
  if (OPT == 0 && threadIdx.x == 0) jh_adam_bias_corrections(hyper, t_new, s_bc[0], s_bc[1]);  // in double, like torch (jh_common.h)
  float coef = 1.f;
  if (max_norm > 0.f) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
    const float total = sqrtf(jh_block_reduce(acc, s_red, JhAdd(), 0.f));
    coef = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  __syncthreads();
  const float lr = hyper[JH_HY_LR], b1 = hyper[JH_HY_B1], b2 = hyper[JH_HY_B2], eps = hyper[JH_HY_EPS];
  const float omb1 = hyper[JH_HY_OMB1], omb2 = hyper[JH_HY_OMB2];  // (float)(1.0 - beta | alpha): torch derives them in double
  const bool centered = hyper[JH_HY_BC1] != 0.f;
  float bc1 = 1.f, bc2s = 1.f;
  if (OPT == 0) { bc1 = s_bc[0]; bc2s = s_bc[1]; }
  const float step_size = lr / bc1;
  auto update = [&](float& pi, float& gi, float& mi, float& vi) {
    if (max_norm > 0.f) gi *= coef;
    if (OPT == 0) {
      mi = mi + omb1 * (gi - mi);  // exp_avg.lerp_(grad, 1 - beta1)
      vi = vi * b2 + omb2 * gi * gi;
      pi = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
    } else {
      vi = vi * b1 + omb1 * gi * gi;  // square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha)
      float avg;
      if (centered) {
        mi = mi + omb1 * (gi - mi);   // grad_avg.lerp_(grad, 1 - alpha)
        avg = sqrtf(vi - mi * mi) + eps;    // square_avg.addcmul(grad_avg, grad_avg, value=-1).sqrt_().add_(eps)
      } else {
        avg = sqrtf(vi) + eps;
      }
      pi = pi - lr * (gi / avg);
    }
  };
  // conv1's gradient still in partials: 32 elements per workgroup and round, summed in jh_rb_parts_sum_kernel's order (8 interleaved runs,
  // then those in order), written to the bucket, and stepped right here
  const int main_wgs = (int)gridDim.x - f.extra_wgs;
  if (f.n_c1 > 0 && (int)blockIdx.x >= main_wgs) {
    __shared__ float s_ps[8][32];
    const int el = threadIdx.x & 31, run = threadIdx.x >> 5;
    for (int e0 = ((int)blockIdx.x - main_wgs) * 32; e0 < f.n_c1; e0 += f.extra_wgs * 32) {
      const int e = e0 + el;
      float acc = 0.f;
      int q = run;
      for (; q + 24 < f.parts; q += 32) {
        const float v0 = f.c1_part[(size_t)q * f.n_c1 + e], v1 = f.c1_part[(size_t)(q + 8) * f.n_c1 + e], v2 = f.c1_part[(size_t)(q + 16) * f.n_c1 + e],
                    v3 = f.c1_part[(size_t)(q + 24) * f.n_c1 + e];
        acc = (((acc + v0) + v1) + v2) + v3;
      }
      for (; q < f.parts; q += 8) acc += f.c1_part[(size_t)q * f.n_c1 + e];
      s_ps[run][el] = acc;
      __syncthreads();
      if (run == 0) {
        float gi = s_ps[0][el];
#pragma unroll
        for (int k = 1; k < 8; ++k) gi += s_ps[k][el];
        float pi = p[e], mi = m[e], vi = v[e];
        update(pi, gi, mi, vi);
        p[e] = pi; g[e] = gi; v[e] = vi;
        if (OPT == 0 || centered) m[e] = mi;
      }
      __syncthreads();
    }
  }
  // 16-byte accesses: the buckets are 16-byte aligned and n is a multiple of 4 (segment offsets are)
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)(f.n_c1 >> 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4 && (int)blockIdx.x < main_wgs; i += (int64_t)main_wgs * 256) {
    float4 p4 = reinterpret_cast<float4*>(p)[i], g4 = reinterpret_cast<float4*>(g)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
    update(p4.x, g4.x, m4.x, v4.x);
    update(p4.y, g4.y, m4.y, v4.y);
    update(p4.z, g4.z, m4.z, v4.z);
    update(p4.w, g4.w, m4.w, v4.w);
    reinterpret_cast<float4*>(p)[i] = p4;
    if (max_norm > 0.f) reinterpret_cast<float4*>(g)[i] = g4;
    if (OPT == 0 || centered) reinterpret_cast<float4*>(m)[i] = m4;
    reinterpret_cast<float4*>(v)[i] = v4;
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n && (int)blockIdx.x < main_wgs; i += (int64_t)main_wgs * 256) {
    float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
    update(pi, gi, mi, vi);
    p[i] = pi; g[i] = gi; m[i] = mi; v[i] = vi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // "the last workgroup to finish stores the new step" as a TWO-LEVEL ticket (round 5): eight counters on separate cache lines, one
    // per residue of blockIdx mod 8, and one on top of them.  Device-scope atomics on ONE address retire at ~9-15 ns apiece, and the
    // workgroups of this launch finish together: with a single counter the launch grew by that much per workgroup (22.5 us at 512
    // workgroups, 30.5 at 1024, 48 at 3072 for the same 3 M parameters -- profiles/r05_optim_grid_sweep.txt), i.e. its tail WAS the queue.
    const unsigned k = blockIdx.x & 7u;
    const unsigned nk = (gridDim.x - k + 7u) >> 3;  // workgroups with this residue
    const unsigned tk = __hip_atomic_fetch_add(ticket + 32u * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == nk - 1) {
      __hip_atomic_store(ticket + 32u * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned groups = gridDim.x < 8u ? gridDim.x : 8u;
      const unsigned top = __hip_atomic_fetch_add(ticket + 256u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (top == groups - 1) {
        __hip_atomic_store(ticket + 256u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hyper[JH_HY_STEP] = t_new;
      }
    }
  }
}

}  // namespace

enum {
  SEG_W1, SEG_B1, SEG_W2, SEG_B2, SEG_W3, SEG_B3, SEG_WL, SEG_BL,
  SEG_MU_AV1, SEG_SIG_AV1, SEG_MUB_AV1, SEG_SIGB_AV1, SEG_MU_A2, SEG_SIG_A2, SEG_MUB_A2, SEG_SIGB_A2,
  SEG_MU_V2, SEG_SIG_V2, SEG_MUB_V2, SEG_SIGB_V2, SEG_COUNT
};

constexpr int kC1Parts = 256;  // conv1 weight gradient: at most this many K slices (workgroups)
struct jh_rbnet {
  jh_ctx* ctx = nullptr;
  int cnn = 0, Cin = 0, Hin = 0, Win = 0, hidden = 0, A = 0, K = 0, NA = 0, maxB = 0, F = 0;
  // kind 0 rainbow: head -> l -> noisy a1|v1 -> noisy a2, v2 -> dueling over K atoms   (network/rainbow.py:8-94)
  //      1 dueling: head -> l1_a|l1_v -> l2_a, l2_v -> dueling combine (K = 1)         (network/dueling.py:8-35)
  //      2 q:       head -> l -> q                                                      (network/q_network.py:8-20)
  int kind = 0, noisy = 1, dueling = 1, has_l = 1, has_av1 = 1, in1 = 0, noise_independent = 0;
  float* norm_partial = nullptr;
  int NA4 = 0, K4 = 0;
  ConvGeom c1{}, c2{}, c3{};
  int P1 = 0, P2 = 0, P3 = 0;  // output pixels per sample
  int64_t seg_off[SEG_COUNT] = {0};
  int seg_rows[SEG_COUNT] = {0}, seg_cols[SEG_COUNT] = {0};
  int64_t n_params = 0;
  float *params = nullptr, *target = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
  float* hyper = nullptr;
  unsigned* ticket = nullptr;
  NoisyDims nd{};
  int64_t n_noisy = 0;
  float* weff = nullptr;                                // [3][set_stride]
  float *act1[2] = {nullptr, nullptr}, *act2[2] = {nullptr, nullptr}, *feat[2] = {nullptr, nullptr}, *h[2] = {nullptr, nullptr};
  float *hav = nullptr, *xa = nullptr, *xv = nullptr;  // [3][maxB][...]
  float *dxa = nullptr, *dxv = nullptr, *dhav = nullptr, *dh = nullptr, *dfeat = nullptr, *dcol = nullptr, *dact2 = nullptr, *dact1 = nullptr;
  float* ws = nullptr;
  size_t ws_floats = 0;
  unsigned* cnt = nullptr;
  int cnt_slots = 0;
  float* c1_part = nullptr;    // conv1 weight-gradient partials [kC1Parts][32 * 64 Cin + 32]
  const void* last_x = nullptr;  // input of the last learn_forward (backward of layer 1 reads it again)
  int last_x_u8 = 0, last_B = 0;
  const float* last_noise = nullptr;
  const float* prepared_noise = nullptr;  // jh_rbnet_prepare_noise materialised the three learn() weight sets for this draw already
  int raw_heads = 0;    // the last learn_heads left xa / xv uncombined (jh_rbnet_learn_heads_raw): jh_rbnet_c51_step combines them
  int dx_ready = 0;     // dxa / dxv already hold the gradient pulled through the dueling combine (jh_rbnet_c51_step)
  int pend_parts = 0;   // conv1 weight-gradient partials not yet summed into the bucket (jh_rbnet_backward_deferred): their count
  std::vector<void*> owned;
};

static int rb_alloc(jh_rbnet* n, void** out, size_t bytes, bool zero) {
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) return jh_fail(JH_ERR_NOMEM, "jh_rbnet: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  n->owned.push_back(*out);
  if (zero) JH_HIP(hipMemset(*out, 0, bytes));
  return JH_OK;
}

static int launch_tgemm(jh_rbnet* net, const char* name, TGemm* probs, int n, hipStream_t st) {
  TGemmWorkspace w;
  w.ws = net->ws; w.ws_floats = net->ws_floats; w.cnt = net->cnt; w.cnt_slots = net->cnt_slots;
  return jh_tgemm_launch(w, name, probs, n, st);
}

static inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }

static int rb_layout(jh_rbnet* n, int32_t kind, int32_t head_cnn, int32_t c_or_s, int32_t h_in, int32_t w_in, int32_t hidden, int32_t A, int32_t K,
                     int32_t max_batch) {
  JH_ARG(kind >= 0 && kind <= 3);
  if (kind == 3) {  // rainbow with independent Gaussian noise (utils.py:72-79)
    n->noise_independent = 1;
    kind = 0;
  }
  JH_ARG(hidden > 0 && hidden % 4 == 0 && A > 0 && max_batch > 0 && c_or_s > 0);
  JH_ARG(kind == 0 ? K > 1 : K == 1);
  n->kind = kind; n->noisy = kind == 0; n->dueling = kind != 2; n->has_l = kind != 1; n->has_av1 = kind != 2;
  n->cnn = head_cnn ? 1 : 0;
  n->Cin = c_or_s; n->Hin = h_in; n->Win = w_in; n->hidden = hidden; n->A = A; n->K = K; n->NA = A * K; n->maxB = max_batch;
  n->NA4 = (int)up4(n->NA); n->K4 = (int)up4(K);
  const int H = hidden;
  if (n->cnn) {
    if (h_in < 36 || w_in < 36) return jh_fail(JH_ERR_ARG, "cnn head needs images >= 36x36 (head.py:26), got %dx%d", h_in, w_in);
    n->c1 = ConvGeom{c_or_s, h_in, w_in, (h_in - 8) / 4 + 1, (w_in - 8) / 4 + 1, 8, 8, 4, nullptr, nullptr};
    n->c2 = ConvGeom{32, n->c1.OH, n->c1.OW, (n->c1.OH - 4) / 2 + 1, (n->c1.OW - 4) / 2 + 1, 4, 4, 2, nullptr, nullptr};
    n->c3 = ConvGeom{64, n->c2.OH, n->c2.OW, n->c2.OH - 3 + 1, n->c2.OW - 3 + 1, 3, 3, 1, nullptr, nullptr};
    n->P1 = n->c1.OH * n->c1.OW; n->P2 = n->c2.OH * n->c2.OW; n->P3 = n->c3.OH * n->c3.OW;
    n->F = 64 * n->P3;
  } else {
    n->F = H;
  }
  auto seg = [&](int id, int rows, int cols) { n->seg_rows[id] = rows; n->seg_cols[id] = cols; };
  if (n->cnn) {
    seg(SEG_W1, 32, c_or_s * 64); seg(SEG_B1, 1, 32);
    seg(SEG_W2, 64, 16 * 32); seg(SEG_B2, 1, 64);
    seg(SEG_W3, 64, 9 * 64); seg(SEG_B3, 1, 64);
  } else {
    seg(SEG_W1, H, c_or_s); seg(SEG_B1, 1, H);
  }
  n->in1 = n->has_l ? H : n->F;
  if (n->has_l) { seg(SEG_WL, H, n->F); seg(SEG_BL, 1, H); }
  if (n->has_av1) {
    seg(SEG_MU_AV1, 2 * H, n->in1); seg(SEG_MUB_AV1, 1, 2 * H);
    if (n->noisy) { seg(SEG_SIG_AV1, 2 * H, n->in1); seg(SEG_SIGB_AV1, 1, 2 * H); }
  }
  seg(SEG_MU_A2, n->NA, H); seg(SEG_MUB_A2, 1, n->NA);
  if (n->noisy) { seg(SEG_SIG_A2, n->NA, H); seg(SEG_SIGB_A2, 1, n->NA); }
  if (n->dueling) {
    seg(SEG_MU_V2, K, H); seg(SEG_MUB_V2, 1, K);
    if (n->noisy) { seg(SEG_SIG_V2, K, H); seg(SEG_SIGB_V2, 1, K); }
  }
  int64_t off = 0;
  for (int i = 0; i < SEG_COUNT; ++i) {
    n->seg_off[i] = off;
    off = up4(off + (int64_t)n->seg_rows[i] * n->seg_cols[i]);
  }
  n->n_params = off;
  NoisyDims& d = n->nd;
  d.H = H; d.NA = n->NA; d.K = K;
  d.o_av1 = 0;
  d.o_bav1 = up4(d.o_av1 + (int64_t)2 * H * H);
  d.o_a2 = up4(d.o_bav1 + 2 * H);
  d.o_ba2 = up4(d.o_a2 + (int64_t)n->NA * H);
  d.o_v2 = up4(d.o_ba2 + n->NA);
  d.o_bv2 = up4(d.o_v2 + (int64_t)K * H);
  d.set_stride = up4(d.o_bv2 + K);
  d.p_mu_av1 = n->seg_off[SEG_MU_AV1]; d.p_sig_av1 = n->seg_off[SEG_SIG_AV1]; d.p_mub_av1 = n->seg_off[SEG_MUB_AV1]; d.p_sigb_av1 = n->seg_off[SEG_SIGB_AV1];
  d.p_mu_a2 = n->seg_off[SEG_MU_A2]; d.p_sig_a2 = n->seg_off[SEG_SIG_A2]; d.p_mub_a2 = n->seg_off[SEG_MUB_A2]; d.p_sigb_a2 = n->seg_off[SEG_SIGB_A2];
  d.p_mu_v2 = n->seg_off[SEG_MU_V2]; d.p_sig_v2 = n->seg_off[SEG_SIG_V2]; d.p_mub_v2 = n->seg_off[SEG_MUB_V2]; d.p_sigb_v2 = n->seg_off[SEG_SIGB_V2];
  d.independent = n->noise_independent;
  d.noise_len = d.independent ? (int64_t)2 * ((int64_t)H * H + H) + (int64_t)H * n->NA + n->NA + (int64_t)H * K + K : (int64_t)6 * H + n->NA + K;
  n->n_noisy = (int64_t)2 * H * H + (int64_t)n->NA * H + (int64_t)K * H + 2 * H + n->NA + K;
  return JH_OK;
}

JH_EXPORT int64_t jh_rbnet_param_count_for(int32_t kind, int32_t head_cnn, int32_t c_or_s, int32_t h_in, int32_t w_in, int32_t hidden, int32_t A, int32_t K) {
  jh_rbnet tmp;
  if (rb_layout(&tmp, kind, head_cnn, c_or_s, h_in, w_in, hidden, A, K, 1)) return -1;
  return tmp.n_params;
}

JH_EXPORT int jh_rbnet_create(jh_ctx* ctx, int32_t kind, int32_t head_cnn, int32_t c_or_s, int32_t h_in, int32_t w_in, int32_t hidden, int32_t A, int32_t K,
                              int32_t max_batch, float* d_params, float* d_target, float* d_grads, float* d_m, float* d_v, jh_rbnet** out) {
  JH_ARG(ctx && out && d_params && d_target && d_grads && d_m && d_v);
  JH_HIP(hipSetDevice(ctx->device));
  jh_rbnet* n = new jh_rbnet();
  n->ctx = ctx;
  int rc = rb_layout(n, kind, head_cnn, c_or_s, h_in, w_in, hidden, A, K, max_batch);
  if (rc) {
    delete n;
    return rc;
  }
  n->params = d_params; n->target = d_target; n->grads = d_grads; n->m = d_m; n->v = d_v;
  const NoisyDims& d = n->nd;
  const int H = hidden;
  const size_t B = (size_t)max_batch;
  auto A4 = [&](float** p, size_t floats, bool zero = true) { if (!rc) rc = rb_alloc(n, (void**)p, floats * sizeof(float), zero); };
  A4(&n->hyper, JH_HY_FLOATS);
  A4(&n->norm_partial, 256);
  if (!rc) rc = rb_alloc(n, (void**)&n->ticket, 2048, true);  // jh_rb_optim_kernel: eight counters 128 bytes apart + the one on top of them
  A4(&n->weff, 3 * (size_t)d.set_stride);
  for (int s = 0; s < 2; ++s) {
    const size_t rows = s == 0 ? 2 * B : B;
    if (n->cnn) {
      A4(&n->act1[s], rows * n->P1 * 32);
      A4(&n->act2[s], rows * n->P2 * 64);
    }
    A4(&n->feat[s], rows * n->F);
    A4(&n->h[s], rows * H);
  }
  A4(&n->hav, 3 * B * 2 * H); A4(&n->xa, 3 * B * n->NA4); A4(&n->xv, 3 * B * n->K4);
  A4(&n->dxa, B * n->NA4); A4(&n->dxv, B * n->K4); A4(&n->dhav, B * 2 * H); A4(&n->dh, B * H); A4(&n->dfeat, B * n->F);
  if (n->cnn) {
    size_t dc = B * n->P3 * 576;
    if (B * n->P2 * 512 > dc) dc = B * n->P2 * 512;
    A4(&n->dcol, dc); A4(&n->dact2, B * n->P2 * 64); A4(&n->dact1, B * n->P1 * 32);
    A4(&n->c1_part, (size_t)kC1Parts * (32 * n->Cin * 64 + 32), false);
  }
  if (n->cnn && !rc) {
    ConvGeom* gs[3] = {&n->c1, &n->c2, &n->c3};
    for (int li = 0; li < 3 && !rc; ++li) {
      ConvGeom& c = *gs[li];
      const bool nchw = li == 0;
      const int n_pix = 2 * max_batch * c.OH * c.OW, n_tap = c.C * c.KH * c.KW;
      std::vector<int> tab((size_t)n_pix + n_tap);
      for (int pix = 0; pix < n_pix; ++pix) {
        const int ox = pix % c.OW, t = pix / c.OW, oy = t % c.OH, b = t / c.OH;
        const int64_t off = nchw ? ((int64_t)b * c.C * c.H + oy * c.S) * c.W + ox * c.S : (((int64_t)b * c.H + oy * c.S) * c.W + ox * c.S) * c.C;
        if (off > 0x7fffffff - (int64_t)c.C * c.H * c.W) rc = jh_fail(JH_ERR_ARG, "batch x image too large for 32-bit im2col offsets");
        tab[pix] = (int)off;
      }
      for (int q = 0; q < n_tap; ++q) {
        if (nchw) {
          const int kx = q % c.KW, t2 = q / c.KW, ky = t2 % c.KH, ch = t2 / c.KH;
          tab[n_pix + q] = (ch * c.H + ky) * c.W + kx;
        } else {
          const int ch = q % c.C, t2 = q / c.C, kx = t2 % c.KW, ky = t2 / c.KW;
          tab[n_pix + q] = (ky * c.W + kx) * c.C + ch;
        }
      }
      int* d_tab = nullptr;
      if (!rc) rc = rb_alloc(n, (void**)&d_tab, tab.size() * sizeof(int), false);
      if (!rc && hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) rc = jh_fail(JH_ERR_HIP, "im2col table upload failed");
      c.pix_tab = d_tab;
      c.tap_tab = d_tab + n_pix;
    }
  }
  n->ws_floats = (size_t)8 << 20;  // 32 MB of split-K partials
  A4(&n->ws, n->ws_floats, false);
  n->cnt_slots = 8192;
  if (!rc) rc = rb_alloc(n, (void**)&n->cnt, sizeof(unsigned) * (size_t)n->cnt_slots * kTgemmCntStride, true);
  if (rc) {
    for (void* p : n->owned) (void)hipFree(p);
    delete n;
    return rc;
  }
  float hy[JH_HY_FLOATS];
  jh_hyper_fill(hy, 1e-3, 0.9, 0.999, 1e-8, 0.0);
  JH_HIP(hipMemcpy(n->hyper, hy, sizeof(hy), hipMemcpyHostToDevice));
  JH_HIP(hipDeviceSynchronize());
  *out = n;
  return JH_OK;
}

JH_EXPORT void jh_rbnet_destroy(jh_rbnet* n) {
  if (!n) return;
  (void)hipSetDevice(n->ctx->device);
  (void)hipDeviceSynchronize();
  for (void* p : n->owned) (void)hipFree(p);
  delete n;
}

JH_EXPORT int64_t jh_rbnet_param_count(const jh_rbnet* n) { return n ? n->n_params : -1; }
JH_EXPORT int32_t jh_rbnet_segment_count(void) { return SEG_COUNT; }
JH_EXPORT int jh_rbnet_segment(const jh_rbnet* n, int32_t i, int64_t* offset, int32_t* rows, int32_t* cols) {
  JH_ARG(n && i >= 0 && i < SEG_COUNT && offset && rows && cols);
  *offset = n->seg_off[i]; *rows = n->seg_rows[i]; *cols = n->seg_cols[i];
  return JH_OK;
}
JH_EXPORT int64_t jh_rbnet_noise_len(const jh_rbnet* n) { return n ? n->nd.noise_len : -1; }

JH_EXPORT int jh_rbnet_set_hyper(jh_rbnet* n, double lr, double beta1, double beta2, double eps, int64_t step, int32_t centered, jh_stream stream) {
  JH_ARG(n != nullptr);
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(n->ctx, 64, &slab);
  if (rc) return rc;
  float* h = (float*)slab->host;
  jh_hyper_fill(h, lr, beta1, beta2, eps, (double)step);
  h[JH_HY_BC1] = centered ? 1.f : 0.f;
  JH_HIP(hipMemcpyAsync(n->hyper, slab->dev, JH_HY_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, jh_s(stream)));
  return jh_ctx_slab_release(n->ctx, slab, jh_s(stream));
}
JH_EXPORT int jh_rbnet_set_lr(jh_rbnet* n, double lr, jh_stream stream) {
  JH_ARG(n != nullptr);
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(n->ctx, 16, &slab);
  if (rc) return rc;
  *(float*)slab->host = (float)lr;
  JH_HIP(hipMemcpyAsync(n->hyper, slab->dev, sizeof(float), hipMemcpyDeviceToDevice, jh_s(stream)));
  return jh_ctx_slab_release(n->ctx, slab, jh_s(stream));
}
JH_EXPORT int jh_rbnet_sync_target(jh_rbnet* n, jh_stream stream) {
  JH_ARG(n != nullptr);
  JH_HIP(hipMemcpyAsync(n->target, n->params, sizeof(float) * (size_t)n->n_params, hipMemcpyDeviceToDevice, jh_s(stream)));
  return JH_OK;
}

// trunk (head + l) for up to two (parameter set, input rows) pairs -> h[slot]
struct TrunkJob {
  const float* P;
  const void* x;
  int rows, slot;
};
static int rb_trunk(jh_rbnet* n, const TrunkJob* jobs, int nj, int x_u8, hipStream_t st) {
  const int H = n->hidden;
  TGemm g[2];
  if (n->cnn) {
    for (int j = 0; j < nj; ++j) {
      const TrunkJob& J = jobs[j];
      g[j] = mk_gemm(J.rows * n->P1, 32, n->Cin * 64, op_conv(OP_NCHW_K, J.x, x_u8, n->c1), op_dense(OP_KCONT, J.P + n->seg_off[SEG_W1], n->Cin * 64),
                     n->act1[J.slot], 32, TEPI_BIAS_RELU, J.P + n->seg_off[SEG_B1]);
    }
    // uint8 frames in the head's own geometry: the dedicated kernel (jh_rb_conv1_fwd_kernel); anything else is a GEMM like the other layers
    static const bool kC1Own = !(getenv("JH_RB_CONV1_FWD") && atoi(getenv("JH_RB_CONV1_FWD")) == 0);
    const ConvGeom& c1 = n->c1;
    int rc = JH_OK;
    if (kC1Own && x_u8 && c1.KH == 8 && c1.KW == 8 && c1.S == 4 && c1.W % 4 == 0 && n->Cin == 4 && n->P1 < (1 << 20)) {
      C1Fwd a{};
      int imgs = 0;
      for (int j = 0; j < nj; ++j) imgs += jobs[j].rows;
      a.nj = nj; a.H = c1.H; a.W = c1.W; a.OW = c1.OW; a.P = n->P1; a.inv_ow = 1.0f / (float)c1.OW;
      a.tiles_per_img = (n->P1 + 15) / 16;
      // >= 256 workgroups once there are 43 frames.  64-191 frames (learn() at B = 32: 96 frames x 25 tiles): ONE tile per wave.  With three
      // workgroups per frame (288 on 256 CUs, waves of 3 | 2 | 2 | 2 tiles) the 32 CUs that got two workgroups ran six tiles on one SIMD
      // while the chip-wide average is 2.3: the launch lasted as long as those (round 5: a tile is 128 MFMAs = 1.7 us of one SIMD)
      a.wgs_per_img = imgs >= 512 ? 1 : (imgs >= 192 ? 2 : (imgs >= 64 ? (a.tiles_per_img + 3) / 4 : 6));
      a.tiles_per_wg = (a.tiles_per_img + a.wgs_per_img - 1) / a.wgs_per_img;
      a.wgs_per_img = (a.tiles_per_img + a.tiles_per_wg - 1) / a.tiles_per_wg;
      int wgs = 0;
      for (int j = 0; j < nj; ++j) {
        const TrunkJob& J = jobs[j];
        a.j[j] = C1Job{(const uint8_t*)J.x, J.P + n->seg_off[SEG_W1], J.P + n->seg_off[SEG_B1], n->act1[J.slot], wgs};
        wgs += J.rows * a.wgs_per_img;
      }
      JH_LAUNCH_IDEM("jh_rb_conv1_fwd_kernel", 2.0 * 32 * 256 * (double)imgs * n->P1, jh_rb_conv1_fwd_kernel<4>, dim3(wgs), dim3(256), 0, st, a);
      JH_LAUNCH_CHECK();
    } else {
      rc = launch_tgemm(n, "jh_tgemm_conv1_fwd", g, nj, st);
    }
    if (rc) return rc;
    for (int j = 0; j < nj; ++j) {
      const TrunkJob& J = jobs[j];
      g[j] = mk_gemm(J.rows * n->P2, 64, 512, op_conv(OP_NHWC_K, n->act1[J.slot], 0, n->c2), op_dense(OP_KCONT, J.P + n->seg_off[SEG_W2], 512),
                     n->act2[J.slot], 64, TEPI_BIAS_RELU, J.P + n->seg_off[SEG_B2]);
    }
    rc = launch_tgemm(n, "jh_tgemm_conv2_fwd", g, nj, st);
    if (rc) return rc;
    for (int j = 0; j < nj; ++j) {
      const TrunkJob& J = jobs[j];
      g[j] = mk_gemm(J.rows * n->P3, 64, 576, op_conv(OP_NHWC_K, n->act2[J.slot], 0, n->c3), op_dense(OP_KCONT, J.P + n->seg_off[SEG_W3], 576),
                     n->feat[J.slot], 64, TEPI_BIAS_RELU, J.P + n->seg_off[SEG_B3]);
    }
    rc = launch_tgemm(n, "jh_tgemm_conv3_fwd", g, nj, st);
    if (rc) return rc;
  } else {
    if (x_u8) return jh_fail(JH_ERR_ARG, "mlp head takes fp32 observations");
    for (int j = 0; j < nj; ++j) {
      const TrunkJob& J = jobs[j];
      g[j] = mk_gemm(J.rows, H, n->Cin, op_dense(OP_KCONT, (const float*)J.x, n->Cin), op_dense(OP_KCONT, J.P + n->seg_off[SEG_W1], n->Cin),
                     n->feat[J.slot], H, TEPI_BIAS_RELU, J.P + n->seg_off[SEG_B1]);
    }
    int rc = launch_tgemm(n, "jh_tgemm_head_fwd", g, nj, st);
    if (rc) return rc;
  }
  if (!n->has_l) return JH_OK;  // dueling.py: the streams read the head's features directly
  for (int j = 0; j < nj; ++j) {
    const TrunkJob& J = jobs[j];
    g[j] = mk_gemm(J.rows, H, n->F, op_dense(OP_KCONT, n->feat[J.slot], n->F), op_dense(OP_KCONT, J.P + n->seg_off[SEG_WL], n->F), n->h[J.slot], H,
                   TEPI_BIAS_RELU, J.P + n->seg_off[SEG_BL]);
  }
  return launch_tgemm(n, "jh_tgemm_fc_fwd", g, nj, st);
}

// weights of the stream layers as the GEMMs see them: the materialised noisy set, or the parameters themselves
struct StreamW {
  const float *av1, *bav1, *a2, *ba2, *v2, *bv2;
};
static StreamW stream_w(const jh_rbnet* n, const float* P, int set) {
  StreamW w{};
  if (n->noisy) {
    const float* W = n->weff + (size_t)set * n->nd.set_stride;
    w.av1 = W + n->nd.o_av1; w.bav1 = W + n->nd.o_bav1; w.a2 = W + n->nd.o_a2; w.ba2 = W + n->nd.o_ba2; w.v2 = W + n->nd.o_v2; w.bv2 = W + n->nd.o_bv2;
  } else {
    w.av1 = P + n->seg_off[SEG_MU_AV1]; w.bav1 = P + n->seg_off[SEG_MUB_AV1]; w.a2 = P + n->seg_off[SEG_MU_A2]; w.ba2 = P + n->seg_off[SEG_MUB_A2];
    w.v2 = P + n->seg_off[SEG_MU_V2]; w.bv2 = P + n->seg_off[SEG_MUB_V2];
  }
  return w;
}

// noisy dueling heads for up to three (parameter set, noise draw, hidden rows) triples -> logits
struct HeadJob {
  const float* P;
  const float* noise;
  const float* h;  // [B][H]
  float* logits;   // [B][A][K]
};
static int rb_heads(jh_rbnet* n, const HeadJob* jobs, int nj, int B, hipStream_t st, bool raw = false) {
  const int H = n->hidden, NA = n->NA, K = n->K, in1 = n->in1;
  const NoisyDims& d = n->nd;
  const bool prepared = n->noisy && nj == 3 && n->prepared_noise != nullptr && n->prepared_noise == jobs[0].noise;
  n->prepared_noise = nullptr;
  if (n->noisy && !prepared) {
    NoiseSets ns{};
    ns.n_sets = nj;
    for (int j = 0; j < nj; ++j) {
      ns.params[j] = jobs[j].P; ns.noise[j] = jobs[j].noise; ns.weff[j] = n->weff + (size_t)j * d.set_stride;
    }
    int64_t blocks = (n->n_noisy + 255) / 256;
    if (blocks > 1024) blocks = 1024;  // (4096 blocks -- one element per thread -- measured 2.6 us SLOWER at Rainbow's 655 k noisy weights x 3 sets, round 5)
    JH_LAUNCH(jh_rb_noise_kernel, dim3((unsigned)blocks, nj), dim3(256), 0, st, d, ns, n->n_noisy);
    JH_LAUNCH_CHECK();
  }
  TGemm g[6];
  const size_t B_ = (size_t)n->maxB;
  int rc;
  if (n->has_av1) {
    for (int j = 0; j < nj; ++j) {
      const StreamW w = stream_w(n, jobs[j].P, j);
      g[j] = mk_gemm(B, 2 * H, in1, op_dense(OP_KCONT, jobs[j].h, in1), op_dense(OP_KCONT, w.av1, in1), n->hav + j * B_ * 2 * H, 2 * H, TEPI_BIAS_RELU, w.bav1);
    }
    rc = launch_tgemm(n, "jh_tgemm_stream1_fwd", g, nj, st);
    if (rc) return rc;
  }
  int ng = 0;
  for (int j = 0; j < nj; ++j) {
    const StreamW w = stream_w(n, jobs[j].P, j);
    const float* in_a = n->has_av1 ? n->hav + j * B_ * 2 * H : jobs[j].h;  // q-network: the q layer reads h
    const int ld_in = n->has_av1 ? 2 * H : H;
    float* out_a = n->dueling ? n->xa + j * B_ * n->NA4 : jobs[j].logits;
    g[ng++] = mk_gemm(B, NA, H, op_dense(OP_KCONT, in_a, ld_in), op_dense(OP_KCONT, w.a2, H), out_a, n->dueling ? n->NA4 : NA, TEPI_BIAS, w.ba2);
    if (n->dueling)
      g[ng++] = mk_gemm(B, K, H, op_dense(OP_KCONT, in_a + H, ld_in), op_dense(OP_KCONT, w.v2, H), n->xv + j * B_ * n->K4, n->K4, TEPI_BIAS, w.bv2);
  }
  rc = launch_tgemm(n, "jh_tgemm_stream2_fwd", g, ng, st);
  if (rc) return rc;
  if (!n->dueling || raw) return JH_OK;
  DuelSets ds{};
  for (int j = 0; j < nj; ++j) {
    ds.xa[j] = n->xa + j * B_ * n->NA4; ds.xv[j] = n->xv + j * B_ * n->K4; ds.out[j] = jobs[j].logits;
  }
  JH_LAUNCH(jh_rb_duel_fwd_kernel, dim3((B * K + 255) / 256, nj), dim3(256), 0, st, ds, B, n->A, K, n->NA4, n->K4);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// network(x, is_train) for acting / evaluation: rows <= max_batch.  which: 0 online, 1 target.
// noise: one noise set (jh_rbnet_noise_len floats of N(0,1)) or null for is_train = False.
JH_EXPORT int jh_rbnet_forward(jh_rbnet* n, int32_t which, const void* d_x, int32_t x_dtype, int32_t rows, const float* d_noise, float* d_logits,
                               jh_stream stream) {
  JH_ARG(n && d_x && d_logits);
  JH_ARG(rows > 0 && rows <= n->maxB && (which == 0 || which == 1));
  JH_ARG(x_dtype == JH_U8 || x_dtype == JH_F32);
  hipStream_t st = jh_s(stream);
  const float* P = which == 0 ? n->params : n->target;
  TrunkJob tj{P, d_x, rows, 0};
  int rc = rb_trunk(n, &tj, 1, x_dtype == JH_U8, st);
  if (rc) return rc;
  HeadJob hj{P, n->noisy ? d_noise : nullptr, n->has_l ? n->h[0] : n->feat[0], d_logits};
  return rb_heads(n, &hj, 1, rows, st);
}

// The forward of an on-policy learner (PPO on the CNN head, ppo.py:127-135): network(x) of the online parameters for B <= max_batch rows with every
// activation kept for jh_rbnet_backward (which then takes d(loss)/d(outputs) [B][A][K] of THESE rows).  Networks without noise only (q, dueling).
JH_EXPORT int jh_rbnet_forward_keep(jh_rbnet* n, const void* d_x, int32_t x_dtype, int32_t B, float* d_logits, jh_stream stream) {
  JH_ARG(n && d_x && d_logits);
  if (n->noisy) return jh_fail(JH_ERR_ARG, "jh_rbnet_forward_keep: a noisy network's learn() is jh_rbnet_learn_forward");
  int rc = jh_rbnet_forward(n, 0, d_x, x_dtype, B, nullptr, d_logits, stream);
  if (rc) return rc;
  n->last_x = d_x; n->last_x_u8 = x_dtype == JH_U8; n->last_B = B;
  n->last_noise = nullptr; n->raw_heads = 0; n->dx_ready = 0;
  return JH_OK;
}

// The three forwards of Rainbow.learn (agent/rainbow.py:160-186) in one pass:
//   d_x = [state (B rows); next_state (B rows)]  (one contiguous batch of 2B observations)
//   logits[0] = online(state; noise 0)   logits[1] = online(next_state; noise 1)   logits[2] = target(next_state; noise 2)
// The online trunk runs once over all 2B rows; the target trunk shares the launches (grouped GEMMs).
// The two halves separately, so that a caller can run what does not depend on the batch on ANOTHER stream meanwhile (round 4: the
// NoisyNet weight sets W = mu + sig * eps of the three forwards only need the noise draw; materialising them costs a 12-us launch that
// used to sit in front of the trunk): jh_rbnet_prepare_noise (any stream) || jh_rbnet_learn_trunk; join; jh_rbnet_learn_heads.
JH_EXPORT int jh_rbnet_prepare_noise(jh_rbnet* n, const float* d_noise, jh_stream stream) {
  JH_ARG(n != nullptr);
  if (!n->noisy) return JH_OK;
  JH_ARG(d_noise != nullptr);
  const NoisyDims& d = n->nd;
  const int64_t L = d.noise_len;
  NoiseSets ns{};
  ns.n_sets = 3;
  const float* P[3] = {n->params, n->params, n->target};
  for (int j = 0; j < 3; ++j) { ns.params[j] = P[j]; ns.noise[j] = d_noise + j * L; ns.weff[j] = n->weff + (size_t)j * d.set_stride; }
  int64_t blocks = (n->n_noisy + 255) / 256;
  if (blocks > 1024) blocks = 1024;  // (4096 blocks -- one element per thread -- measured 2.6 us SLOWER at Rainbow's 655 k noisy weights x 3 sets, round 5)
  JH_LAUNCH(jh_rb_noise_kernel, dim3((unsigned)blocks, 3), dim3(256), 0, jh_s(stream), d, ns, n->n_noisy);
  JH_LAUNCH_CHECK();
  n->prepared_noise = d_noise;
  return JH_OK;
}

JH_EXPORT int jh_rbnet_learn_trunk(jh_rbnet* n, const void* d_x, int32_t x_dtype, int32_t B, jh_stream stream) {
  JH_ARG(n && d_x);
  JH_ARG(B > 0 && B <= n->maxB);
  JH_ARG(x_dtype == JH_U8 || x_dtype == JH_F32);
  const size_t row_elems = n->cnn ? (size_t)n->Cin * n->Hin * n->Win : (size_t)n->Cin;
  const size_t esz = x_dtype == JH_U8 ? 1 : 4;
  const void* x_next = (const char*)d_x + (size_t)B * row_elems * esz;
  TrunkJob tj[2] = {{n->params, d_x, 2 * B, 0}, {n->target, x_next, B, 1}};
  int rc = rb_trunk(n, tj, 2, x_dtype == JH_U8, jh_s(stream));
  if (rc) return rc;
  n->last_x = d_x; n->last_x_u8 = x_dtype == JH_U8; n->last_B = B;
  return JH_OK;
}

static int rb_learn_heads(jh_rbnet* n, int32_t B, const float* d_noise, float* d_logits, bool raw, jh_stream stream) {
  JH_ARG(n && (d_logits || raw));
  JH_ARG(d_noise || !n->noisy);
  JH_ARG(B > 0 && B == n->last_B);
  if (raw && !n->dueling) return jh_fail(JH_ERR_ARG, "jh_rbnet_learn_heads_raw: a dueling network's call");
  const int64_t L = n->noisy ? n->nd.noise_len : 0;
  const size_t lsz = (size_t)B * n->NA;
  const float* nz = n->noisy ? d_noise : nullptr;
  float* const* sin = n->has_l ? n->h : n->feat;  // what the streams read
  HeadJob hj[3] = {{n->params, nz, sin[0], d_logits},
                   {n->params, nz ? nz + L : nullptr, sin[0] + (size_t)B * n->in1, d_logits + lsz},
                   {n->target, nz ? nz + 2 * L : nullptr, sin[1], d_logits + 2 * lsz}};
  int rc = rb_heads(n, hj, 3, B, jh_s(stream), raw);
  if (rc) return rc;
  n->last_noise = d_noise;
  n->raw_heads = raw;
  n->dx_ready = 0;
  return JH_OK;
}
JH_EXPORT int jh_rbnet_learn_heads(jh_rbnet* n, int32_t B, const float* d_noise, float* d_logits, jh_stream stream) {
  return rb_learn_heads(n, B, d_noise, d_logits, false, stream);
}
// ... the same up to the advantage / value streams of the three forwards (xa, xv stay inside the network): jh_rbnet_c51_step forms the
// logits in the loss kernel (one launch less, and the gradient comes back already pulled through the combine: one more)
JH_EXPORT int jh_rbnet_learn_heads_raw(jh_rbnet* n, int32_t B, const float* d_noise, jh_stream stream) {
  return rb_learn_heads(n, B, d_noise, nullptr, true, stream);
}

// Rainbow.learn's loss step (rainbow.py:160-235) on the network's own stream outputs, after jh_rbnet_learn_trunk + jh_rbnet_learn_heads_raw:
//   launch 1  dueling combine of the three forwards (-> d_logits [3][B][A][K], bit-identical to jh_rbnet_learn_heads) + double-Q action +
//             n-step projection + KL + priorities KL^alpha + d(loss)/d(xa), d(loss)/d(xv)   (jh_c51_block_kernel<true>)
//   launch 2  batch statistics (d_stats, as jh_c51_loss) + the new priorities written into the sum tree's leaves at d_tree_idx
//   launch 3  the tree climb (jh_per.hip) -- per == null: launch 2 is statistics only, no launch 3
// Six launches before (heads' combine, loss, statistics, leaf write-back, climb, gradient through the combine).  jh_rbnet_backward(n, null)
// continues from the gradient left in the network.
JH_EXPORT int jh_rbnet_c51_step(jh_rbnet* n, jh_per* per, int32_t B, int32_t n_step, int32_t flags, const float* d_action, const float* d_reward,
                                const float* d_done, const float* d_weights, const int64_t* d_tree_idx, float v_min, float v_max, float gamma,
                                float alpha, float* d_logits, float* d_prio, float* d_kl, float* d_stats, jh_stream stream) {
  JH_ARG(n && d_action && d_reward && d_done && d_logits && d_prio);
  JH_ARG(n->dueling && n->K > 1 && n->K <= 256);
  JH_ARG(B > 0 && B <= 1024 && B == n->last_B && n_step >= 0);
  JH_ARG((flags & JH_C51_DOUBLE) != 0);
  JH_ARG(!(flags & JH_C51_PER) || d_weights);
  JH_ARG(!per || d_tree_idx);
  if (!n->raw_heads) return jh_fail(JH_ERR_STATE, "jh_rbnet_c51_step without a preceding jh_rbnet_learn_heads_raw");
  hipStream_t st = jh_s(stream);
  const size_t B_ = (size_t)n->maxB, lsz = (size_t)B * n->NA;
  C51Args a{};
  a.B = B; a.A = n->A; a.K = n->K; a.n = n_step > 0 ? n_step : 1; a.flags = flags;
  a.action = d_action; a.reward = d_reward; a.done = d_done; a.weights = d_weights; a.v_min = v_min; a.v_max = v_max; a.gamma = gamma;
  a.alpha = alpha; a.prio = d_prio; a.kl = d_kl; a.stats = d_stats;
  C51Duel d{};
  for (int j = 0; j < 3; ++j) { d.xa[j] = n->xa + j * B_ * n->NA4; d.xv[j] = n->xv + j * B_ * n->K4; d.out[j] = d_logits + j * lsz; }
  d.dxa = n->dxa; d.dxv = n->dxv; d.ld_a = n->NA4; d.ld_v = n->K4;
  PerDeltaArgs pa{};
  int rc;
  if (per) {
    rc = jh_per_delta_args(per, B, d_tree_idx, d_prio, JH_F32, &pa);
    if (rc) return rc;
  }
  rc = jh_c51_run(n->ctx, a, &d, per ? &pa : nullptr, st);
  if (rc) return rc;
  n->raw_heads = 0;
  n->dx_ready = 1;
  return per ? jh_per_climb(per, B, d_tree_idx, st) : JH_OK;
}

JH_EXPORT int jh_rbnet_learn_forward(jh_rbnet* n, const void* d_x, int32_t x_dtype, int32_t B, const float* d_noise, float* d_logits, jh_stream stream) {
  JH_ARG(n && d_x && d_logits);
  JH_ARG(d_noise || !n->noisy);
  int rc = jh_rbnet_learn_trunk(n, d_x, x_dtype, B, stream);
  if (rc) return rc;
  return jh_rbnet_learn_heads(n, B, d_noise, d_logits, stream);
}

static int rb_noisy_grad(jh_rbnet* n, hipStream_t st) {
  int64_t blocks = (n->n_noisy + 255) / 256;
  if (blocks > 1024) blocks = 1024;  // (4096 blocks -- one element per thread -- measured 2.6 us SLOWER at Rainbow's 655 k noisy weights x 3 sets, round 5)
  JH_LAUNCH(jh_rb_noisy_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n->nd, n->last_noise, n->grads, n->n_noisy);
  JH_LAUNCH_CHECK();
  return JH_OK;
}
static int rb_parts_sum(jh_rbnet* n, int parts, hipStream_t st) {
  const int n_w = 32 * n->Cin * 64, n_all = n_w + 32;
  float* G = n->grads;
  JH_LAUNCH(jh_rb_parts_sum_kernel, dim3((n_all + 31) / 32), dim3(256), 0, st, n->c1_part, G + n->seg_off[SEG_W1], G + n->seg_off[SEG_B1], n_w, n_all, parts);
  JH_LAUNCH_CHECK();
  return JH_OK;
}
// what a deferred backward left undone (the gradient bucket is complete afterwards)
static int rb_flush_grads(jh_rbnet* n, hipStream_t st) {
  int rc = JH_OK;
  if (n->pend_parts) rc = rb_parts_sum(n, n->pend_parts, st);
  n->pend_parts = 0;
  return rc;
}

// Backward of logits[0] of the last jh_rbnet_learn_forward: d_g = d(loss)/d(logits) [B][A][K]; fills the
// gradient bucket (same layout as the parameters).  Effective weights of noise set 0 and all online
// activations of the `state` rows are still in place.
static int rb_backward(jh_rbnet* n, const float* d_g, bool defer, jh_stream stream) {
  JH_ARG(n != nullptr);
  if (!n->last_x) return jh_fail(JH_ERR_STATE, "jh_rbnet_backward without a preceding jh_rbnet_learn_forward");
  if (!d_g && !n->dx_ready) return jh_fail(JH_ERR_STATE, "jh_rbnet_backward(null gradient) without a preceding jh_rbnet_c51_step");
  n->pend_parts = 0;
  hipStream_t st = jh_s(stream);
  const int B = n->last_B, H = n->hidden, NA = n->NA, K = n->K, F = n->F, in1 = n->in1;
  const StreamW w0 = stream_w(n, n->params, 0);  // noise set 0 / the online parameters
  const float* hav0 = n->hav;
  const float* sin0 = n->has_l ? n->h[0] : n->feat[0];  // input of the streams (rows of `state`)
  float* dsin = n->has_l ? n->dh : n->dfeat;
  float* G = n->grads;
  const float* dxa = d_g;
  int ld_dxa = NA;
  if (n->dueling) {
    if (d_g) {
      JH_LAUNCH(jh_rb_duel_bwd_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, d_g, B, n->A, K, n->dxa, n->NA4, n->dxv, n->K4);
      JH_LAUNCH_CHECK();
    }  // else: jh_rbnet_c51_step left the gradient in dxa / dxv
    dxa = n->dxa;
    ld_dxa = n->NA4;
  }
  n->dx_ready = 0;
  TGemm g[4];
  int ng = 0, rc;
  // last stream layer(s): weight gradients (+ bias gradients as row sums) and data gradients (+ relu')
  const float* in_a = n->has_av1 ? hav0 : sin0;
  const int ld_in = n->has_av1 ? 2 * H : H;
  float* d_in = n->has_av1 ? n->dhav : dsin;
  // factorised NoisyNet layers: d(sig) = d(mu) * f(e_in) f(e_out)^T and d(sig_b) = d(mu_b) * f(e_out) leave the weight-gradient GEMMs' epilogues
  // beside d(mu), d(mu_b) (jh_tgemm.h: C2 / rowsum2).  Noise set 0 = [ei_a1 H][ej_a1 H][ei_v1 H][ej_v1 H][ei_a2 H][ej_a2 NA][ei_v2 H][ej_v2 K].
  const bool sig_epi = n->noisy && !n->noise_independent;
  const float* e0 = n->last_noise;
  auto with_sigma = [&](TGemm& t, int seg_sig, int seg_sigb, const float* ej, const float* ei, int split, const float* ej2, const float* ei2) {
    if (!sig_epi) return;
    t.C2 = G + n->seg_off[seg_sig]; t.rowsum2 = G + n->seg_off[seg_sigb];
    t.nz_m = ej; t.nz_n = ei; t.nz_split = split; t.nz_m2 = ej2; t.nz_n2 = ei2;
  };
  g[ng++] = mk_gemm(NA, H, B, op_dense(OP_XCONT, dxa, ld_dxa), op_dense(OP_XCONT, in_a, ld_in), G + n->seg_off[SEG_MU_A2], H, TEPI_NONE, nullptr, nullptr, 0,
                    G + n->seg_off[SEG_MUB_A2]);
  with_sigma(g[ng - 1], SEG_SIG_A2, SEG_SIGB_A2, e0 + 5 * H, e0 + 4 * H, NA, nullptr, nullptr);
  g[ng++] = mk_gemm(B, H, NA, op_dense(OP_KCONT, dxa, ld_dxa), op_dense(OP_XCONT, w0.a2, H), d_in, ld_in, TEPI_MASK, nullptr, in_a, ld_in);
  if (n->dueling) {
    g[ng++] = mk_gemm(K, H, B, op_dense(OP_XCONT, n->dxv, n->K4), op_dense(OP_XCONT, in_a + H, ld_in), G + n->seg_off[SEG_MU_V2], H, TEPI_NONE, nullptr, nullptr, 0,
                      G + n->seg_off[SEG_MUB_V2]);
    with_sigma(g[ng - 1], SEG_SIG_V2, SEG_SIGB_V2, e0 + 6 * H + NA, e0 + 5 * H + NA, K, nullptr, nullptr);
    g[ng++] = mk_gemm(B, H, K, op_dense(OP_KCONT, n->dxv, n->K4), op_dense(OP_XCONT, w0.v2, H), d_in + H, ld_in, TEPI_MASK, nullptr, in_a + H, ld_in);
  }
  rc = launch_tgemm(n, "jh_tgemm_stream2_bwd", g, ng, st);
  if (rc) return rc;
  if (n->has_av1) {  // first stream layer (a1 | v1 stacked)
    g[0] = mk_gemm(2 * H, in1, B, op_dense(OP_XCONT, n->dhav, 2 * H), op_dense(OP_XCONT, sin0, in1), G + n->seg_off[SEG_MU_AV1], in1, TEPI_NONE, nullptr, nullptr, 0,
                   G + n->seg_off[SEG_MUB_AV1]);
    // rows m < H are a1 (e_out at H + m, e_in at k), rows m >= H are v1 (e_out at 3H + (m - H) = 2H + m, e_in at 2H + k)
    with_sigma(g[0], SEG_SIG_AV1, SEG_SIGB_AV1, e0 + H, e0, H, e0 + 2 * H, e0 + 2 * H);
    g[1] = mk_gemm(B, in1, 2 * H, op_dense(OP_KCONT, n->dhav, 2 * H), op_dense(OP_XCONT, w0.av1, in1), dsin, in1, TEPI_MASK, nullptr, sin0, in1);
    rc = launch_tgemm(n, "jh_tgemm_stream1_bwd", g, 2, st);
    if (rc) return rc;
  }
  if (n->noisy && !sig_epi) {  // independent noise (utils.py:73-76): eps is a matrix of its own, the elementwise kernel forms d(sig)
    rc = rb_noisy_grad(n, st);
    if (rc) return rc;
  }
  if (n->has_l) {  // l: F -> H
    g[0] = mk_gemm(H, F, B, op_dense(OP_XCONT, n->dh, H), op_dense(OP_XCONT, n->feat[0], F), G + n->seg_off[SEG_WL], F, TEPI_NONE, nullptr, nullptr, 0,
                   G + n->seg_off[SEG_BL]);
    g[1] = mk_gemm(B, F, H, op_dense(OP_KCONT, n->dh, H), op_dense(OP_XCONT, n->params + n->seg_off[SEG_WL], F), n->dfeat, F, TEPI_MASK, nullptr, n->feat[0], F);
    rc = launch_tgemm(n, "jh_tgemm_fc_bwd", g, 2, st);
    if (rc) return rc;
  }
  if (!n->cnn) {
    g[0] = mk_gemm(H, n->Cin, B, op_dense(OP_XCONT, n->dfeat, H), op_dense(OP_XCONT, (const float*)n->last_x, n->Cin), G + n->seg_off[SEG_W1], n->Cin, TEPI_NONE,
                   nullptr, nullptr, 0, G + n->seg_off[SEG_B1]);
    return launch_tgemm(n, "jh_tgemm_head_bwd", g, 1, st);
  }
  // conv3: d(feat) is d(act3) in NHWC [B*P3][64]
  g[0] = mk_gemm(64, 576, B * n->P3, op_dense(OP_XCONT, n->dfeat, 64), op_conv(OP_NHWC_X, n->act2[0], 0, n->c3), G + n->seg_off[SEG_W3], 576, TEPI_NONE, nullptr,
                 nullptr, 0, G + n->seg_off[SEG_B3]);
  g[1] = mk_gemm(B * n->P3, 576, 64, op_dense(OP_KCONT, n->dfeat, 64), op_dense(OP_XCONT, n->params + n->seg_off[SEG_W3], 576), n->dcol, 576, TEPI_NONE);
  rc = launch_tgemm(n, "jh_tgemm_conv3_bwd", g, 2, st);
  if (rc) return rc;
  {
    const int n_pix = B * n->P2;
    JH_LAUNCH((jh_rb_col2im_kernel<3, 3, 1>), dim3((unsigned)(((int64_t)n_pix * 16 + 255) / 256)), dim3(256), 0, st, n->dcol, n->act2[0], n->dact2, n_pix, 64,
              n->c3.H, n->c3.W, n->c3.OH, n->c3.OW);
    JH_LAUNCH_CHECK();
  }
  g[0] = mk_gemm(64, 512, B * n->P2, op_dense(OP_XCONT, n->dact2, 64), op_conv(OP_NHWC_X, n->act1[0], 0, n->c2), G + n->seg_off[SEG_W2], 512, TEPI_NONE, nullptr,
                 nullptr, 0, G + n->seg_off[SEG_B2]);
  g[1] = mk_gemm(B * n->P2, 512, 64, op_dense(OP_KCONT, n->dact2, 64), op_dense(OP_XCONT, n->params + n->seg_off[SEG_W2], 512), n->dcol, 512, TEPI_NONE);
  rc = launch_tgemm(n, "jh_tgemm_conv2_bwd", g, 2, st);
  if (rc) return rc;
  {
    const int n_pix = B * n->P1;
    JH_LAUNCH((jh_rb_col2im_kernel<4, 4, 2>), dim3((unsigned)(((int64_t)n_pix * 8 + 255) / 256)), dim3(256), 0, st, n->dcol, n->act1[0], n->dact1, n_pix, 32,
              n->c2.H, n->c2.W, n->c2.OH, n->c2.OW);
    JH_LAUNCH_CHECK();
  }
  // uint8 frames in the head's own geometry: the dedicated K-parallel kernel (above); anything else is a GEMM like the other layers
  static const bool kC1Own = !(getenv("JH_RB_CONV1_WGRAD") && atoi(getenv("JH_RB_CONV1_WGRAD")) == 0);
  const ConvGeom& c1 = n->c1;
  const int64_t c1_groups = (int64_t)B * ((n->P1 + 3) / 4);
  if (kC1Own && n->last_x_u8 && c1.KH == 8 && c1.KW == 8 && c1.S == 4 && c1.W % 4 == 0 && n->Cin % 4 == 0 && n->P1 % 4 == 0 && c1.OW >= 4 && c1_groups < (1 << 20)) {
    constexpr int UN = 5;
    const int gpi = (n->P1 + 3) / 4, total = (int)c1_groups;
    int per = (total + kC1Parts - 1) / kC1Parts;
    per = ((per + 4 * UN - 1) / (4 * UN)) * (4 * UN);  // four K slices per workgroup, whole rounds of UN groups each
    const int parts = (total + per - 1) / per;
    const int n_w = 32 * n->Cin * 64, n_all = n_w + 32;
    JH_LAUNCH_IDEM("jh_rb_conv1_wgrad_kernel", 2.0 * 32 * n->Cin * 64 * (double)B * n->P1, jh_rb_conv1_wgrad_kernel<UN>, dim3(parts), dim3(1024), 0, st, (const uint8_t*)n->last_x, n->dact1, n->c1_part, n->Cin, c1.H, c1.W, c1.OW, n->P1,
              1.0f / (float)c1.OW, gpi, total, per);
    JH_LAUNCH_CHECK();
    if (defer && n->seg_off[SEG_W1] == 0 && n->seg_off[SEG_B1] == n_w && n->seg_off[SEG_W2] == n_all) {
      n->pend_parts = parts;
      return JH_OK;
    }
    return rb_parts_sum(n, parts, st);
  }
  g[0] = mk_gemm(32, n->Cin * 64, B * n->P1, op_dense(OP_XCONT, n->dact1, 32), op_conv(OP_NCHW_X, n->last_x, n->last_x_u8, n->c1), G + n->seg_off[SEG_W1],
                 n->Cin * 64, TEPI_NONE, nullptr, nullptr, 0, G + n->seg_off[SEG_B1]);
  return launch_tgemm(n, "jh_tgemm_conv1_bwd", g, 1, st);
}
JH_EXPORT int jh_rbnet_backward(jh_rbnet* n, const float* d_g, jh_stream stream) { return rb_backward(n, d_g, false, stream); }
// The same, but one tail stays undone -- the sum of conv1's weight-gradient partials -- for jh_rbnet_optim_step to do inside the optimizer's
// launch (which it does when no clipping is asked for; otherwise, and in jh_rbnet_flush_grads, it runs as the launch it was).  Whoever reads
// the gradient bucket before the optimizer step (a data-parallel all-reduce, a test) calls jh_rbnet_flush_grads first or uses jh_rbnet_backward.
JH_EXPORT int jh_rbnet_backward_deferred(jh_rbnet* n, const float* d_g, jh_stream stream) { return rb_backward(n, d_g, true, stream); }
JH_EXPORT int jh_rbnet_flush_grads(jh_rbnet* n, jh_stream stream) {
  JH_ARG(n != nullptr);
  return rb_flush_grads(n, jh_s(stream));
}

JH_EXPORT int jh_rbnet_optim_step(jh_rbnet* n, int32_t optimizer, float max_norm, jh_stream stream) {
  JH_ARG(n != nullptr);
  JH_ARG(optimizer == 0 || optimizer == 1);
  hipStream_t st = jh_s(stream);
  static const bool kFuse = !(getenv("JH_RB_OPTIM_FUSE") && atoi(getenv("JH_RB_OPTIM_FUSE")) == 0);
  OptFuse f{};
  if (max_norm <= 0.f && kFuse && n->pend_parts) {  // (the norm needs the finished gradient)
    f.n_c1 = 32 * n->Cin * 64 + 32; f.parts = n->pend_parts; f.c1_part = n->c1_part; f.extra_wgs = (f.n_c1 + 31) / 32;
    n->pend_parts = 0;
  }
  int rc = rb_flush_grads(n, st);  // whatever is still pending runs as the launch it was
  if (rc) return rc;
  if (max_norm > 0.f) {
    JH_LAUNCH(jh_rb_gradnorm_kernel, dim3(256), dim3(256), 0, st, n->n_params, n->grads, n->norm_partial);
    JH_LAUNCH_CHECK();
  }
  static const unsigned kOptGrid = getenv("JH_RB_OPTIM_GRID") ? (unsigned)atoi(getenv("JH_RB_OPTIM_GRID")) : 512u;
  if (optimizer == 0) {
    JH_LAUNCH(jh_rb_optim_kernel<0>, dim3(kOptGrid + f.extra_wgs), dim3(256), 0, st, n->n_params, n->params, n->grads, n->m, n->v, n->hyper, n->ticket, n->norm_partial, 256, max_norm, f);
  } else {
    JH_LAUNCH(jh_rb_optim_kernel<1>, dim3(kOptGrid + f.extra_wgs), dim3(256), 0, st, n->n_params, n->params, n->grads, n->m, n->v, n->hyper, n->ticket, n->norm_partial, 256, max_norm, f);
  }
  JH_LAUNCH_CHECK();
  return JH_OK;
}

JH_EXPORT int jh_rbnet_adam_step(jh_rbnet* n, jh_stream stream) { return jh_rbnet_optim_step(n, 0, 0.f, stream); }
