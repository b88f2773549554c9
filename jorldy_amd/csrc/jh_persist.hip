// Persistent acting kernel for the native sync collector ("on-GPU batched
// acting"): ONE launch serves all T timesteps of a rollout.
//
// Per-timestep acting is a latency problem, not a throughput problem: W = 8 rows through a
// 4-512-512-3 MLP is ~4 MFLOP, but a launch costs ~5 us of dispatch latency plus ~4 us of fixed
// kernel overhead plus PCIe round trips, 128 times per iteration.  This kernel removes all of it:
//   * grid = H/16 workgroups, each owns 16 hidden-2 columns and keeps ITS slice of every weight
//     (W2 rows, W1, biases, head-weight columns) resident in LDS for the whole rollout: no global
//     load on the per-step critical path;
//   * the host publishes the observations of step t as 8-byte {tag = t, value} granules in
//     device-mapped pinned memory ("the data IS the flag", CDNA guide G16/R2): one PCIe read round
//     trip both detects the step and fetches the data;
//   * every workgroup computes h1 (VALU) and its h2 tile (fp32 MFMA, in-workgroup split-K), reduces
//     the tile against the head weights and writes its partial head outputs as 16-byte row granules
//     {out0, out1, out2, tag} straight into pinned host memory -- W consecutive granules per
//     workgroup in one store instruction, fire and forget (no fence, no acknowledgement wait); the
//     host sums the 32 partials per output, samples, steps the envs;
//   * no inter-workgroup communication on the device at all -> nothing to deadlock on; every poll
//     loop is bounded and a timeout makes all workgroups exit (the host then falls back to the
//     one-launch-per-step path).
#include "jh_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PersistArgs {
  int W, S, H, n_out, T;
  const float *W1, *b1, *W2, *b2;
  const float* wh[8];
  const float* hbias[8];
  const unsigned long long* obs_gran;  // pinned: [W*S] granules {tag << 32 | float bits}
  float4* part;                        // pinned: [tiles][16] row granules {out0, out1, out2, tag bits}
  unsigned* tile_flag;                 // pinned: [tiles] (unused by the granule protocol, kept for debugging)
  unsigned* abort_flag;                // pinned: set by the kernel on timeout / by the host to stop early
  unsigned seq0;                       // tag of the first step (tags are seq0+1 .. seq0+T)
  long max_polls;
  unsigned long long* dbg;             // optional [T][8]: per-step timestamps of workgroup 0 (diagnostics)
  int poll_sleep;                      // back-off between polls: 0 none, 1 s_sleep 1, 2 s_sleep 8, 3 s_sleep 32
  int groups;                          // 1, or 2: the env rows are served as two half-batches per timestep so that the host
                                       // (sampling, env.step, next observations) of one half overlaps the GPU work of the other
};

__global__ void __launch_bounds__(256) jh_act_persist_kernel(PersistArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = p.H, S = p.S, ldh = H + 4;
  float* w2s = smem;                 // [16][H+4]  this tile's rows of W2 (B operand, k contiguous)
  float* h1s = w2s + 16 * ldh;       // [16][H+4]  layer-1 activations of the current step
  float* w1s = h1s + 16 * ldh;       // [H][S]
  float* b1s = w1s + H * S;          // [H]
  float* xs = b1s + H;               // [16][S]
  float* whs = xs + 16 * S;          // [8][16] head-weight columns of this tile
  float* misc = whs + 8 * 16;        // [16] b2 slice, [8] head biases
  float* outs_s = misc + 32;         // [16][4] staging of this tile's partial head outputs
  float* s_acc = outs_s + 64;        // [4][64][4] split-K combine
  __shared__ int s_go;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tile = blockIdx.x, n0 = tile * 16;
  const int r = lane & 15, kq = lane >> 4;
  // ---- one-time: weights into LDS
  for (int i = threadIdx.x; i < 16 * H; i += 256) {
    const int rr = i / H, k = i - rr * H;
    w2s[rr * ldh + k] = p.W2[(size_t)(n0 + rr) * H + k];
  }
  for (int i = threadIdx.x; i < H * S; i += 256) w1s[i] = p.W1[i];
  for (int i = threadIdx.x; i < H; i += 256) b1s[i] = p.b1[i];
  if (threadIdx.x < 16) misc[threadIdx.x] = p.b2[n0 + threadIdx.x];
  if (threadIdx.x < 8) misc[16 + threadIdx.x] = threadIdx.x < p.n_out ? *p.hbias[threadIdx.x] : 0.f;
  for (int i = threadIdx.x; i < 8 * 16; i += 256) {
    const int o = i >> 4, c = i & 15;
    whs[i] = o < p.n_out ? p.wh[o][n0 + c] : 0.f;
  }
  __syncthreads();
  const int kper = H / 4;  // H % 64 == 0 is checked on the host
  const int kbeg = wid * kper;

  const int rows_per = p.W / p.groups;
  for (int tg = 0; tg < p.T * p.groups; ++tg) {
    const int t = tg / p.groups + 1, grp = tg - (t - 1) * p.groups;
    const int row0 = grp * rows_per, row1 = row0 + rows_per;  // the env rows of this half-batch
    const unsigned tag = p.seq0 + (unsigned)t;
    // ---- wait for the host's observations of step t (granule sweep, bounded)
    if (wid == 0) {
      bool ok = false;
      for (long spin = 0; spin < p.max_polls; ++spin) {
        bool mine = true;
        for (int i = row0 * S + lane; i < row1 * S; i += 64) {
          const unsigned long long gq = __hip_atomic_load(p.obs_gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if ((unsigned)(gq >> 32) == tag) xs[i] = __uint_as_float((unsigned)gq);
          else mine = false;
        }
        if (__all(mine)) { ok = true; break; }
        if ((spin & 63) == 63 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        if (p.poll_sleep == 1) __builtin_amdgcn_s_sleep(1);
        else if (p.poll_sleep == 2) __builtin_amdgcn_s_sleep(8);
        else if (p.poll_sleep == 3) __builtin_amdgcn_s_sleep(32);
      }
      if (lane == 0) s_go = ok ? 1 : 0;
    }
    __syncthreads();
    if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) { p.dbg[(t - 1) * 8 + 0] = wall_clock64(); p.dbg[(t - 1) * 8 + 1] = __builtin_readcyclecounter(); }
    if (!s_go) {  // timeout or host abort: tell the host and leave (all workgroups decide alike or time out too)
      if (threadIdx.x == 0) __hip_atomic_store(p.abort_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    // ---- layer 1 for the (<=16) env rows ON THE MFMA: h1[16 x H] = x[16 x S] * W1^T[S x H].  K = S is
    // tiny (one 16x16x4 MFMA per 16 hidden units when S <= 4), and this form needs one LDS read per
    // tile instead of re-reading every observation for every hidden unit (the VALU version spent
    // 3.2 us per step issuing ~170 LDS instructions per wave; this one ~0.2 us).
    {
      const int u_per_wave = H / 4;  // hidden units of this wave (H % 64 == 0)
      for (int u0 = wid * u_per_wave; u0 < (wid + 1) * u_per_wave; u0 += 16) {  // independent tiles: overlap their LDS/MFMA latencies
        f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < S; q0 += 4) {
          const int q = q0 + kq;
          const float av = (r < p.W && q < S) ? xs[r * S + q] : 0.f;       // A[row r][k = q]
          const float bv = q < S ? w1s[(u0 + r) * S + q] : 0.f;            // B[k = q][unit u0 + r]
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c1, 0, 0, 0);
        }
        const float bb = b1s[u0 + r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // C: col = lane & 15 -> unit u0 + r, row = kq * 4 + i -> env row
          const float v = c1[i] + bb;
          h1s[(kq * 4 + i) * ldh + u0 + r] = v > 0.f ? v : 0.f;
        }
      }
    }
    __syncthreads();
    if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) { p.dbg[(t - 1) * 8 + 2] = wall_clock64(); p.dbg[(t - 1) * 8 + 3] = __builtin_readcyclecounter(); }
    // ---- h2 tile = h1 (16 x H) * W2_tile^T (H x 16): fp32 MFMA, this wave's K quarter
    // two independent accumulators (the 16x16x4 fp32 MFMA has a 40-cycle dependent latency vs a
    // 32-cycle issue interval) and an unrolled body so the ds_read_b128 of later chunks are in flight
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kbeg + kper; k0 += 32) {
      const int kb = k0 + 4 * kq;
      const float4 av = *reinterpret_cast<const float4*>(h1s + r * ldh + kb);
      const float4 bv = *reinterpret_cast<const float4*>(w2s + r * ldh + kb);
      const float4 av2 = *reinterpret_cast<const float4*>(h1s + r * ldh + kb + 16);
      const float4 bv2 = *reinterpret_cast<const float4*>(w2s + r * ldh + kb + 16);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av2.x, bv2.x, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av2.y, bv2.y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av2.z, bv2.z, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av2.w, bv2.w, acc2, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += acc2[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s_acc[(wid * 64 + lane) * 4 + i] = acc[i];
    __syncthreads();
    if (wid == 0) {
      // C/D fragment: col = lane & 15 (hidden-2 column n0 + r), row = kq * 4 + i (env row)
      // h2 tile -> LDS (transposed into A-operand order), then the head partials
      //   part[row][o] = sum_col h2[row][col] * Wh[o][n0 + col]
      // are ONE more 16x16x16 product on the MFMA (4 steps) -- no cross-lane shuffles at all.
      float* h2s = s_acc;  // [16 rows][17]: safe to overwrite, every wave's partials were consumed above
      float hv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = ((s_acc[(0 * 64 + lane) * 4 + i] + s_acc[(1 * 64 + lane) * 4 + i]) + s_acc[(2 * 64 + lane) * 4 + i]) +
                  s_acc[(3 * 64 + lane) * 4 + i];
        v += misc[r];
        hv[i] = v > 0.f ? v : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
      for (int i = 0; i < 4; ++i) h2s[(kq * 4 + i) * 17 + r] = hv[i];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      f32x4 ph = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float av = h2s[r * 17 + 4 * c + kq];                           // A[row r][k = col 4c+kq]
        const float bv = r < p.n_out ? whs[r * 16 + 4 * c + kq] : 0.f;       // B[k = col][n = output r]
        ph = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, ph, 0, 0, 0);
      }
      // ph: col = lane & 15 -> output o = r, row = kq * 4 + i -> env row
      if (r < p.n_out) {
#pragma unroll
        for (int i = 0; i < 4; ++i) outs_s[(kq * 4 + i) * 4 + r] = ph[i] + (tile == 0 ? misc[16 + r] : 0.f);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      // One 16-byte row granule {out0, out1, out2, tag}: the partial IS its own flag.  Lanes 0..W-1
      // store W consecutive granules with ONE instruction (W*16 contiguous bytes -> one or two PCIe
      // write TLPs per tile instead of dozens of 4-byte ones); fire and forget: no release fence, no
      // acknowledgement wait on the per-step critical path.
      if (lane >= row0 && lane < row1) {
        const float4 o4 = *reinterpret_cast<const float4*>(outs_s + lane * 4);
        const f32x4 gq = (f32x4){o4.x, o4.y, o4.z, __uint_as_float(tag)};
        float4* dst = p.part + (size_t)tile * 16 + lane;
        // write-through system-scope 16-byte store (a plain store lingers in L2 for milliseconds); hipcc
        // does not track asm stores: nothing here waits for it on purpose, the s_nop keeps the data
        // registers intact until the store has read them (CDNA guide §5.7)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(gq) : "memory");
      }
    }
    __syncthreads();  // s_acc / h1s are reused by the next step
    if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) { p.dbg[(t - 1) * 8 + 4] = wall_clock64(); p.dbg[(t - 1) * 8 + 5] = __builtin_readcyclecounter(); }
  }
}

// ---------------------------------------------------------------------------------- host side
struct jh_persist {
  jh_pponet* net = nullptr;
  unsigned long long* gran_h = nullptr;
  unsigned long long* gran_d = nullptr;
  float4 *part_h = nullptr, *part_d = nullptr;
  unsigned *flag_h = nullptr, *flag_d = nullptr;  // [tiles] + abort word at [tiles]
  unsigned seq = 0;
  int groups = 1;
  int tiles = 0;
  size_t lds = 0;
  unsigned long long *dbg_h = nullptr, *dbg_d = nullptr;
};

int jh_persist_create(jh_pponet* n, jh_persist** out) {
  JH_ARG(n && out);
  JH_ARG(!n->cont && n->H % 128 == 0 && n->A + 1 <= 3 && (16 * n->S) % 4 == 0);
  jh_persist* p = new jh_persist();
  p->net = n;
  p->tiles = n->H / 16;
  const int H = n->H, S = n->S;
  p->lds = sizeof(float) * ((size_t)2 * 16 * (H + 4) + (size_t)H * S + H + 16 * (size_t)S + 8 * 16 + 32 + 64 + 4 * 64 * 4);
  if (p->lds > 160 * 1024) {
    delete p;
    return jh_fail(JH_ERR_ARG, "persistent acting needs %zu B of LDS (> 160 KiB) for H=%d S=%d", p->lds, H, S);
  }
  JH_HIP(hipFuncSetAttribute((const void*)jh_act_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds));
  JH_HIP(hipHostMalloc((void**)&p->gran_h, sizeof(unsigned long long) * 16 * (size_t)S, hipHostMallocMapped));
  JH_HIP(hipHostGetDevicePointer((void**)&p->gran_d, p->gran_h, 0));
  JH_HIP(hipHostMalloc((void**)&p->part_h, sizeof(float4) * 16 * (size_t)p->tiles, hipHostMallocMapped));
  memset(p->part_h, 0, sizeof(float4) * 16 * (size_t)p->tiles);
  JH_HIP(hipHostGetDevicePointer((void**)&p->part_d, p->part_h, 0));
  JH_HIP(hipHostMalloc((void**)&p->flag_h, sizeof(unsigned) * (size_t)(p->tiles + 16), hipHostMallocMapped));
  JH_HIP(hipHostGetDevicePointer((void**)&p->flag_d, p->flag_h, 0));
  memset(p->gran_h, 0, sizeof(unsigned long long) * 16 * (size_t)S);
  memset(p->flag_h, 0, sizeof(unsigned) * (size_t)(p->tiles + 16));
  if (getenv("JH_PERSIST_DEBUG")) {
    // timestamps go to DEVICE memory (a store to host memory would stall the wave at the next barrier
    // until the PCIe write is acknowledged and distort the measurement)
    p->dbg_h = (unsigned long long*)malloc(sizeof(unsigned long long) * 8 * 4096);
    JH_HIP(hipMalloc((void**)&p->dbg_d, sizeof(unsigned long long) * 8 * 4096));
  }
  p->seq = 1000;  // tags never collide with the zero-initialised granules
  *out = p;
  return JH_OK;
}

void jh_persist_destroy(jh_persist* p) {
  if (!p) return;
  (void)hipHostFree(p->gran_h);
  (void)hipHostFree(p->part_h);
  (void)hipHostFree(p->flag_h);
  delete p;
}

// Launch the persistent kernel for T steps of W <= 16 envs.
int jh_persist_begin(jh_persist* p, int W, int T, int groups, hipStream_t st) {
  jh_pponet* n = p->net;
  JH_ARG(W > 0 && W <= 16 && T > 0);
  JH_ARG(groups == 1 || (groups == 2 && W % 2 == 0));
  p->groups = groups;
  PersistArgs a{};
  a.W = W; a.S = n->S; a.H = n->H; a.T = T;
  a.W1 = n->params + n->o_w1; a.b1 = n->params + n->o_b1; a.W2 = n->params + n->o_w2; a.b2 = n->params + n->o_b2;
  int o = 0;
  for (int k = 0; k < n->A; ++k, ++o) { a.wh[o] = n->params + n->o_wh0 + (int64_t)k * n->H; a.hbias[o] = n->params + n->o_bh0 + k; }
  a.wh[o] = n->params + n->o_wv; a.hbias[o] = n->params + n->o_bv; ++o;
  a.n_out = o;
  a.obs_gran = p->gran_d; a.part = p->part_d; a.tile_flag = p->flag_d; a.abort_flag = p->flag_d + p->tiles;
  a.seq0 = p->seq;
  a.dbg = p->dbg_d;
  a.groups = groups;
  a.poll_sleep = 2;
  if (const char* e = getenv("JH_PERSIST_SLEEP")) a.poll_sleep = atoi(e);
  a.max_polls = 400000;  // x (~0.5 us per poll) = ~0.2 s without observations -> give up
  p->flag_h[p->tiles] = 0;
  JH_LAUNCH(jh_act_persist_kernel, dim3(p->tiles), dim3(256), p->lds, st, a);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// Tag of the next timestep (all half-batches of a timestep share it).
unsigned jh_persist_next_tag(jh_persist* p) { return ++p->seq; }

// Publish the observations of env rows [r0, r1) for the timestep `tag` (h_obs is the full [W][S] array).
void jh_persist_publish(jh_persist* p, int r0, int r1, const float* h_obs, unsigned tag) {
  const int S = p->net->S;
  for (int i = r0 * S; i < r1 * S; ++i) {
    unsigned bits;
    memcpy(&bits, h_obs + i, 4);
    __atomic_store_n(p->gran_h + i, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELEASE);
  }
}

// Wait until every tile's partial granules of rows [r0, r1) carry `tag`, finish the heads on the host and sample.
// The sampling stream is keyed by (act_ctr, row): identical whichever way the rows are batched; the caller advances
// act_ctr once per timestep (jh_persist_end_step).  JH_ERR_STATE if the kernel gave up.
int jh_persist_collect(jh_persist* p, int r0, int r1, unsigned tag, int64_t* h_action, int training) {
  jh_pponet* n = p->net;
  const int A = n->A, n_out = A + 1;
  volatile unsigned* abort_w = p->flag_h + p->tiles;
  const volatile unsigned* part = reinterpret_cast<const volatile unsigned*>(p->part_h);  // [tiles][16][4 words]
  bool all = false;
  for (long spin = 0; spin < 40000000L && !all; ++spin) {
    all = true;
    for (int t = 0; t < p->tiles && all; ++t)
      for (int wq = r0; wq < r1; ++wq)
        if (part[((size_t)t * 16 + wq) * 4 + 3] != tag) { all = false; break; }
    if (!all) {
      if ((spin & 1023) == 1023 && *abort_w == 2u) break;  // the kernel timed out
      __builtin_ia32_pause();
    }
  }
  if (!all) return jh_fail(JH_ERR_STATE, "persistent acting kernel did not answer step tag %u", tag);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int wq = r0; wq < r1; ++wq) {
    float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < p->tiles; ++t) {
      for (int o = 0; o < n_out; ++o) {
        const unsigned bits = part[((size_t)t * 16 + wq) * 4 + o];
        float v;
        memcpy(&v, &bits, 4);
        z[o] += v;
      }
    }
    int act = 0;
    float mx = z[0];
    for (int k = 1; k < A; ++k)
      if (z[k] > mx) { mx = z[k]; act = k; }
    if (training) {
      float e[8], se = 0.f;
      for (int k = 0; k < A; ++k) { e[k] = expf(z[k] - mx); se += e[k]; }
      uint64_t x = n->act_seed * 0x100000001B3ull + n->act_ctr * 0x9E3779B97F4A7C15ull + (uint64_t)wq;
      x += 0x9E3779B97F4A7C15ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      x = x ^ (x >> 31);
      const float u = (float)((double)(x >> 11) * (1.0 / 9007199254740992.0)) * se;
      float c = 0.f;
      act = A - 1;
      for (int k = 0; k < A; ++k) {
        c += e[k];
        if (u < c) { act = k; break; }
      }
    }
    h_action[wq] = act;
  }
  return JH_OK;
}

void jh_persist_end_step(jh_persist* p) { p->net->act_ctr += 1; }

// One whole timestep, all rows at once (groups == 1).
int jh_persist_step(jh_persist* p, int W, const float* h_obs, int64_t* h_action, int training) {
  const unsigned tag = jh_persist_next_tag(p);
  jh_persist_publish(p, 0, W, h_obs, tag);
  int rc = jh_persist_collect(p, 0, W, tag, h_action, training);
  if (rc) return rc;
  jh_persist_end_step(p);
  return JH_OK;
}

// Diagnostics: print per-step phase durations of workgroup 0 for the last rollout.
void jh_persist_dump_debug(jh_persist* p, int T) {
  if (!p->dbg_h) return;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(p->dbg_h, p->dbg_d, sizeof(unsigned long long) * 8 * (size_t)T, hipMemcpyDeviceToHost);
  double a = 0, b = 0, c = 0, cyc = 0;
  for (int t = 1; t < T; ++t) {
    const unsigned long long* d = p->dbg_h + (size_t)t * 8;
    const unsigned long long* pr = p->dbg_h + (size_t)(t - 1) * 8;
    a += (double)(d[0] - pr[4]);   // wait for observations (incl. host work + PCIe), 100 MHz ticks
    b += (double)(d[2] - d[0]);    // layer 1
    c += (double)(d[4] - d[2]);    // MFMA + combine + heads + store
    cyc += (double)(d[5] - d[1]) / ((double)(d[4] - d[0]) + 1e-9);  // shader cycles per 10 ns tick
  }
  const double n = T - 1;
  fprintf(stderr, "[jh_persist] per step (wall_clock64 ticks = 10 ns): wait %.1f  layer1 %.1f  gemm+heads %.1f ; shader clock ~%.0f MHz\n",
          a / n, b / n, c / n, cyc / n * 100.0);
}

// Stop a running kernel early (error paths): it sees the word at its next poll and exits.
void jh_persist_abort(jh_persist* p) { __atomic_store_n(p->flag_h + p->tiles, 1u, __ATOMIC_RELEASE); }
