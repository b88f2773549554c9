// Persistent acting kernel for the native sync collector ("on-GPU batched
// acting"): ONE launch serves all T timesteps of a rollout.
//
// Per-timestep acting is a latency problem, not a throughput problem: W = 8 rows through a
// 4-512-512-3 MLP is ~4 MFLOP, but it happens 128 times per iteration between two PCIe crossings.
// Round 2 rebuilt the kernel around the two things that were left on the per-step critical path
// (round 1: 6-10 us waiting for the observations + 3.2 us of kernel per step):
//   * EVERY weight a lane needs lives in its REGISTERS for the whole rollout: its 4 k-values x NCH
//     chunks of the W2 column it owns, the W1 rows + biases of exactly those k.  A step is then:
//     read the env row's S observations from LDS, S x 4 x NCH FMAs generate the layer-1 fragment in
//     place (no layer-1 MFMA pass, no h1 round trip through LDS, one barrier less), 4 x NCH MFMAs,
//     in-workgroup split-K combine, head partials on one more MFMA pass.  LDS staging buffers are
//     double-buffered by step parity, which removes the end-of-step barrier as well;
//   * PIPELINED POLLING: a poll of the host's observation granules is a PCIe read round trip
//     (~1.6 us); polling "load, wait, compare, repeat" detects a publication 0.5-1.5 round trips
//     late, and the host can only continue when the SLOWEST of the 32 workgroups has answered.
//     Here wave 0 keeps D polls in flight as LDS-DMA loads (global_load_lds_dwordx4: no VGPR
//     destination, so polls that are still in flight when the step is detected just land in their
//     LDS ring slot later and are consumed -- stale -- at the start of the next step), paced
//     period ticks apart: a publication is seen ~ one round trip + period later by every workgroup.
//   * unchanged: the host publishes observations as 8-byte {tag, value} granules in device-mapped
//     pinned memory ("the data IS the flag", CDNA guide G16/R2); every workgroup answers with 16-byte
//     {out, out, out, tag} row granules written through to pinned host memory, fire and forget; no
//     inter-workgroup communication on the device; every poll loop is bounded and a timeout makes
//     all workgroups exit (the host then falls back to one launch per step).
// Round 4: up to 32 rows per step (RT = 2 row tiles of 16).  The rows need not be 32 ENVIRONMENTS: the native collector's
// lookahead form (jh_collect.hip) publishes, for W <= 10 discrete-action envs, each env's current state AND the two states its two
// actions lead to (the envs are host objects that can be forked), so that one PCIe round trip serves TWO timesteps: the host samples
// a_t from the root row, picks that child and samples a_{t+1} from the child's row, which is already there.  The second row tile
// is a second set of 32 workgroups (grid = column tiles x row tiles): a step's compute stays the one-tile 1.65 us.
// Heads: G = ceil(n_out / 3) granules per (tile, row), n_out = A logits (or mu / log_std of a continuous policy) + the value
// head (last output; handed to the learner by the collector's capture): discrete A <= 11, continuous A <= 5.
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include "jh_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PersistArgs {
  int W, S, H, n_out, T, G;
  const float *W1, *b1, *W2, *b2;
  const float* wh[12];
  const float* hbias[12];
  const unsigned long long* obs_gran;  // pinned: [W*S] granules {tag << 32 | float bits}
  float4* part;                        // pinned: [tiles][G][rows_ld] row granules {out, out, out, tag bits}
  int rows_ld;                         // 16 x row tiles
  unsigned* abort_flag;                // pinned: set by the kernel on timeout / by the host to stop early
  unsigned seq0;                       // tag of the first step (tags are seq0+1 .. seq0+T)
  long max_polls;
  unsigned long long* dbg;             // optional [T][8]: per-step timestamps of workgroup 0 (diagnostics)
  int period;                          // spacing of the polls in flight, wall_clock64 ticks (10 ns)
  unsigned long long* mbox;            // device memory [64] granules: relay of the observations (null: every workgroup polls the host)
  float4* dpart;                       // device memory [row tiles][G][16 rows][tiles] granules, or null: round 6, the partial heads are summed ON THE DEVICE
  int split;                           // round 6: the two row tiles are exchanged INDEPENDENTLY (W = 32): a workgroup waits for, polls and relays only ITS row tile's
                                       // observation granules, so the host can publish one half's step t + 1 while the other half's step t is still in flight
};

// One poll: lanes 0..n16-1 of the calling wave each fetch 16 bytes (two granules) of the observation area
// straight into LDS (lane-linear at lds_off).  No VGPR destination: hipcc neither tracks nor waits for it.
__device__ __forceinline__ void persist_poll_issue(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}

// SP: observation width padded to 4 / 8 (the W1 rows of a lane's k-values live in its registers) or 12 / 16 (they
// live in LDS: 4 x NCH x SP registers would not fit).
// PI: poll instructions per poll = ceil(W x S / 128): up to 128 PI observation granules, lane l owns granules l, 64 + l, ... (round 5:
// PI = 3 carries config.ppo.mujoco's 32 workers x 11 observations; PI = 1 is the code of rounds 2-4, instruction for instruction).
// EVERY poll instruction of a poll has at least one active lane (the host picks PI exactly), so that "the oldest poll has landed" stays
// a fixed count of outstanding loads.
template <int SP, int NCH, int D, int PI>
__global__ void __launch_bounds__(256, 1) jh_act_persist_kernel(PersistArgs p) {
  constexpr bool W1LDS = SP > 8;
  constexpr int NJ = 2 * PI;  // granules per lane
  __shared__ __attribute__((aligned(16))) unsigned long long s_ring[D][128 * PI];  // poll landing slots
  __shared__ __attribute__((aligned(16))) float s_w1[W1LDS ? 64 * NCH * SP : 4];  // [H][SP] (W1LDS)
  __shared__ __attribute__((aligned(16))) float s_x[2][16][SP];
  __shared__ __attribute__((aligned(16))) float s_acc[2][4][64][4];
  __shared__ float s_h2[16][17];
  __shared__ float s_wh[12][16];
  __shared__ float s_b2[16], s_hb[12];
  __shared__ __attribute__((aligned(16))) float s_out[16][12];
  __shared__ int s_go[2];
  __shared__ __attribute__((aligned(16))) f32x4 s_red[32][64];  // device-side sum of the partial heads: [column tile][(g, row) lane] granules (32 KB)
  const int H = p.H, S = p.S;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: an SGPR
  // grid = (H / 16 column tiles) x (row tiles of 16): workgroup (tile, rt) owns rows 16 rt .. 16 rt + 15 of column tile `tile`.
  // Splitting the ROWS over workgroups (instead of a second accumulator set per wave: 411 VGPRs, +1.2 us of compute per step)
  // keeps a step's compute at the one-tile 1.65 us whatever the number of rows
  const int tiles_n = H / 16;
  const int tile = blockIdx.x % tiles_n, rt = blockIdx.x / tiles_n, n0 = tile * 16;
  const int r = lane & 15, kq = lane >> 4;
  const int kbeg = wid * 16 * NCH;  // H == 64 * NCH
  // ---- one-time: this lane's weight fragments into registers
  float w2f[NCH][4], b1f[NCH][4], w1f[NCH][4][W1LDS ? 1 : SP];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int kb = kbeg + 16 * u + 4 * kq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w2f[u][j] = p.W2[(size_t)(n0 + r) * H + kb + j];
      b1f[u][j] = p.b1[kb + j];
      if (!W1LDS) {
#pragma unroll
        for (int q = 0; q < SP; ++q) w1f[u][j][q] = q < S ? p.W1[(size_t)(kb + j) * S + q] : 0.f;
      }
    }
  }
  if (W1LDS) {
    for (int i = threadIdx.x; i < H * SP; i += 256) {
      const int k = i / SP, q = i - k * SP;
      s_w1[i] = q < S ? p.W1[(size_t)k * S + q] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < 12 * 16; i += 256) {
    const int o = i >> 4, c = i & 15;
    s_wh[o][c] = o < p.n_out ? p.wh[o][n0 + c] : 0.f;
  }
  if (threadIdx.x < 16) s_b2[threadIdx.x] = p.b2[n0 + threadIdx.x];
  if (threadIdx.x < 12) s_hb[threadIdx.x] = threadIdx.x < p.n_out ? *p.hbias[threadIdx.x] : 0.f;
  for (int i = threadIdx.x; i < 2 * 16 * SP; i += 256) (&s_x[0][0][0])[i] = 0.f;  // rows >= W / columns >= S stay 0
  for (int i = threadIdx.x; i < D * 128 * PI; i += 256) (&s_ring[0][0])[i] = 0ull;
  __syncthreads();

  // the window of observation granules this workgroup waits for: all of them, or (split) its row tile's rows -- S x 16 x 8 bytes = a multiple of 16
  const int g0 = p.split ? rt * 16 * S : 0;
  const int n_gran = p.split ? (p.W - 16 * rt < 16 ? p.W - 16 * rt : 16) * S : p.W * S;  // <= 128 PI: lane l owns granules g0 + l + 64 j
  const int n16 = (n_gran + 1) >> 1;     // 16-byte pieces per poll (two granules each); instruction q fetches pieces 64 q .. 64 q + 63
  const char* my_src = reinterpret_cast<const char*>(p.obs_gran + g0) + 16 * (lane < n16 ? lane : 0);
  const bool poller = !p.mbox || (p.split ? tile == 0 : blockIdx.x == 0);  // polls the HOST (and relays to its row tile / to everybody)
  unsigned long long next_issue = 0;
  int slot = 0;  // ring slot of the OLDEST poll in flight (wave 0)
  const unsigned ring_lane = (unsigned)(uintptr_t)&s_ring[0][0] + 8u * (unsigned)lane;  // LDS byte address of this lane's granule in slot 0
  // Every load issued so far (the weight fragments) must have LANDED here: hipcc would otherwise wait for them
  // lazily at their first use INSIDE the step loop with s_waitcnt vmcnt(N..0), and a vmcnt(0) in the loop body
  // also waits for the polls in flight and the previous step's PCIe store (measured: +2.4 us per step).  The
  // builtin (unlike an asm wait) clears the compiler's own scoreboard.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
  if (wid == 0 && poller) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int q = 0; q < PI; ++q)
        if (64 * q + lane < n16) persist_poll_issue(my_src + 1024 * q, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&s_ring[d][0]) + 1024u * q);
    }
    next_issue = wall_clock64();
  }

  for (int t = 1; t <= p.T; ++t) {
    const int par = t & 1;
    const unsigned tag = p.seq0 + (unsigned)t;
    // ---- wait for the host's observations of step t
    // Relay (p.mbox): only workgroup 0 polls the HOST; it republishes the granules in device memory and the other
    // workgroups poll that.  32 workgroups x 256 bytes of PCIe reads in flight queue behind each other on the
    // link's non-posted request tags (measured: every extra poll in flight per workgroup ADDS microseconds);
    // one poller sees the bare PCIe round trip, and the fan-out is an on-chip hand-off (~1 us).
    if (wid == 0 && !poller) {
      bool ok = false;
      for (long spin = 0; spin < p.max_polls * 4; ++spin) {
        unsigned long long gq[NJ];
        bool mine = true;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int g = 64 * j + lane;
          gq[j] = (j == 0 || 64 * j < n_gran) ? __hip_atomic_load(p.mbox + g0 + (g < n_gran ? g : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
          mine = mine && (g >= n_gran || (unsigned)(gq[j] >> 32) == tag);
        }
        if (__all(mine)) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int g = 64 * j + lane, gg = g0 + g;
            if (g < n_gran && (gg / S) >> 4 == rt) s_x[par][(gg / S) & 15][gg % S] = __uint_as_float((unsigned)gq[j]);
          }
          ok = true;
          break;
        }
        if ((spin & 1023) == 1023 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
      }
      if (lane == 0) s_go[par] = ok ? 1 : 0;
    } else if (wid == 0) {
      bool ok = false;
      for (long spin = 0; spin < p.max_polls; ++spin) {
        // the oldest of the D polls in flight has landed once at most D-1 are outstanding (loads return in order;
        // this wave's granule stores of the previous step can only make the wait longer, never shorter)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PI * (D - 1)) : "memory");
        // (a ds_read in asm: through a generic pointer hipcc emits flat_load + s_waitcnt vmcnt(0), which would wait
        // for EVERY poll in flight and serialise the ring)
        unsigned long long gq[NJ];
        {
          const unsigned ra = ring_lane + (unsigned)slot * (1024u * PI);
          // (reads and their wait are ONE statement with early-clobber outputs: an asm read's destination counts as written when the
          // statement ends, so with the wait in a later statement the compiler is free to move the registers while the data is in flight)
          if constexpr (PI == 1) {
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)" : "=&v"(gq[0]), "=&v"(gq[1]) : "v"(ra) : "memory");
          } else if constexpr (PI == 2) {
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(gq[0]), "=&v"(gq[1]), "=&v"(gq[2]), "=&v"(gq[3]) : "v"(ra) : "memory");
          } else if constexpr (PI == 3) {
            asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:512\n\tds_read_b64 %2, %6 offset:1024\n\tds_read_b64 %3, %6 offset:1536\n\t"
                         "ds_read_b64 %4, %6 offset:2048\n\tds_read_b64 %5, %6 offset:2560\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(gq[0]), "=&v"(gq[1]), "=&v"(gq[2]), "=&v"(gq[3]), "=&v"(gq[4]), "=&v"(gq[5]) : "v"(ra) : "memory");
          } else {
            static_assert(PI == 4, "1 .. 4 poll instructions per poll are spelled out");
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                         "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(gq[0]), "=&v"(gq[1]), "=&v"(gq[2]), "=&v"(gq[3]), "=&v"(gq[4]), "=&v"(gq[5]), "=&v"(gq[6]), "=&v"(gq[7]) : "v"(ra) : "memory");
          }
        }
        bool mine = true;
#pragma unroll
        for (int j = 0; j < NJ; ++j) mine = mine && (64 * j + lane >= n_gran || (unsigned)(gq[j] >> 32) == tag);
        const bool done = __all(mine);
        if (done) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int g = 64 * j + lane, gg = g0 + g;
            if (g < n_gran) {
              if ((gg / S) >> 4 == rt) s_x[par][(gg / S) & 15][gg % S] = __uint_as_float((unsigned)gq[j]);
              if (p.mbox) __hip_atomic_store(p.mbox + gg, gq[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: the granule is its own flag
            }
          }
        }
        // re-arm the slot (the LDS read above has returned), paced so that the D polls in flight stay ~period apart
        // instead of bunching up behind the one that just landed (not when the step was just detected: nothing may
        // delay the compute)
        if (p.period > 0 && !done) {
          while (wall_clock64() < next_issue) __builtin_amdgcn_s_sleep(1);
        }
        next_issue = wall_clock64() + (unsigned long long)p.period;
#pragma unroll
        for (int q = 0; q < PI; ++q)
          if (64 * q + lane < n16) persist_poll_issue(my_src + 1024 * q, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&s_ring[slot][0]) + 1024u * q);
        slot = __builtin_amdgcn_readfirstlane(slot + 1 == D ? 0 : slot + 1);
        if (done) { ok = true; break; }
        if ((spin & 255) == 255 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
      }
      if (lane == 0) s_go[par] = ok ? 1 : 0;
    }
    __syncthreads();
    if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) { p.dbg[(t - 1) * 8 + 0] = wall_clock64(); p.dbg[(t - 1) * 8 + 1] = __builtin_readcyclecounter(); }
    if (!s_go[par]) {  // timeout or host abort: tell the host and leave (all workgroups decide alike or time out too)
      if (threadIdx.x == 0) __hip_atomic_store(p.abort_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    // ---- layer 1 generated in registers as the A operand, straight into the MFMAs of this wave's K quarter
    float xr[SP];
#pragma unroll
    for (int q4 = 0; q4 < SP / 4; ++q4) {
      const float4 v = *reinterpret_cast<const float4*>(&s_x[par][r][4 * q4]);
      xr[4 * q4] = v.x; xr[4 * q4 + 1] = v.y; xr[4 * q4 + 2] = v.z; xr[4 * q4 + 3] = v.w;
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f;
        if (W1LDS) {
          const float* wr = s_w1 + (size_t)(kbeg + 16 * u + 4 * kq + j) * SP;
#pragma unroll
          for (int q4 = 0; q4 < SP / 4; ++q4) {
            const float4 w = *reinterpret_cast<const float4*>(wr + 4 * q4);
            a = fmaf(xr[4 * q4], w.x, a); a = fmaf(xr[4 * q4 + 1], w.y, a); a = fmaf(xr[4 * q4 + 2], w.z, a); a = fmaf(xr[4 * q4 + 3], w.w, a);
          }
        } else {
#pragma unroll
          for (int q = 0; q < SP; ++q) a = fmaf(xr[q], w1f[u][j][q], a);
        }
        a += b1f[u][j];
        a = a > 0.f ? a : 0.f;
        // two accumulators: the 16x16x4 fp32 MFMA has a 40-cycle dependent latency vs a 32-cycle issue interval
        if (j & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w2f[u][j], acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w2f[u][j], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s_acc[par][wid][lane][i] = acc[i] + acc2[i];
    __syncthreads();
    if (wid == 0) {
      // C/D fragment: col = lane & 15 (hidden-2 column n0 + r), row = kq * 4 + i (env row)
      float hv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = ((s_acc[par][0][lane][i] + s_acc[par][1][lane][i]) + s_acc[par][2][lane][i]) + s_acc[par][3][lane][i];
        v += s_b2[r];
        hv[i] = v > 0.f ? v : 0.f;
      }
      // h2 tile -> LDS in A-operand order; the head partials part[row][o] = sum_col h2[row][col] * Wh[o][n0 + col]
      // are one more 16x16x16 product on the MFMA (4 steps)
#pragma unroll
      for (int i = 0; i < 4; ++i) s_h2[kq * 4 + i][r] = hv[i];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      f32x4 ph = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float av = s_h2[r][4 * c + kq];                        // A[row r][k = col 4c+kq]
        const float bv = r < 12 ? s_wh[r][4 * c + kq] : 0.f;         // B[k = col][n = output r]
        ph = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, ph, 0, 0, 0);
      }
      // ph: col = lane & 15 -> output o = r, row = kq * 4 + i -> env row
      if (r < 12) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_out[kq * 4 + i][r] = ph[i] + (tile == 0 ? s_hb[r] : 0.f);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      // 16-byte row granules {out 3g, out 3g+1, out 3g+2, tag}: the partial IS its own flag.  Lanes (g, row) store
      // W consecutive granules per g with ONE instruction; fire and forget (no fence, no acknowledgement wait)
      const int g = lane >> 4, row = lane & 15;
      if (g < p.G && 16 * rt + row < p.W) {
        const f32x4 gq = (f32x4){s_out[row][3 * g], s_out[row][3 * g + 1], s_out[row][3 * g + 2], __uint_as_float(tag)};
        // write-through 16-byte store (a plain store lingers in L2 for milliseconds); hipcc does not track asm stores: the s_nop keeps
        // the data registers intact until the store has read them (CDNA guide §5.7).  Host memory, or -- round 6 -- the device-side
        // mailbox of this (row tile, g, row): the column tiles' granules side by side for the reducing wave below
        if (p.dpart) {
          float4* dst = p.dpart + ((size_t)(rt * p.G + g) * 16 + row) * tiles_n + tile;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(gq) : "memory");
        } else {
          float4* dst = p.part + ((size_t)tile * p.G + g) * p.rows_ld + 16 * rt + row;
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(gq) : "memory");
        }
      }
    }
    // ---- round 6: the column tiles' partial heads are summed on the DEVICE.  With every workgroup answering the host
    // itself, a step of config.ppo.mujoco's 32 workers x 7 outputs sends 32 tiles x 3 granules x 32 rows = 48 KB over PCIe and the host
    // gathers and adds them (14 us from publication to the last granule read).  Here column tile 0's workgroup -- idle from now to the next
    // step -- collects the tiles' granules from the device mailbox: its four waves each fetch a quarter of the tiles for every (g, row)
    // lane (agent-coherent 16-byte loads, all of a wave's in flight together; the granule's tag is its flag) into LDS, then ONE wave adds
    // them IN TILE ORDER (the host's order: the same bits) and sends one granule per (row, g) to the host, into tile 0's slots.
    if (p.dpart && tile == 0) {
      const int g = lane >> 4, row = lane & 15;
      const bool mine = g < p.G && 16 * rt + row < p.W;
      const int per = tiles_n / 4;  // tiles per wave: 1, 2, 4, 8 (hidden 64 .. 512)
      bool ok = true;
      if (mine) {
        const float4* src = p.dpart + ((size_t)(rt * p.G + g) * 16 + row) * tiles_n + wid * per;
        f32x4 q[8];
        long spin = 0;
        for (;;) {
          // the loads of one round and their wait are ONE asm statement (early-clobber outputs: see jh_tgemm.hip's split-K reduce); a wave
          // with fewer than 8 tiles re-reads its last one (clamped offsets) and drops the copies
          const float4* a1 = per > 4 ? src + 4 : src;
          asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
                       "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %12, off offset:16 sc1\n\t"
                       "global_load_dwordx4 %6, %12, off offset:32 sc1\n\tglobal_load_dwordx4 %7, %12, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                       : "v"(src), "v"(src + (per > 1 ? 1 : 0)), "v"(src + (per > 2 ? 2 : 0)), "v"(src + (per > 2 ? 3 : 0)), "v"(a1)
                       : "memory");
          bool all = true, newer = false;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned seen = __float_as_uint(q[j][3]);
            if (j < per) {  // (the clamped re-reads beyond this wave's tiles are somebody else's granules: not waited for)
              all = all && seen == tag;
              newer = newer || (int)(seen - tag) > 0;  // a later step's granule: this step's answer is no longer wanted (nobody waits for it)
            }
          }
          if (all) break;
          if (newer || ++spin > 2000000L || ((spin & 1023) == 1023 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) { ok = false; break; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < per) s_red[wid * per + j][lane] = ok ? q[j] : (f32x4){0.f, 0.f, 0.f, __uint_as_float(tag + 0x40000000u)};  // (a tag no step carries: the row is dropped below)
      }
      __syncthreads();  // (uniform: every wave of this workgroup is here)
      if (wid == 1 && mine) {
        float z0 = 0.f, z1 = 0.f, z2 = 0.f;
        bool good = true;
        for (int tt = 0; tt < tiles_n; ++tt) {
          const f32x4 v = s_red[tt][lane];
          good = good && __float_as_uint(v[3]) == tag;
          z0 += v[0]; z1 += v[1]; z2 += v[2];
        }
        if (good) {
          const f32x4 gq = (f32x4){z0, z1, z2, __uint_as_float(tag)};
          float4* dst = p.part + (size_t)g * p.rows_ld + 16 * rt + row;
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(gq) : "memory");
        }
      }
      // (s_red is rewritten one step later, behind the next step's barriers: the summing wave is long done)
    }
    // no end-of-step barrier: s_x / s_acc alternate by step parity, s_h2 / s_out belong to wave 0
    if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) { p.dbg[(t - 1) * 8 + 4] = wall_clock64(); p.dbg[(t - 1) * 8 + 5] = __builtin_readcyclecounter(); }
  }
  // the polls still in flight target this workgroup's LDS: let them land before the wave ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------- host side
constexpr int kPersistMaxGranules = 512;  // observation values per exchange: 128 per poll instruction, up to four of them (jh_collect.hip mirrors the limit)
struct jh_persist {
  jh_pponet* net = nullptr;
  unsigned long long* gran_h = nullptr;
  unsigned long long* gran_d = nullptr;
  float4 *part_h = nullptr, *part_d = nullptr;
  unsigned *flag_h = nullptr, *flag_d = nullptr;  // abort word
  unsigned seq = 0;
  int tiles = 0, n_out = 0, G = 0;
  int rows_ld = 16;  // rows per (tile, g) block of `part` in the running kernel: 16 RT
  unsigned long long *dbg_h = nullptr, *dbg_d = nullptr;
  unsigned long long* mbox = nullptr;  // device: relay of the observation granules
  float4* dpart = nullptr;             // device: the column tiles' partial heads, summed by a wave of the kernel (round 6)
  bool reduced = false;                // the running kernel answers with ONE granule per (row, g) in tile 0's slots
  int rows_published = 0;              // rows of the last jh_persist_publish (a two-timestep exchange carries more rows than its first read asks for)
  bool split = false;                  // the running kernel exchanges its two row tiles independently (jh_persist_publish_rows)
};

// heads needed to ACT: A logits (discrete) | A mu + A log_std (continuous) (ppo.py:55-69) -- plus the value head as the LAST
// output: V(s_t) of the states acted on is exactly what PPO.learn recomputes with the same weights in its no-grad pass
// (ppo.py:83-94; in sync mode the actors' weights ARE the learner's), so the collector can hand it over instead
// (jh_collector_set_capture).  Discrete A = 2: the third float of the one 16-byte granule per (tile, row) was unused.
static int persist_heads(const jh_pponet* n) { return (n->cont ? 2 * n->A : n->A) + 1; }

int jh_persist_create(jh_pponet* n, jh_persist** out) {
  JH_ARG(n && out);
  const int H = n->H, S = n->S;
  JH_ARG(S >= 1 && S <= 16 && (H == 64 || H == 128 || H == 256 || H == 512) && persist_heads(n) <= 12);
  jh_persist* p = new jh_persist();
  p->net = n;
  p->tiles = H / 16;
  p->n_out = persist_heads(n);
  p->G = (p->n_out + 2) / 3;
  JH_HIP(hipHostMalloc((void**)&p->gran_h, sizeof(unsigned long long) * kPersistMaxGranules, hipHostMallocMapped));
  JH_HIP(hipHostGetDevicePointer((void**)&p->gran_d, p->gran_h, 0));
  const size_t part_bytes = sizeof(float4) * 32 * (size_t)p->tiles * p->G;  // room for two row tiles
  JH_HIP(hipHostMalloc((void**)&p->part_h, part_bytes, hipHostMallocMapped));
  memset(p->part_h, 0, part_bytes);
  JH_HIP(hipHostGetDevicePointer((void**)&p->part_d, p->part_h, 0));
  JH_HIP(hipHostMalloc((void**)&p->flag_h, sizeof(unsigned) * 16, hipHostMallocMapped));
  JH_HIP(hipHostGetDevicePointer((void**)&p->flag_d, p->flag_h, 0));
  JH_HIP(hipMalloc((void**)&p->mbox, sizeof(unsigned long long) * kPersistMaxGranules));
  JH_HIP(hipMemset(p->mbox, 0, sizeof(unsigned long long) * kPersistMaxGranules));
  JH_HIP(hipMalloc((void**)&p->dpart, sizeof(float4) * (2 * 4 * 16 * (size_t)p->tiles + 8)));  // (+ 8: a wave with fewer than 8 tiles reads a few granules past its last one and drops them)
  JH_HIP(hipMemset(p->dpart, 0, sizeof(float4) * (2 * 4 * 16 * (size_t)p->tiles + 8)));
  memset(p->gran_h, 0, sizeof(unsigned long long) * kPersistMaxGranules);
  memset(p->flag_h, 0, sizeof(unsigned) * 16);
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
  (void)hipFree(p->mbox);
  (void)hipFree(p->dpart);
  if (p->dbg_d) (void)hipFree(p->dbg_d);
  free(p->dbg_h);
  delete p;
}

// pi: poll instructions per poll (1: up to 128 granules -- all poll depths; 2-4: the default depth of four polls in flight only)
template <int SP, int NCH>
static void persist_launch(int depth, int pi, int grid, hipStream_t st, const PersistArgs& a) {
  if (pi == 2) JH_LAUNCH_NAMED("jh_act_persist_kernel", (jh_act_persist_kernel<SP, NCH, 4, 2>), dim3(grid), dim3(256), 0, st, a);
  else if (pi == 3) JH_LAUNCH_NAMED("jh_act_persist_kernel", (jh_act_persist_kernel<SP, NCH, 4, 3>), dim3(grid), dim3(256), 0, st, a);
  else if (pi == 4) JH_LAUNCH_NAMED("jh_act_persist_kernel", (jh_act_persist_kernel<SP, NCH, 4, 4>), dim3(grid), dim3(256), 0, st, a);
  else if (depth >= 4) JH_LAUNCH_NAMED("jh_act_persist_kernel", (jh_act_persist_kernel<SP, NCH, 4, 1>), dim3(grid), dim3(256), 0, st, a);
  else if (depth >= 2) JH_LAUNCH_NAMED("jh_act_persist_kernel", (jh_act_persist_kernel<SP, NCH, 2, 1>), dim3(grid), dim3(256), 0, st, a);
  else JH_LAUNCH_NAMED("jh_act_persist_kernel", (jh_act_persist_kernel<SP, NCH, 1, 1>), dim3(grid), dim3(256), 0, st, a);
}

// Launch the persistent kernel for T steps of W <= 32 rows (W * S <= 512 observation granules).
// split: the two row tiles of a W = 32 exchange advance independently (tags per row tile; jh_persist_publish_rows / jh_persist_collect_rows).
int jh_persist_begin(jh_persist* p, int W, int T, hipStream_t st, int split) {
  jh_pponet* n = p->net;
  JH_ARG(W > 0 && W <= 32 && W * n->S <= kPersistMaxGranules && T > 0 && (!split || W == 32));
  const int win = split ? 16 * n->S : W * n->S;  // observation granules a workgroup polls
  const int pi = ((win + 1) / 2 + 63) / 64;  // exactly the instructions that have a lane to fetch for (see the kernel)
  PersistArgs a{};
  a.W = W; a.S = n->S; a.H = n->H; a.T = T; a.G = p->G;
  a.W1 = n->params + n->o_w1; a.b1 = n->params + n->o_b1; a.W2 = n->params + n->o_w2; a.b2 = n->params + n->o_b2;
  int o = 0;
  for (int k = 0; k < n->A; ++k, ++o) { a.wh[o] = n->params + n->o_wh0 + (int64_t)k * n->H; a.hbias[o] = n->params + n->o_bh0 + k; }
  if (n->cont)
    for (int k = 0; k < n->A; ++k, ++o) { a.wh[o] = n->params + n->o_wh1 + (int64_t)k * n->H; a.hbias[o] = n->params + n->o_bh1 + k; }
  a.wh[o] = n->params + n->o_wv; a.hbias[o] = n->params + n->o_bv; ++o;  // value head last
  a.n_out = o;
  a.obs_gran = p->gran_d; a.part = p->part_d; a.abort_flag = p->flag_d;
  a.seq0 = p->seq;
  a.dbg = p->dbg_d;
  // polls in flight per workgroup and their spacing (10 ns ticks): 4 x 0.4 us covers a ~1.6 us PCIe read round trip
  static const int depth = getenv("JH_PERSIST_DEPTH") ? atoi(getenv("JH_PERSIST_DEPTH")) : 4;
  static const int period = getenv("JH_PERSIST_PERIOD") ? atoi(getenv("JH_PERSIST_PERIOD")) : 10;
  a.period = depth > 1 ? period : 0;
  static const int relay = getenv("JH_PERSIST_RELAY") ? atoi(getenv("JH_PERSIST_RELAY")) : 1;
  a.mbox = relay ? p->mbox : nullptr;
  // device-side sum of the partial heads: JH_PERSIST_REDUCE=1.  OFF by default: measured (profiles/r06_ab_persist_device_reduce.txt), the
  // on-device hand-off -- write-through granules of 64 workgroups on 8 XCDs, fetched agent-coherently by one workgroup -- costs 3.2 us
  // per step, more than the host's gather of the 48 KB it replaces (config.ppo.mujoco, 32 workers: publication-to-heads 15.2 us direct,
  // 15.8 us reduced; 8 workers: 6.9 vs 8.4 us).  The cross-XCD visibility of a store is the price, not the bytes.
  // ... but ON by default for SPLIT exchanges (end of round 6): there the extra hand-off latency hides under the other half's host work, and what is left on the
  // host's critical path -- reading 48 KB of partial heads per timestep (~4 us) -- shrinks to 1.5 KB: configs[4] end to end 0.79 -> 0.88 M env transitions/s.
  const int reduce = getenv("JH_PERSIST_REDUCE") ? atoi(getenv("JH_PERSIST_REDUCE")) : (split ? 1 : 0);  // (read per launch: the tests switch it)
  p->reduced = reduce == 1;
  a.dpart = p->reduced ? p->dpart : nullptr;
  a.split = split ? 1 : 0;
  p->split = split != 0;
  a.max_polls = 600000;  // x (>= 0.3 us per consumed poll) = >= 0.2 s without observations -> give up
  p->flag_h[0] = 0;
  const int nch = n->H / 64;
  const int sp = (n->S + 3) / 4 * 4;  // 4, 8: W1 fragments in registers; 12, 16: in LDS
  const int rt = W > 16 ? 2 : 1;  // row tiles = workgroups per column tile
  p->rows_ld = 16 * rt;
  a.rows_ld = p->rows_ld;
  const int grid = p->tiles * rt;
#define JH_PERSIST_CASE(NCH)                                   \
  if (nch == NCH) {                                            \
    if (sp == 4) persist_launch<4, NCH>(depth, pi, grid, st, a);   \
    else if (sp == 8) persist_launch<8, NCH>(depth, pi, grid, st, a);   \
    else if (sp == 12) persist_launch<12, NCH>(depth, pi, grid, st, a); \
    else persist_launch<16, NCH>(depth, pi, grid, st, a);          \
  }
  JH_PERSIST_CASE(1) else JH_PERSIST_CASE(2) else JH_PERSIST_CASE(4) else JH_PERSIST_CASE(8)
#undef JH_PERSIST_CASE
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// Publish the observations of all W env rows for the next timestep; returns its tag.
unsigned jh_persist_publish(jh_persist* p, int W, const float* h_obs) {
  const unsigned tag = ++p->seq;
  p->rows_published = W;
  const int n = W * p->net->S;
  for (int i = 0; i < n; ++i) {
    unsigned bits;
    memcpy(&bits, h_obs + i, 4);
    __atomic_store_n(p->gran_h + i, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELEASE);
  }
  return tag;
}

// Split exchanges: publish rows r0 .. r1-1 (h_obs: the FULL [W][S] array) under `tag` = the persist's sequence number at jh_persist_begin + the step (1 ..).
unsigned jh_persist_seq(const jh_persist* p) { return p->seq; }
void jh_persist_publish_rows(jh_persist* p, int r0, int r1, const float* h_obs, unsigned tag) {
  const int S = p->net->S;
  if (r1 > p->rows_published) p->rows_published = r1;
  for (int i = r0 * S; i < r1 * S; ++i) {
    unsigned bits;
    memcpy(&bits, h_obs + i, 4);
    __atomic_store_n(p->gran_h + i, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELEASE);
  }
  if ((int)(tag - p->seq) > 0) p->seq = tag;
}

// Wait until every tile's granules of the listed rows (rows == NULL: rows 0 .. n_rows-1) carry `tag`, then sum the per-tile partials
// in tile order: h_heads [n_rows][n_out] raw head outputs (logits | mu_raw, log_std_raw; value last).  JH_ERR_STATE if the kernel gave up.
static int persist_collect(jh_persist* p, const int* rows, int n_rows, unsigned tag, float* h_heads, int r_first, bool block_order);
int jh_persist_collect_rows(jh_persist* p, const int* rows, int n_rows, unsigned tag, float* h_heads) {
  return persist_collect(p, rows, n_rows, tag, h_heads, 0, !rows && n_rows > 16);
}
// rows r0 .. r0 + n_rows - 1 in the blocks' storage order (a split exchange's half: 16 consecutive row granules per (tile, g) block)
int jh_persist_collect_range(jh_persist* p, int r0, int n_rows, unsigned tag, float* h_heads) { return persist_collect(p, nullptr, n_rows, tag, h_heads, r0, true); }
static int persist_collect(jh_persist* p, const int* rows, int n_rows, unsigned tag, float* h_heads, int r_first, bool block_order) {
  const int n_out = p->n_out, G = p->G, ld = p->rows_ld, tiles = p->reduced ? 1 : p->tiles;  // (reduced: the device summed the tiles; tile 0's slots hold the answer)
  volatile unsigned* abort_w = p->flag_h;
  // [tiles][G][ld] granules of 16 bytes {out, out, out, tag}: ONE 16-byte load per granule serves the tag check and the sum (the
  // device's 16-byte store is one PCIe write: tag and payload arrive together); a row is summed tile by tile as its granules are
  // found, and only the missing rows are re-polled
  struct alignas(16) Gran16 { unsigned w[4]; };
  const Gran16* part = reinterpret_cast<const Gran16*>(p->part_h);
  float z[32][12];
  unsigned char have[32];
  JH_ARG(n_rows <= 32);
  if (block_order) {
    // All rows 0 .. n_rows-1 of a two-row-tile exchange (up to 16 rows keep the row-major walk below: for config.ppo.cartpole's 8
    // root rows the block order measured 1.4 % of the whole step SLOWER -- it waits block by block for rows the other order has
    // already summed): walk the
    // (tile, g) blocks in storage order -- n_rows consecutive granules each, so the 48 KB of a 32-row x 7-output step stream through
    // the host's prefetcher instead of being gathered row by row with a 512-byte stride (round 5: 22.8 -> 18.9 us per step at
    // config.ppo.mujoco's 32 workers).  Per row the partials are still added in tile order: the same bits.
    for (int k = 0; k < n_rows; ++k)
      for (int o = 0; o < 12; ++o) z[k][o] = 0.f;
    long spins = 0;
    for (int t = 0; t < tiles; ++t)
      for (int g = 0; g < G; ++g) {
        const Gran16* blk = part + ((size_t)t * G + g) * ld + r_first;
        for (;;) {  // every row's granule of this block carries the tag (tag first, acquire: see below)
          bool all = true;
          for (int k = 0; k < n_rows; ++k)
            if (__atomic_load_n(reinterpret_cast<const unsigned*>(blk + k) + 3, __ATOMIC_ACQUIRE) != tag) { all = false; break; }
          if (all) break;
          if (++spins >= 40000000L || ((spins & 1023) == 1023 && *abort_w == 2u))
            return jh_fail(JH_ERR_STATE, "persistent acting kernel did not answer step tag %u", tag);
          __builtin_ia32_pause();
        }
        for (int k = 0; k < n_rows; ++k) {
          alignas(16) unsigned w4[4];
#if defined(__x86_64__)
          _mm_store_si128(reinterpret_cast<__m128i*>(w4), _mm_load_si128(reinterpret_cast<const __m128i*>(blk + k)));
#else
          memcpy(w4, blk + k, 16);
#endif
          if (w4[3] != tag) return jh_fail(JH_ERR_STATE, "persistent acting kernel: granule of step tag %u rewritten while it was read", tag);
          float f3[3];
          memcpy(f3, w4, 12);
          z[k][3 * g] += f3[0]; z[k][3 * g + 1] += f3[1]; z[k][3 * g + 2] += f3[2];
        }
      }
    for (int k = 0; k < n_rows; ++k) memcpy(h_heads + (size_t)k * n_out, z[k], sizeof(float) * n_out);
    return JH_OK;
  }
  int missing = n_rows;
  const int pf_rows = (!rows && p->rows_published > n_rows) ? p->rows_published - n_rows : 0;
  for (int k = 0; k < n_rows; ++k) have[k] = 0;
  for (long spin = 0; spin < 40000000L && missing; ++spin) {
    for (int k = 0; k < n_rows; ++k) {
      if (have[k]) continue;
      const int wq = rows ? rows[k] : k;
      float acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      bool ok = true;
      for (int t = 0; t < tiles && ok; ++t)
        for (int g = 0; g < G; ++g) {
          // tag first (acquire), payload after it: the device writes a granule as ONE 16-byte store, so a host that has seen the new
          // tag sees the new payload in any LATER load (x86: loads are not reordered with older loads).  This does not lean on the
          // 16-byte load itself being single-copy atomic (guaranteed on AVX-capable CPUs only); the tag inside the value is
          // checked once more in case the granule was rewritten between the two loads
          const Gran16* gp = part + ((size_t)t * G + g) * ld + wq;
          if (__atomic_load_n(reinterpret_cast<const unsigned*>(gp) + 3, __ATOMIC_ACQUIRE) != tag) { ok = false; break; }
#if defined(__x86_64__)
          // the first read of a two-timestep exchange (rows 0 .. W-1 of 3 W): this workgroup wrote its granules of ALL rows with the same
          // store instruction, so the successors' rows of this (tile, g) are in host memory too -- pull them towards the cache now; the
          // second read (the chosen successors, right after the sampling) then finds them there instead of missing 32 times per row
          // (round 5: 3.10 -> 2.92 us per timestep, 0.901 -> 0.882 ms per PPO step; also prefetching the first read's own rows 4 .. 7
          // added nothing: profiles/r05_ab_host_prefetch_ppo.txt)
          if (pf_rows > 0 && k == 0) {
            const char* q = reinterpret_cast<const char*>(part + ((size_t)t * G + g) * ld + n_rows);
            for (int b = 0; b < pf_rows * 16; b += 64) _mm_prefetch(q + b, _MM_HINT_T0);
          }
#endif
          alignas(16) unsigned w4[4];
#if defined(__x86_64__)
          _mm_store_si128(reinterpret_cast<__m128i*>(w4), _mm_load_si128(reinterpret_cast<const __m128i*>(gp)));
#else
          memcpy(w4, gp, 16);
#endif
          if (w4[3] != tag) { ok = false; break; }
          float f3[3];
          memcpy(f3, w4, 12);
          acc[3 * g] += f3[0]; acc[3 * g + 1] += f3[1]; acc[3 * g + 2] += f3[2];  // tile order: the same bits as the device-side reduce would give
        }
      if (ok) {
        memcpy(z[k], acc, sizeof(acc));
        have[k] = 1;
        --missing;
      }
    }
    if (missing) {
      if ((spin & 1023) == 1023 && *abort_w == 2u) break;  // the kernel timed out
      __builtin_ia32_pause();
    }
  }
  if (missing) return jh_fail(JH_ERR_STATE, "persistent acting kernel did not answer step tag %u", tag);
  for (int k = 0; k < n_rows; ++k) memcpy(h_heads + (size_t)k * n_out, z[k], sizeof(float) * n_out);
  return JH_OK;
}

int jh_persist_collect(jh_persist* p, int W, unsigned tag, float* h_heads) { return jh_persist_collect_rows(p, nullptr, W, tag, h_heads); }

int jh_persist_heads(const jh_persist* p) { return p->n_out; }

// Diagnostics: print per-step phase durations of workgroup 0 for the last rollout.
void jh_persist_dump_debug(jh_persist* p, int T) {
  if (!p->dbg_h) return;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(p->dbg_h, p->dbg_d, sizeof(unsigned long long) * 8 * (size_t)T, hipMemcpyDeviceToHost);
  double a = 0, c = 0, cyc = 0;
  for (int t = 1; t < T; ++t) {
    const unsigned long long* d = p->dbg_h + (size_t)t * 8;
    const unsigned long long* pr = p->dbg_h + (size_t)(t - 1) * 8;
    a += (double)(d[0] - pr[4]);   // wait for observations (incl. host work + PCIe), 100 MHz ticks
    c += (double)(d[4] - d[0]);    // layer 1 + MFMA + combine + heads + store
    cyc += (double)(d[5] - d[1]) / ((double)(d[4] - d[0]) + 1e-9);  // shader cycles per 10 ns tick
  }
  const double n = T - 1;
  fprintf(stderr, "[jh_persist] per step (wall_clock64 ticks = 10 ns): wait %.1f  compute %.1f ; shader clock ~%.0f MHz\n",
          a / n, c / n, cyc / n * 100.0);
}

// Has the kernel of the last jh_persist_begin timed out waiting for observations (it sets the word to 2 and exits)?
bool jh_persist_gave_up(const jh_persist* p) { return __atomic_load_n(p->flag_h, __ATOMIC_ACQUIRE) == 2u; }

// Stop a running kernel early (error paths): it sees the word at its next abort check and exits.
void jh_persist_abort(jh_persist* p) { __atomic_store_n(p->flag_h, 1u, __ATOMIC_RELEASE); }
