// Pieces that more than one translation unit launches from: the PER leaf write-back as a device function (jh_per.hip's own
// kernel and the fused "C51 statistics + write-back" kernel of jh_dqn.hip run the same code), the C51 argument block, and the
// host entry points the Rainbow step (jh_rbnet.hip: jh_rbnet_c51_step) strings together.
//
// Why: at Rainbow's B = 32 the update is a chain of ~5-us launches: the dueling combine of the three forwards, the
// gradient through it, the C51 statistics and the PER leaf write-back were four of them; they are row-local prologues /
// epilogues of kernels that already hold the row.
#pragma once
#include "jh_common.h"

// ---------------------------------------------------------------------------------------------- PER write-back
constexpr int kPerChunk = 2048;  // items per pass, a power of two <= 4096 (LDS: 2048 * (8 + 8) B = 32 KiB)

// mode bits
#define PER_DISTINCT 1  // caller guarantees all leaves distinct (push of consecutive leaves)
#define PER_CONTIG 2    // caller guarantees each node's items are contiguous in batch order

struct PerDeltaArgs {
  double* tree;
  double* maxp;
  const int64_t* idx;  // tree indices, or null: push_start + i
  int64_t push_start;
  const void* prio;    // null: the running maximum (a push)
  int prio_dt, mode;
  int64_t tree_size, first_leaf;
  double* delta_out;   // [B] for the climb
  int climb_depth;     // > 0 (round 6): B <= kPerSmall items -- the climb runs in THIS launch (jh_per_climb_small), over depths 0 .. climb_depth - 1
};
constexpr int kPerSmall = 64;  // write-backs / pushes of up to this many items: leaves and climb in ONE launch

// In-LDS bitonic sort of n (power of two, <= kPerChunk) 64-bit keys by 256 threads.  Keys are
// (node << 12 | batch position): equal nodes become one contiguous run ordered by batch position,
// which is exactly the order in which the reference applies its `+= delta` to that node.
__device__ __forceinline__ void jh_bitonic_sort(unsigned long long* keys, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < n; t += 256) {
        const int ixj = t ^ j;
        if (ixj > t) {
          const unsigned long long a = keys[t], b = keys[ixj];
          const bool up = (t & k) == 0;
          if ((a > b) == up) { keys[t] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int jh_pow2_ge(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// Round 6: the climb of a SMALL batch (B <= 64: Rainbow's write-back of 32, a step's stores) inside the launch that wrote the leaves --
// one launch instead of two on the learn() chain.  Same arithmetic as jh_per_climb_kernel: for every depth, every node hit
// by the batch gets its deltas added in ascending batch order, float64, one rounding per add.  A WAVE per depth (different depths never
// share a node): the 64 lanes hold (node << 12 | batch position) keys, sorted with shuffles (no barriers), the head lane of a run of equal
// nodes adds the run.  Called by all 256 threads of the workgroup right behind jh_per_delta_body's leaf phase.
__device__ __forceinline__ int jh_node_depth(long long i) { return 63 - __clzll((unsigned long long)(i + 1)); }
__device__ __forceinline__ void jh_per_climb_small(const PerDeltaArgs& a, int B) {
  __shared__ double s_dl[kPerSmall];
  __shared__ unsigned long long s_k[4][64];
  __shared__ double s_sd[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // delta_out is written (and the callers' LDS is free)
  if (threadIdx.x < B) s_dl[threadIdx.x] = a.delta_out[threadIdx.x];
  __syncthreads();
  long long ix = 0;
  int dep = 0;
  if (lane < B) {
    ix = a.idx ? a.idx[lane] : a.push_start + lane;
    ix = ix < a.first_leaf ? a.first_leaf : (ix >= a.tree_size ? a.tree_size - 1 : ix);
    dep = jh_node_depth(ix);
  }
  for (int d = wid; d < a.climb_depth; d += 4) {
    unsigned long long key = ~0ull;
    if (lane < B && dep > d) key = ((unsigned long long)(((ix + 1) >> (dep - d)) - 1) << 12) | (unsigned long long)lane;
    // bitonic sort of the wave's 64 keys, ascending
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const unsigned long long other = __shfl_xor(key, j, 64);
        const bool keep_min = ((lane & k) == 0) == ((lane & j) == 0);
        key = keep_min ? (key < other ? key : other) : (key > other ? key : other);
      }
    }
    s_k[wid][lane] = key;
    s_sd[wid][lane] = key == ~0ull ? 0.0 : s_dl[(int)(key & 4095ull)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (key != ~0ull) {
      const unsigned long long node = key >> 12;
      if (lane == 0 || (s_k[wid][lane - 1] >> 12) != node) {  // head of this node's run
        double v = a.tree[node];
        for (int j = lane; j < 64 && (s_k[wid][j] >> 12) == node; ++j) v += s_sd[wid][j];
        a.tree[node] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the next depth of this wave rewrites s_k / s_sd)
  }
}

// One workgroup of 256 threads: delta_out[i] = new_i - (what the reference would find in the leaf at that moment), leaves
// written, *maxp raised.  s_key / s_new: kPerChunk entries of LDS each, s_red 16 doubles.
__device__ __forceinline__ void jh_per_delta_body(const PerDeltaArgs& a, int B, unsigned long long* s_key, double* s_new, double* s_red) {
  const double cur_max = *a.maxp;
  const int n2 = jh_pow2_ge(B);
  double my_max = cur_max;
  for (int i = threadIdx.x; i < n2; i += 256) {
    if (i < B) {
      int64_t ix = a.idx ? a.idx[i] : a.push_start + i;
      // a bad index must not corrupt internal nodes: clamp into the leaf range
      ix = ix < a.first_leaf ? a.first_leaf : (ix >= a.tree_size ? a.tree_size - 1 : ix);
      double p;
      if (!a.prio) p = cur_max;
      else if (a.prio_dt == JH_F32) p = (double)((const float*)a.prio)[i];  // fp32 tensor .item() -> python float
      else p = ((const double*)a.prio)[i];
      s_key[i] = ((unsigned long long)ix << 12) | (unsigned long long)i;
      s_new[i] = p;
      my_max = fmax(my_max, p);
    } else {
      s_key[i] = ~0ull;
    }
  }
  __syncthreads();
  if (!(a.mode & PER_DISTINCT)) jh_bitonic_sort(s_key, n2);
  // sorted position q holds item i = key & 4095 of leaf key >> 12.  Within a run of equal leaves the
  // reference sees: old = tree[leaf] for the first write, the previous write's value afterwards, and
  // the leaf ends up with the last write (per_buffer.py:42-46 applied in batch order).
  for (int q = threadIdx.x; q < B; q += 256) {
    const unsigned long long kq = s_key[q];
    const int i = (int)(kq & 4095ull);
    const int64_t leaf = (int64_t)(kq >> 12);
    const bool head = q == 0 || (s_key[q - 1] >> 12) != (unsigned long long)leaf;
    const double oldp = head ? a.tree[leaf] : s_new[(int)(s_key[q - 1] & 4095ull)];
    a.delta_out[i] = s_new[i] - oldp;
  }
  __syncthreads();  // all leaf reads are done before any leaf is overwritten
  for (int q = threadIdx.x; q < B; q += 256) {
    const unsigned long long kq = s_key[q];
    const int64_t leaf = (int64_t)(kq >> 12);
    const bool tail = q == B - 1 || (s_key[q + 1] >> 12) != (unsigned long long)leaf;
    if (tail) a.tree[leaf] = s_new[(int)(kq & 4095ull)];
  }
  const double m = jh_block_reduce(my_max, s_red, JhMax(), 0.0);
  if (threadIdx.x == 0) *a.maxp = m;  // max(max_priority, new...) per_buffer.py:48
  if (a.climb_depth > 0) jh_per_climb_small(a, B);
}

struct jh_per;
// jh_per.hip: the argument block of a general write-back of B <= kPerChunk (tree index, priority) pairs, and the climb behind it
int jh_per_delta_args(jh_per* p, int B, const int64_t* d_idx, const void* d_prio, int prio_dt, PerDeltaArgs* out);
int jh_per_climb(jh_per* p, int B, const int64_t* d_idx, hipStream_t st);

// ---------------------------------------------------------------------------------------------- C51
struct C51Args {
  int B, A, K, n, flags;
  const float *logit, *next_logit, *target_logit, *action, *reward, *done, *weights;
  float v_min, v_max, gamma, alpha;
  float *grad, *prio, *kl, *stats, *partial;
  const float* wmean;  // wave-per-sample kernel: the batch mean of `weights`, computed once by jh_mean_f32 in front of it
};

// Dueling heads around the loss (network/rainbow.py:88-93): sets 0 / 1 / 2 = online(state), online(next_state), target(next_state).
// The kernel forms logits = (xa - mean_a xa) + xv itself (and writes them to out[set], [B][A][K]), and returns the gradient
// already pulled through the combine: dxv = sum_a g, dxa = g - mean_a g.
struct C51Duel {
  const float* xa[3];
  const float* xv[3];
  float* out[3];
  float *dxa, *dxv;
  int ld_a, ld_v;
};

// jh_dqn.hip.  a.partial is filled in here.  duel: block kernel with the combine folded in (B <= 1024).  per: the statistics
// launch also writes the priorities a.prio back into the tree's leaves (the climb is the caller's next launch).
int jh_c51_run(jh_ctx* ctx, C51Args a, const C51Duel* duel, const PerDeltaArgs* per, hipStream_t st);
