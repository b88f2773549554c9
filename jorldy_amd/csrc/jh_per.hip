// Prioritized replay on the GPU: float64 sum tree in the reference's array-heap layout
// (core/buffer/per_buffer.py:7-105), bit-identical to the reference's numpy tree.
//
// Why it can be bit-identical: the reference never rebuilds the tree; every priority change is
//   delta = new - tree[leaf]; tree[leaf] = new; for every ancestor a: tree[a] += delta
// (per_buffer.py:42-54).  A node's value therefore depends only on the ORDER in which deltas reach
// it.  The batch kernels below keep that order per node:
//   * jh_per_delta_kernel resolves duplicate leaves sequentially (the 2nd write of a leaf in one
//     batch sees the 1st as its old value) and writes the leaves;
//   * jh_per_climb_kernel runs one workgroup per tree DEPTH (a node has exactly one depth, so no
//     two workgroups ever touch the same node) and, for every node hit by the batch, adds the
//     deltas in ascending batch order, in float64, one rounding per add -- exactly numpy's `+=`;
//   * both find "the items of one leaf / node, in batch order" with an in-LDS bitonic sort of
//     (node << 12 | batch position) keys: O(B log^2 B / 256) per workgroup instead of O(B^2).
// The descent (per_buffer.py:56-68) is a dependent chain of ~log2(N) 8-byte loads per sample:
// latency-bound, one lane per sample, `num <= left` goes left.
#include "jh_common.h"
#include "jh_fused.h"

namespace {
constexpr int kChunk = kPerChunk;  // items per kernel pass (jh_fused.h)

struct PerWs {
  double* delta = nullptr;   // [kChunk]
  double* prio = nullptr;    // [ws_cap] sampled priorities
  double* w = nullptr;       // [ws_cap] unnormalised weights
  double* partial = nullptr; // [2 * kMaxBlocks]
  int64_t ws_cap = 0;
};
constexpr int kMaxBlocks = 1024;
}  // namespace

struct jh_per {
  jh_ctx* ctx = nullptr;
  int64_t N = 0, tree_size = 0;
  double usp = 0;
  double* tree = nullptr;
  double* maxp = nullptr;  // device scalar, starts at 1.0 (per_buffer.py:16)
  PerWs ws;
  int64_t tree_index = 0;  // per_buffer.py:14, wraps independently of the store's buffer_index
  int64_t counter = 0;
  int depth_max = 0;        // depth of the deepest leaf = floor(log2(2N-1))
  int64_t deep_first = 0;   // first tree index at depth_max (= 2^depth_max - 1)
};

// ----------------------------------------------------------------------------- kernels
__device__ __forceinline__ int node_depth(int64_t i) { return 63 - __clzll((unsigned long long)(i + 1)); }

// leaf write-back: the body lives in jh_fused.h (the fused C51 statistics kernel of jh_dqn.hip runs it too)
__global__ void __launch_bounds__(256) jh_per_delta_kernel(PerDeltaArgs a, int B) {
  __shared__ unsigned long long s_key[kChunk];  // (leaf << 12) | i, sorted unless PER_DISTINCT
  __shared__ double s_new[kChunk];
  __shared__ double s_red[16];
  jh_per_delta_body(a, B, s_key, s_new, s_red);
}

__global__ void __launch_bounds__(256) jh_per_climb_kernel(double* __restrict__ tree, int B,
                                                           const int64_t* __restrict__ idx, int64_t push_start,
                                                           const double* __restrict__ delta, int mode,
                                                           int64_t tree_size, int64_t first_leaf) {
  __shared__ unsigned long long s_key[kChunk];  // (node << 12) | i ; ~0 for items without a node at this depth
  __shared__ double s_delta[kChunk];
  __shared__ double s_sd[kChunk];
  const int d = blockIdx.x;  // this workgroup owns every node at depth d
  const int n2 = jh_pow2_ge(B);
  for (int i = threadIdx.x; i < n2; i += 256) {
    unsigned long long key = ~0ull;
    if (i < B) {
      int64_t ix = idx ? idx[i] : push_start + i;
      ix = ix < first_leaf ? first_leaf : (ix >= tree_size ? tree_size - 1 : ix);
      const int dep = node_depth(ix);
      if (dep > d) key = ((unsigned long long)(((ix + 1) >> (dep - d)) - 1) << 12) | (unsigned long long)i;
      s_delta[i] = delta[i];
    }
    s_key[i] = key;
  }
  __syncthreads();
  // pushes of consecutive leaves (split at the wrap and the depth boundary by the host) already have
  // every node's items contiguous and in order; general write-backs are sorted
  if (!(mode & PER_CONTIG)) jh_bitonic_sort(s_key, n2);
  // deltas in sorted order, so the serial chain below streams LDS instead of chasing an index
  for (int q = threadIdx.x; q < B; q += 256) {
    const unsigned long long kq = s_key[q];
    s_sd[q] = kq == ~0ull ? 0.0 : s_delta[(int)(kq & 4095ull)];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < B; q += 256) {
    const unsigned long long kq = s_key[q];
    if (kq == ~0ull) continue;
    const unsigned long long node = kq >> 12;
    if (q > 0 && (s_key[q - 1] >> 12) == node) continue;  // not the head of this node's run
    // end of the run: first sorted position whose key exceeds (node, 4095) -- binary search
    const unsigned long long hi_key = (node << 12) | 4095ull;
    int lo = q, hi = B;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_key[mid] <= hi_key) lo = mid + 1;
      else hi = mid;
    }
    // ordered float64 chain: tree[node] += delta for every item that reaches the node, in batch order
    // (the loads do not depend on v, so they pipeline; only the adds are serial)
    double v = tree[node];
    int j = q;
    for (; j + 4 <= lo; j += 4) {
      const double d0 = s_sd[j], d1 = s_sd[j + 1], d2 = s_sd[j + 2], d3 = s_sd[j + 3];
      v += d0;
      v += d1;
      v += d2;
      v += d3;
    }
    for (; j < lo; ++j) v += s_sd[j];
    tree[node] = v;
  }
}

// one lane per sample: descent + importance weight (unnormalised) + per-block partials
__global__ void __launch_bounds__(256) jh_per_sample_kernel(const double* __restrict__ tree, int64_t first_leaf,
                                                            int64_t B, int64_t n_uniform,
                                                            const int64_t* uni_slot,  // (no __restrict__ on these two: a restrict-qualified
                                                            const double* u,          //  fetch is free to sink below the barrier, behind the staging)
                                                            int64_t counter, double usp,
                                                            double beta, int64_t* __restrict__ idx_out,
                                                            double* __restrict__ prio_ws, double* __restrict__ w_ws,
                                                            double* __restrict__ partial, int fuse_norm,
                                                            double* __restrict__ w64, float* __restrict__ w32,
                                                            double* __restrict__ stats) {
  __shared__ double s_red[16];
  // The top ten levels of the tree (nodes 0 .. 1022, 8 KB) go to LDS in ONE batch of fetches, together with the root and this
  // thread's first draw (device-mapped host memory: a PCIe read): the descent is a chain of dependent reads, 20 levels at N = 2^20,
  // and its first ten now cost an LDS read each instead of an L2 round trip (round 5; 10.3 us at B = 32 before).  Same values, same
  // comparisons: the indices are the reference's bit for bit.
  constexpr int NC = 1023;
  __shared__ double s_top[NC + 1];
  const int64_t tree_size = 2 * first_leaf + 1;
  const int64_t b_first = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool first_draws = b_first < B && b_first >= n_uniform;
  const double u_first = u[first_draws ? b_first - n_uniform : 0];
  const int64_t slot_first = uni_slot[(b_first < B && b_first < n_uniform) ? b_first : 0];
  {
    double tv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = threadIdx.x + 256 * q;
      tv[q] = tree[i < tree_size ? i : 0];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s_top[threadIdx.x + 256 * q] = tv[q];
  }
  __syncthreads();
  const double root = s_top[0];
  const double uniform_probs = 1.0 / (double)counter;
  double my_w = 0.0, my_p = 0.0;
  for (int64_t b = b_first; b < B; b += (int64_t)gridDim.x * 256) {
    int64_t index;
    if (b < n_uniform) {
      index = (b == b_first ? slot_first : uni_slot[b]) + first_leaf;  // per_buffer.py:75-78
    } else {
      double num = (b == b_first ? u_first : u[b - n_uniform]) * root;  // per_buffer.py:81
      index = 0;
      while (index < first_leaf && 2 * index + 1 < NC) {  // per_buffer.py:56-68, levels held in LDS
        const int64_t left = 2 * index + 1;
        const double l = s_top[left];
        if (num <= l) index = left;
        else { num -= l; index = left + 1; }
      }
      while (index < first_leaf) {
        const int64_t left = 2 * index + 1;
        const double l = tree[left];
        if (num <= l) index = left;
        else { num -= l; index = left + 1; }
      }
    }
    const double p = tree[index];
    const double prioritized = p / root;
    const double sample_prob = (1.0 - usp) * prioritized + usp * uniform_probs;  // per_buffer.py:91-92
    const double w = pow(uniform_probs / sample_prob, beta);
    idx_out[b] = index;
    prio_ws[b] = p;
    w_ws[b] = w;
    my_w = fmax(my_w, w);
    my_p += p;
  }
  const double bw = jh_block_reduce(my_w, s_red, JhMax(), 0.0);
  const double bp = jh_block_reduce(my_p, s_red, JhAdd(), 0.0);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = bw;
    partial[2 * blockIdx.x + 1] = bp;
  }
  if (fuse_norm) {
    // B <= 256: ONE workgroup holds the whole batch, a thread's only sample is still in its registers -- the
    // normalisation of jh_per_norm_kernel right here (same values: its max / sum over ONE partial are bw / bp), one
    // launch less in front of every learn() of the B = 32 configurations
    if (threadIdx.x < B) {
      const double w = my_w / bw;  // per_buffer.py:94  (my_w = fmax(0, w) = w: the weights are positive)
      if (w64) w64[threadIdx.x] = w;
      if (w32) w32[threadIdx.x] = (float)w;
    }
    if (threadIdx.x == 0 && stats) {
      stats[0] = bp / (double)B;
      stats[1] = root / (double)counter;
      stats[2] = root;
      stats[3] = bw;
    }
  }
}

__global__ void __launch_bounds__(256) jh_per_norm_kernel(const double* __restrict__ tree, int64_t B, int nb_partial,
                                                          const double* __restrict__ partial,
                                                          const double* __restrict__ w_ws, int64_t counter,
                                                          double* __restrict__ w64, float* __restrict__ w32,
                                                          double* __restrict__ stats) {
  __shared__ double s_red[16];
  double mw = 0.0, sp = 0.0;
  for (int i = threadIdx.x; i < nb_partial; i += 256) {
    mw = fmax(mw, partial[2 * i]);
    sp += partial[2 * i + 1];
  }
  mw = jh_block_reduce(mw, s_red, JhMax(), 0.0);
  sp = jh_block_reduce(sp, s_red, JhAdd(), 0.0);
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
    const double w = w_ws[b] / mw;  // per_buffer.py:94
    if (w64) w64[b] = w;
    if (w32) w32[b] = (float)w;  // torch.FloatTensor(weights)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats) {
    const double root = tree[0];
    stats[0] = sp / (double)B;           // sampled_p  per_buffer.py:99
    stats[1] = root / (double)counter;   // mean_p     per_buffer.py:100
    stats[2] = root;
    stats[3] = mw;
  }
}

__global__ void jh_per_init_kernel(double* maxp) { *maxp = 1.0; }

// ----------------------------------------------------------------------------- host API
JH_EXPORT int jh_per_create(jh_ctx* ctx, int64_t capacity, double usp, jh_per** out) {
  JH_ARG(ctx && out);
  JH_ARG(capacity > 0 && capacity < ((int64_t)1 << 40));
  JH_HIP(hipSetDevice(ctx->device));
  jh_per* p = new jh_per();
  p->ctx = ctx;
  p->N = capacity;
  p->tree_size = 2 * capacity - 1;
  p->usp = usp;
  p->tree_index = capacity - 1;
  p->depth_max = 63 - __builtin_clzll((unsigned long long)(p->tree_size));  // depth(2N-2) = floor(log2(2N-1))
  p->deep_first = ((int64_t)1 << p->depth_max) - 1;
  hipError_t e = hipMalloc((void**)&p->tree, sizeof(double) * (size_t)p->tree_size);
  if (e != hipSuccess) {
    delete p;
    return jh_fail(JH_ERR_NOMEM, "hipMalloc of the sum tree (%lld doubles) failed: %s", (long long)p->tree_size,
                   hipGetErrorString(e));
  }
  JH_HIP(hipMemset(p->tree, 0, sizeof(double) * (size_t)p->tree_size));
  JH_HIP(hipMalloc((void**)&p->maxp, sizeof(double)));
  JH_HIP(hipMalloc((void**)&p->ws.delta, sizeof(double) * kChunk));
  JH_HIP(hipMalloc((void**)&p->ws.partial, sizeof(double) * 2 * kMaxBlocks));
  JH_LAUNCH(jh_per_init_kernel, dim3(1), dim3(1), 0, 0, p->maxp);
  JH_LAUNCH_CHECK();
  JH_HIP(hipDeviceSynchronize());
  *out = p;
  return JH_OK;
}

JH_EXPORT void jh_per_destroy(jh_per* p) {
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(p->tree);
  (void)hipFree(p->maxp);
  (void)hipFree(p->ws.delta);
  (void)hipFree(p->ws.partial);
  if (p->ws.prio) (void)hipFree(p->ws.prio);
  if (p->ws.w) (void)hipFree(p->ws.w);
  delete p;
}

static PerDeltaArgs per_delta_args(jh_per* p, const int64_t* d_idx, int64_t push_start, const void* prio, int prio_dt, int mode) {
  PerDeltaArgs a{};
  a.tree = p->tree; a.maxp = p->maxp; a.idx = d_idx; a.push_start = push_start; a.prio = prio; a.prio_dt = prio_dt; a.mode = mode;
  a.tree_size = p->tree_size; a.first_leaf = p->N - 1; a.delta_out = p->ws.delta;
  return a;
}

static int per_climb(jh_per* p, int B, const int64_t* d_idx, int64_t push_start, int mode, hipStream_t st) {
  if (p->depth_max > 0) {
    JH_LAUNCH(jh_per_climb_kernel, dim3(p->depth_max), dim3(256), 0, st, p->tree, B, d_idx, push_start,
                       p->ws.delta, mode, p->tree_size, p->N - 1);
    JH_LAUNCH_CHECK();
  }
  return JH_OK;
}

// leaves and climb in one launch (jh_fused.h: jh_per_climb_small)?  JH_PER_FUSED_CLIMB=1; OFF by default: measured on Rainbow's loop
// (profiles/r06_ab_per_fused_climb.txt: 3 638-3 768 updates/s with the two launches, 3 610-3 623 fused) it buys nothing -- the write-back
// runs on a side stream beside the backward, and five serial rounds of four depth-waves are no shorter than twenty depth-workgroups
bool jh_per_small(const jh_per* p, int B) {
  static const bool on = getenv("JH_PER_FUSED_CLIMB") && atoi(getenv("JH_PER_FUSED_CLIMB")) == 1;
  return on && B <= kPerSmall && p->depth_max > 0;
}

static int per_apply(jh_per* p, int B, const int64_t* d_idx, int64_t push_start, const void* prio, int prio_dt, int mode,
                     hipStream_t st) {
  PerDeltaArgs a = per_delta_args(p, d_idx, push_start, prio, prio_dt, mode);
  const bool small = jh_per_small(p, B);
  a.climb_depth = small ? p->depth_max : 0;
  JH_LAUNCH(jh_per_delta_kernel, dim3(1), dim3(256), 0, st, a, B);
  JH_LAUNCH_CHECK();
  return small ? JH_OK : per_climb(p, B, d_idx, push_start, mode, st);
}

// the two halves for a caller that folds the leaf write-back into a kernel of its own (jh_fused.h)
int jh_per_delta_args(jh_per* p, int B, const int64_t* d_idx, const void* d_prio, int prio_dt, PerDeltaArgs* out) {
  JH_ARG(p && d_idx && d_prio && out);
  JH_ARG(B > 0 && B <= kChunk && (prio_dt == JH_F32 || prio_dt == JH_F64));
  *out = per_delta_args(p, d_idx, 0, d_prio, prio_dt, 0);
  out->climb_depth = jh_per_small(p, B) ? p->depth_max : 0;  // small batches: the caller's launch climbs too, jh_per_climb below is a no-op
  return JH_OK;
}
int jh_per_climb(jh_per* p, int B, const int64_t* d_idx, hipStream_t st) { return jh_per_small(p, B) ? JH_OK : per_climb(p, B, d_idx, 0, 0, st); }

JH_EXPORT int jh_per_push(jh_per* p, int64_t n, const double* h_prio, jh_stream stream) {
  JH_ARG(p != nullptr && n >= 0);
  hipStream_t st = jh_s(stream);
  int64_t done = 0;
  while (done < n) {
    // a segment never wraps (2N-2 -> N-1), never crosses the depth boundary (2^D-2 -> 2^D-1)
    // and holds at most kChunk leaves: inside it every node's items are contiguous.
    int64_t seg = n - done;
    if (seg > kChunk) seg = kChunk;
    if (p->tree_index + seg > p->tree_size) seg = p->tree_size - p->tree_index;
    if (p->tree_index < p->deep_first && p->tree_index + seg > p->deep_first) seg = p->deep_first - p->tree_index;
    const void* prio_dev = nullptr;
    jh_pinned_slab* slab = nullptr;
    if (h_prio) {
      int rc = jh_ctx_slab(p->ctx, sizeof(double) * (size_t)seg, &slab);
      if (rc) return rc;
      memcpy(slab->host, h_prio + done, sizeof(double) * (size_t)seg);
      prio_dev = slab->dev;
    }
    int rc = per_apply(p, (int)seg, nullptr, p->tree_index, prio_dev, JH_F64, PER_DISTINCT | PER_CONTIG, st);
    if (rc) return rc;
    if (slab) {
      rc = jh_ctx_slab_release(p->ctx, slab, st);
      if (rc) return rc;
    }
    p->tree_index += seg;
    if (p->tree_index == p->tree_size) p->tree_index = p->N - 1;  // per_buffer.py:38-40
    p->counter = p->counter + seg < p->N ? p->counter + seg : p->N;
    done += seg;
  }
  return JH_OK;
}

// jh_per_push for priorities that are already in HBM (a device-side producer: jh_feed_tick's actor-side priorities).
JH_EXPORT int jh_per_push_device(jh_per* p, int64_t n, const double* d_prio, jh_stream stream) {
  JH_ARG(p != nullptr && n >= 0 && d_prio != nullptr);
  hipStream_t st = jh_s(stream);
  int64_t done = 0;
  while (done < n) {  // same segmentation as jh_per_push
    int64_t seg = n - done;
    if (seg > kChunk) seg = kChunk;
    if (p->tree_index + seg > p->tree_size) seg = p->tree_size - p->tree_index;
    if (p->tree_index < p->deep_first && p->tree_index + seg > p->deep_first) seg = p->deep_first - p->tree_index;
    int rc = per_apply(p, (int)seg, nullptr, p->tree_index, d_prio + done, JH_F64, PER_DISTINCT | PER_CONTIG, st);
    if (rc) return rc;
    p->tree_index += seg;
    if (p->tree_index == p->tree_size) p->tree_index = p->N - 1;  // per_buffer.py:38-40
    p->counter = p->counter + seg < p->N ? p->counter + seg : p->N;
    done += seg;
  }
  return JH_OK;
}

JH_EXPORT int jh_per_update(jh_per* p, int64_t B, const int64_t* d_idx, const void* d_prio, int32_t prio_dtype,
                            jh_stream stream) {
  JH_ARG(p && d_idx && d_prio);
  JH_ARG(B >= 0);
  JH_ARG(prio_dtype == JH_F32 || prio_dtype == JH_F64);
  const size_t es = prio_dtype == JH_F32 ? 4 : 8;
  for (int64_t done = 0; done < B; done += kChunk) {
    const int seg = (int)(B - done < kChunk ? B - done : kChunk);
    int rc = per_apply(p, seg, d_idx + done, 0, (const char*)d_prio + es * (size_t)done, prio_dtype, 0, jh_s(stream));
    if (rc) return rc;
  }
  return JH_OK;
}

JH_EXPORT int jh_per_sample(jh_per* p, int64_t B, double beta, int64_t n_uniform, const int64_t* h_uniform_slot,
                            const double* h_u, int64_t* d_idx, double* d_w64, float* d_w32, double* d_stats,
                            jh_stream stream) {
  JH_ARG(p && d_idx);
  JH_ARG(B > 0 && n_uniform >= 0 && n_uniform <= B);
  JH_ARG(n_uniform == 0 || h_uniform_slot);
  JH_ARG(n_uniform == B || h_u);
  if (p->counter <= 0) return jh_fail(JH_ERR_STATE, "jh_per_sample on an empty buffer (per_buffer.py:71 asserts root > 0)");
  hipStream_t st = jh_s(stream);
  if (p->ws.ws_cap < B) {
    JH_HIP(hipStreamSynchronize(st));
    if (p->ws.prio) JH_HIP(hipFree(p->ws.prio));
    if (p->ws.w) JH_HIP(hipFree(p->ws.w));
    int64_t cap = 1024;
    while (cap < B) cap <<= 1;
    JH_HIP(hipMalloc((void**)&p->ws.prio, sizeof(double) * (size_t)cap));
    JH_HIP(hipMalloc((void**)&p->ws.w, sizeof(double) * (size_t)cap));
    p->ws.ws_cap = cap;
  }
  // RNG draws: written into a pinned, device-mapped slab that the kernel reads in place
  // (B*8 bytes: one PCIe read burst, cheaper than a separate DMA for B = 32..512)
  jh_pinned_slab* slab = nullptr;
  const size_t off_u = (sizeof(int64_t) * (size_t)n_uniform + 255) & ~(size_t)255;
  int rc = jh_ctx_slab(p->ctx, off_u + sizeof(double) * (size_t)(B - n_uniform) + 256, &slab);
  if (rc) return rc;
  if (n_uniform) memcpy(slab->host, h_uniform_slot, sizeof(int64_t) * (size_t)n_uniform);
  if (B - n_uniform) memcpy((char*)slab->host + off_u, h_u, sizeof(double) * (size_t)(B - n_uniform));
  int64_t nbl = (B + 255) / 256;
  const int nb = (int)(nbl < kMaxBlocks ? nbl : kMaxBlocks);
  const int fuse = B <= 256 ? 1 : 0;
  JH_LAUNCH(jh_per_sample_kernel, dim3(nb), dim3(256), 0, st, p->tree, p->N - 1, B, n_uniform,
                     (const int64_t*)slab->dev, (const double*)((char*)slab->dev + off_u), p->counter, p->usp, beta,
                     d_idx, p->ws.prio, p->ws.w, p->ws.partial, fuse, d_w64, d_w32, d_stats);
  JH_LAUNCH_CHECK();
  if (!fuse) {
    JH_LAUNCH(jh_per_norm_kernel, dim3(nb), dim3(256), 0, st, p->tree, B, nb, p->ws.partial, p->ws.w,
                       p->counter, d_w64, d_w32, d_stats);
    JH_LAUNCH_CHECK();
  }
  return jh_ctx_slab_release(p->ctx, slab, st);
}

JH_EXPORT int jh_per_state(jh_per* p, double* max_priority, double* root, int64_t* tree_index, int64_t* counter,
                           jh_stream stream) {
  JH_ARG(p != nullptr);
  hipStream_t st = jh_s(stream);
  if (max_priority) JH_HIP(hipMemcpyAsync(max_priority, p->maxp, sizeof(double), hipMemcpyDeviceToHost, st));
  if (root) JH_HIP(hipMemcpyAsync(root, p->tree, sizeof(double), hipMemcpyDeviceToHost, st));
  JH_HIP(hipStreamSynchronize(st));
  if (tree_index) *tree_index = p->tree_index;
  if (counter) *counter = p->counter;
  return JH_OK;
}

JH_EXPORT int jh_per_load(jh_per* p, const double* h_tree, double max_priority, int64_t tree_index, int64_t counter) {
  JH_ARG(p && h_tree);
  JH_ARG(tree_index >= p->N - 1 && tree_index < p->tree_size && counter >= 0 && counter <= p->N);
  JH_HIP(hipDeviceSynchronize());
  JH_HIP(hipMemcpy(p->tree, h_tree, sizeof(double) * (size_t)p->tree_size, hipMemcpyHostToDevice));
  JH_HIP(hipMemcpy(p->maxp, &max_priority, sizeof(double), hipMemcpyHostToDevice));
  p->tree_index = tree_index;
  p->counter = counter;
  return JH_OK;
}

JH_EXPORT int jh_per_dump(jh_per* p, double* h_tree, jh_stream stream) {
  JH_ARG(p && h_tree);
  JH_HIP(hipStreamSynchronize(jh_s(stream)));
  JH_HIP(hipMemcpy(h_tree, p->tree, sizeof(double) * (size_t)p->tree_size, hipMemcpyDeviceToHost));
  return JH_OK;
}

JH_EXPORT double* jh_per_tree_ptr(jh_per* p) { return p ? p->tree : nullptr; }
JH_EXPORT int64_t jh_per_tree_size(const jh_per* p) { return p ? p->tree_size : -1; }

// ------------------------------------------------------------------------------ sharded PER (data-parallel learners)
// Every rank owns a shard (own tree) of one logical buffer; sampling is local, the IS weights are those of the single
// logical tree (jorldy_amd/parallel.py: sharded_is_weights is the reference form of this math):
//   jh_per_shard_stats      d_out3 = {root_g, count_g, min priority of the LAST sample}   -> all-gathered over the ranks
//   jh_per_weights_sharded  w_i = ((1 / COUNT) / ((1 - usp) p_i / ROOT + usp / COUNT))^beta / w(min p over all ranks),
//                           ROOT / COUNT = sums over the G gathered triples (per_buffer.py:88-94 on the logical buffer)
__global__ void __launch_bounds__(256) jh_per_shard_stats_kernel(const double* __restrict__ tree, int64_t counter, int64_t B,
                                                                 const double* __restrict__ prio, double* __restrict__ out3) {
  __shared__ double s_red[16];
  double mn = 1.7976931348623157e308;
  for (int64_t b = threadIdx.x; b < B; b += 256) mn = fmin(mn, prio[b]);
  mn = jh_block_reduce(mn, s_red, JhMin(), 1.7976931348623157e308);
  if (threadIdx.x == 0) {
    out3[0] = tree[0];
    out3[1] = (double)counter;
    out3[2] = mn;
  }
}

__global__ void __launch_bounds__(256) jh_per_weights_sharded_kernel(int64_t B, double beta, double usp, const double* __restrict__ prio,
                                                                     const double* __restrict__ all3, int G, double* __restrict__ w64,
                                                                     float* __restrict__ w32) {
  double root = 0.0, count = 0.0, min_p = 1.7976931348623157e308;
  for (int g = 0; g < G; ++g) {  // rank order: the same sums on every rank
    root += all3[3 * g];
    count += all3[3 * g + 1];
    min_p = fmin(min_p, all3[3 * g + 2]);
  }
  const double uni = 1.0 / count;
  const double w_max = pow(uni / ((1.0 - usp) * (min_p / root) + usp * uni), beta);
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
    const double w = pow(uni / ((1.0 - usp) * (prio[b] / root) + usp * uni), beta) / w_max;
    if (w64) w64[b] = w;
    if (w32) w32[b] = (float)w;
  }
}

JH_EXPORT int jh_per_shard_stats(jh_per* p, int64_t B, double* d_out3, jh_stream stream) {
  JH_ARG(p && d_out3 && B > 0 && B <= p->ws.ws_cap);
  JH_LAUNCH(jh_per_shard_stats_kernel, dim3(1), dim3(256), 0, jh_s(stream), p->tree, p->counter, B, p->ws.prio, d_out3);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

JH_EXPORT int jh_per_weights_sharded(jh_per* p, int64_t B, double beta, const double* d_all3, int32_t n_shards, double* d_w64,
                                     float* d_w32, jh_stream stream) {
  JH_ARG(p && d_all3 && n_shards > 0 && B > 0 && B <= p->ws.ws_cap && (d_w64 || d_w32));
  int64_t nbl = (B + 255) / 256;
  JH_LAUNCH(jh_per_weights_sharded_kernel, dim3((unsigned)(nbl < 64 ? nbl : 64)), dim3(256), 0, jh_s(stream), B, beta, p->usp, p->ws.prio,
            d_all3, n_shards, d_w64, d_w32);
  JH_LAUNCH_CHECK();
  return JH_OK;
}
