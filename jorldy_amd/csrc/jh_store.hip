// GPU-resident SoA transition store: ring push via pinned staging + async H2D, and a fused
// multi-column gather that also performs the fp32 cast of BaseAgent.as_tensor.
// Replaces core/buffer/replay_buffer.py:8-35, rollout_buffer.py:6-24, base.py:42-56.
#include <stdlib.h>

#include "jh_common.h"


JH_EXPORT int jh_store_create(jh_ctx* ctx, int64_t capacity, int32_t n_cols, const jh_col_desc* cols, jh_store** out) {
  JH_ARG(ctx && out && cols);
  JH_ARG(capacity > 0 && n_cols > 0 && n_cols <= 16);
  JH_HIP(hipSetDevice(ctx->device));
  jh_store* s = new jh_store();
  s->ctx = ctx;
  s->capacity = capacity;
  s->n_cols = n_cols;
  for (int c = 0; c < n_cols; ++c) {
    size_t es = jh_dtype_size(cols[c].dtype);
    if (es == 0 || cols[c].elems <= 0) {
      delete s;
      return jh_fail(JH_ERR_ARG, "column %d: bad dtype %d / elems %lld", c, cols[c].dtype, (long long)cols[c].elems);
    }
    s->cols.push_back(cols[c]);
    s->row_bytes.push_back(es * (size_t)cols[c].elems);
  }
  for (int c = 0; c < n_cols; ++c) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, s->row_bytes[c] * (size_t)capacity);
    if (e != hipSuccess) {
      for (void* q : s->dev) (void)hipFree(q);
      delete s;
      return jh_fail(JH_ERR_NOMEM, "hipMalloc of column %d (%zu bytes) failed: %s", c, s->row_bytes[c] * (size_t)capacity,
                     hipGetErrorString(e));
    }
    s->dev.push_back(p);
  }
  s->staged_off.resize(n_cols);
  *out = s;
  return JH_OK;
}

JH_EXPORT void jh_store_destroy(jh_store* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  (void)hipDeviceSynchronize();
  if (s->staged) (void)jh_ctx_slab_release(s->ctx, s->staged, nullptr);  // a begun, never committed stage: hand the slab back
  for (void* p : s->dev) (void)hipFree(p);
  delete s;
}

// ONE launch for a ring append of all columns (<= 8) from sources a kernel can read: device memory, or device-mapped
// pinned memory for small appends.  A rollout commit used to be one hipMemcpyAsync per column (5-10 SDMA copies of a few
// KB each, ~3 us apiece on the host and again on the copy engine, back to back in front of learn()).
constexpr int kCopyJobs = 14;  // store columns (<= 8) + extra plain copies riding in the same launch (the collector's captured heads / values)
struct CopyCols {
  const char* src[kCopyJobs];
  char* dst[kCopyJobs];       // ring position of the first row
  char* dst_wrap[kCopyJobs];  // column base (rows behind the ring's wrap)
  int64_t first[kCopyJobs], total[kCopyJobs];  // bytes before the wrap / in all
  const unsigned* gate;  // optional flag word (device-mapped pinned memory): every workgroup waits until it holds gate_val
  unsigned gate_val;
};
__global__ void __launch_bounds__(256) jh_store_copy_cols_kernel(CopyCols a) {
  if (a.gate) {
    // A launch enqueued BEFORE its sources are complete (jh_collector_begin): lane 0 polls the host's release store (bounded: ~2 s;
    // the launch sits behind the acting kernel in stream order, so in practice the flag arrives within a microsecond or two), then
    // the sources -- fine-grained host memory, never cached on the device -- are read.
    // The host ABORTS a run by storing gate_val | 0x80000000 (jh_collector_loop after an error: half-filled staging rows must not reach
    // the store): the launch then copies nothing.  On a timeout nothing is copied either and gate[1] carries the run's tag: the stream
    // drains, the collector's next run reports the failed one (jh_collect.hip: run_prepare).
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
      int ok = 0;
      for (long spin = 0; spin < 4000000L; ++spin) {
        const unsigned v = __hip_atomic_load(a.gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == a.gate_val) { ok = 1; break; }
        if (v == (a.gate_val | 0x80000000u)) break;
        __builtin_amdgcn_s_sleep(8);
      }
      s_ok = ok;
      // a TIMEOUT (not an abort) is reported: gate[1] <- the run's tag, in the same device-mapped page; the collector's next run fails
      // with it instead of training on rows that never arrived
      if (!ok && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (a.gate_val | 0x80000000u))
        __hip_atomic_store(const_cast<unsigned*>(a.gate) + 1, a.gate_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (!s_ok) return;
  }
  const int c = blockIdx.y;
  const char* src = a.src[c];
  char* d0 = a.dst[c];
  char* d1 = a.dst_wrap[c];
  const int64_t first = a.first[c], total = a.total[c];
  const bool vec = ((((uintptr_t)src | (uintptr_t)d0 | (uintptr_t)d1) & 15) == 0) && ((first & 15) == 0);
  const int64_t nv = vec ? total >> 4 : 0;
  // (one 16-byte piece per thread for every commit below 8 MB per column -- see the grid in store_append_kernel: the sources are
  // device-mapped HOST memory for every collector / single-transition commit, and with 64 bytes per thread this loop was four PCIe
  // round trips in series, 6-9 us per commit; round 5)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const int64_t o = i << 4;
    const uint4 v = *reinterpret_cast<const uint4*>(src + o);
    *reinterpret_cast<uint4*>(o < first ? d0 + o : d1 + (o - first)) = v;
  }
  for (int64_t o = (nv << 4) + (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256)
    *(o < first ? d0 + o : d1 + (o - first)) = src[o];
}

// `cols`: device-visible sources.  Advances the ring like jh_store_append.  n_extra plain copies (x_src -> x_dst, x_bytes) ride along.
static int store_append_kernel(jh_store* s, int64_t n, const void* const* cols, hipStream_t st, int n_extra = 0, const void* const* x_src = nullptr,
                               void* const* x_dst = nullptr, const int64_t* x_bytes = nullptr, const unsigned* gate = nullptr, unsigned gate_val = 0) {
  const int64_t first = s->capacity - s->index < n ? s->capacity - s->index : n;
  CopyCols a{};
  a.gate = gate; a.gate_val = gate_val;
  size_t most = 0;
  for (int e = 0; e < n_extra; ++e) {
    const int c = s->n_cols + e;
    a.src[c] = (const char*)x_src[e];
    a.dst[c] = a.dst_wrap[c] = (char*)x_dst[e];
    a.first[c] = a.total[c] = x_bytes[e];
    if ((size_t)x_bytes[e] > most) most = (size_t)x_bytes[e];
  }
  for (int c = 0; c < s->n_cols; ++c) {
    const size_t rb = s->row_bytes[c];
    a.src[c] = (const char*)cols[c];
    a.dst[c] = (char*)s->dev[c] + rb * (size_t)s->index;
    a.dst_wrap[c] = (char*)s->dev[c];
    a.first[c] = (int64_t)(rb * (size_t)first);
    a.total[c] = (int64_t)(rb * (size_t)n);
    if (rb * (size_t)n > most) most = rb * (size_t)n;
  }
  unsigned gx = (unsigned)((most + 4095) / 4096);  // 16 bytes per thread: one round for every commit below 8 MB per column
  if (gx < 1) gx = 1;
  if (gx > 2048) gx = 2048;
  JH_LAUNCH(jh_store_copy_cols_kernel, dim3(gx, (unsigned)(s->n_cols + n_extra)), dim3(256), 0, st, a);
  JH_LAUNCH_CHECK();
  s->index = (s->index + n) % s->capacity;
  s->counter = s->counter + n < s->capacity ? s->counter + n : s->capacity;
  return JH_OK;
}

JH_EXPORT int jh_store_stage_begin(jh_store* s, int64_t n, void** h_cols_out) {
  JH_ARG(s && h_cols_out);
  JH_ARG(n > 0 && n <= s->capacity);
  if (s->staged) return jh_fail(JH_ERR_STATE, "jh_store_stage_begin called twice without commit");
  size_t total = 0;
  for (int c = 0; c < s->n_cols; ++c) {
    s->staged_off[c] = total;
    total += (s->row_bytes[c] * (size_t)n + 255) & ~(size_t)255;
  }
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(s->ctx, total, &slab);
  if (rc) return rc;
  s->staged = slab;
  s->staged_n = n;
  for (int c = 0; c < s->n_cols; ++c) h_cols_out[c] = (char*)slab->host + s->staged_off[c];
  return JH_OK;
}

// Drop what was staged: the slab goes back, the ring does not move (internal: a collector run that failed half way must not append its
// half-written rows -- ADVICE r5).
int jh_store_stage_abort(jh_store* s, hipStream_t st) {
  if (!s || !s->staged) return JH_OK;
  const int rc = jh_ctx_slab_release(s->ctx, s->staged, st);
  s->staged = nullptr;
  return rc;
}

JH_EXPORT int jh_store_stage_commit(jh_store* s, jh_stream stream) { return jh_store_stage_commit_extra(s, 0, nullptr, nullptr, nullptr, jh_s(stream)); }

// The commit with n_extra (<= 4) plain device-visible -> device copies in the SAME launch (internal: the collector's captured
// heads / values land next to the rollout rows without further launches or SDMA copies).
int jh_store_stage_commit_extra(jh_store* s, int n_extra, const void* const* x_src, void* const* x_dst, const int64_t* x_bytes, hipStream_t st) {
  return jh_store_stage_commit_gated(s, n_extra, x_src, x_dst, x_bytes, nullptr, 0, st);
}

static size_t store_kernel_commit_max() {
  static const size_t v = getenv("JH_STORE_KERNEL_COMMIT_MAX") ? (size_t)atoll(getenv("JH_STORE_KERNEL_COMMIT_MAX")) : ((size_t)512 << 10);
  return v;
}
// Does a commit of n staged rows take the one-launch form (the only one that can be gated)?
bool jh_store_commit_is_one_launch(const jh_store* s, int64_t n) {
  size_t bytes = 0;
  for (int c = 0; c < s->n_cols; ++c) bytes += s->row_bytes[c] * (size_t)n;
  return s->n_cols <= 8 && bytes <= store_kernel_commit_max();
}

// ... and optionally gated: the launch waits for *gate == gate_val (device-mapped pinned word) before it reads its sources, so it can be
// enqueued before the host has filled them.  Only the one-launch form can wait: bigger commits are refused when a gate is given.
int jh_store_stage_commit_gated(jh_store* s, int n_extra, const void* const* x_src, void* const* x_dst, const int64_t* x_bytes, const unsigned* gate,
                                unsigned gate_val, hipStream_t st) {
  JH_ARG(s != nullptr && n_extra >= 0 && s->n_cols + n_extra <= kCopyJobs);
  if (!s->staged) return jh_fail(JH_ERR_STATE, "jh_store_stage_commit without begin");
  const int64_t n = s->staged_n;
  size_t bytes = 0;
  for (int c = 0; c < s->n_cols; ++c) bytes += s->row_bytes[c] * (size_t)n;
  const size_t kKernelCommitMax = store_kernel_commit_max();
  if (s->n_cols <= 8 && bytes <= kKernelCommitMax) {
    // small commits (a PPO rollout: 46 KB; Rainbow's 4 deferred rows: 226 KB): one kernel reads the slab in place
    const void* src[8];
    for (int c = 0; c < s->n_cols; ++c) src[c] = (const char*)s->staged->dev + s->staged_off[c];
    int rc = store_append_kernel(s, n, src, st, n_extra, x_src, x_dst, x_bytes, gate, gate_val);
    int rc2 = jh_ctx_slab_release(s->ctx, s->staged, st);
    s->staged = nullptr;
    return rc ? rc : rc2;
  }
  if (gate) return jh_fail(JH_ERR_ARG, "a gated commit needs the one-launch form (<= %zu bytes of rows, <= 8 columns)", kKernelCommitMax);
  const int64_t first = s->capacity - s->index < n ? s->capacity - s->index : n;  // rows before the wrap
  for (int c = 0; c < s->n_cols; ++c) {
    const size_t rb = s->row_bytes[c];
    const char* src = (const char*)s->staged->host + s->staged_off[c];
    JH_HIP(hipMemcpyAsync((char*)s->dev[c] + rb * (size_t)s->index, src, rb * (size_t)first, hipMemcpyHostToDevice, st));
    if (first < n)
      JH_HIP(hipMemcpyAsync(s->dev[c], src + rb * (size_t)first, rb * (size_t)(n - first), hipMemcpyHostToDevice, st));
  }
  for (int e = 0; e < n_extra; ++e) JH_HIP(hipMemcpyAsync(x_dst[e], x_src[e], (size_t)x_bytes[e], hipMemcpyDefault, st));
  int rc = jh_ctx_slab_release(s->ctx, s->staged, st);
  s->staged = nullptr;
  if (rc) return rc;
  s->index = (s->index + n) % s->capacity;
  s->counter = s->counter + n < s->capacity ? s->counter + n : s->capacity;
  return JH_OK;
}

JH_EXPORT int jh_store_push(jh_store* s, int64_t n, const void* const* h_cols, jh_stream stream) {
  JH_ARG(s && h_cols);
  if (n == 0) return JH_OK;
  // a push longer than the ring keeps only the last `capacity` rows, like the reference's loop
  if (n > s->capacity) {
    // advance the ring as the reference would, then store the surviving tail
    const int64_t skip = n - s->capacity;
    std::vector<const void*> tail(s->n_cols);
    for (int c = 0; c < s->n_cols; ++c) tail[c] = (const char*)h_cols[c] + s->row_bytes[c] * (size_t)skip;
    s->index = (s->index + skip) % s->capacity;
    s->counter = s->capacity;
    return jh_store_push(s, s->capacity, tail.data(), stream);
  }
  std::vector<void*> dst(s->n_cols);
  int rc = jh_store_stage_begin(s, n, dst.data());
  if (rc) return rc;
  for (int c = 0; c < s->n_cols; ++c) memcpy(dst[c], h_cols[c], s->row_bytes[c] * (size_t)n);
  return jh_store_stage_commit(s, stream);
}

// ring append of n <= capacity rows from per-column sources the copy engine can read asynchronously
// (device memory, or pinned host memory that stays untouched until `st` reaches this point)
int jh_store_append(jh_store* s, int64_t n, const void* const* cols, hipMemcpyKind kind, hipStream_t st) {
  const int64_t first = s->capacity - s->index < n ? s->capacity - s->index : n;
  for (int c = 0; c < s->n_cols; ++c) {
    const size_t rb = s->row_bytes[c];
    const char* src = (const char*)cols[c];
    JH_HIP(hipMemcpyAsync((char*)s->dev[c] + rb * (size_t)s->index, src, rb * (size_t)first, kind, st));
    if (first < n) JH_HIP(hipMemcpyAsync(s->dev[c], src + rb * (size_t)first, rb * (size_t)(n - first), kind, st));
  }
  s->index = (s->index + n) % s->capacity;
  s->counter = s->counter + n < s->capacity ? s->counter + n : s->capacity;
  return JH_OK;
}

JH_EXPORT int jh_store_push_device(jh_store* s, int64_t n, const void* const* d_cols, jh_stream stream) {
  JH_ARG(s && d_cols);
  JH_ARG(n >= 0 && n <= s->capacity);
  if (n == 0) return JH_OK;
  if (s->n_cols <= 8) return store_append_kernel(s, n, d_cols, jh_s(stream));  // one launch instead of a copy per column
  return jh_store_append(s, n, d_cols, hipMemcpyDeviceToDevice, jh_s(stream));
}

// ------------------------------------------------------------------------------ positional row writes
// rows land at caller-chosen slots (frame pool of the de-duplicated Atari replay): the
// rows and their slot numbers are staged in one pinned, device-mapped slab that the kernel reads in place.
__global__ void __launch_bounds__(256) jh_scatter_rows_kernel(const char* __restrict__ src, const int64_t* __restrict__ slots, int64_t n, int64_t row_bytes,
                                                              char* __restrict__ dst, int64_t capacity) {
  const int64_t row = blockIdx.y;
  const int64_t slot = slots[row];
  if (slot < 0 || slot >= capacity) return;
  const char* s = src + row * row_bytes;
  char* d = dst + slot * row_bytes;
  if ((row_bytes & 15) == 0) {
    const int64_t nv = row_bytes >> 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256)
      reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row_bytes; i += (int64_t)gridDim.x * 256) d[i] = s[i];
  }
}

JH_EXPORT int jh_store_write_rows(jh_store* s, int64_t n, const int64_t* h_slots, const void* const* h_cols, jh_stream stream) {
  JH_ARG(s && h_slots && h_cols);
  JH_ARG(n >= 0 && n <= 65535);
  if (n == 0) return JH_OK;
  hipStream_t st = jh_s(stream);
  size_t total = ((sizeof(int64_t) * (size_t)n) + 255) & ~(size_t)255;
  std::vector<size_t> off(s->n_cols);
  for (int c = 0; c < s->n_cols; ++c) {
    off[c] = total;
    total += (s->row_bytes[c] * (size_t)n + 255) & ~(size_t)255;
  }
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(s->ctx, total, &slab);
  if (rc) return rc;
  memcpy(slab->host, h_slots, sizeof(int64_t) * (size_t)n);
  for (int c = 0; c < s->n_cols; ++c) memcpy((char*)slab->host + off[c], h_cols[c], s->row_bytes[c] * (size_t)n);
  for (int c = 0; c < s->n_cols; ++c) {
    const int64_t rb = (int64_t)s->row_bytes[c];
    unsigned gx = (unsigned)((rb / 16 + 255) / 256);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    JH_LAUNCH(jh_scatter_rows_kernel, dim3(gx, (unsigned)n), dim3(256), 0, st, (const char*)slab->dev + off[c], (const int64_t*)slab->dev, n, rb, (char*)s->dev[c],
              s->capacity);
    JH_LAUNCH_CHECK();
  }
  return jh_ctx_slab_release(s->ctx, slab, st);
}

// ------------------------------------------------------------------------------ gather
struct GatherCol {
  const void* src;
  void* dst;
  int64_t elems;
  int32_t src_dt;
  int32_t dst_dt;
};
struct GatherArgs {
  GatherCol col[16];
};

template <typename S>
__device__ __forceinline__ float to_f32(S v) { return (float)v; }

// Generic element-wise path over the flattened [B*elems] output: consecutive lanes write
// consecutive output elements (fully coalesced stores; loads are coalesced within a row).
template <typename S, typename D>
__device__ __forceinline__ void gather_rows(const S* __restrict__ src, D* __restrict__ dst, int64_t elems, int64_t B,
                                            const int64_t* __restrict__ idx, int64_t idx_off, int64_t cap) {
  const int64_t total = B * elems;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / elems, e = i - b * elems;
    int64_t r = idx[b] - idx_off;
    r = r < 0 ? 0 : (r >= cap ? cap - 1 : r);  // never read out of bounds on a bad index
    dst[i] = (D)src[r * elems + e];
  }
}

// uint8 rows whose byte length is a multiple of 16 (Atari frame stacks: 4*84*84 = 28224 = 16*1764):
// 16 B per lane loads; uint8 out -> one 16 B store, fp32 out -> four 16 B stores.
template <bool TO_F32>
__device__ __forceinline__ void gather_u8_vec(const uint8_t* __restrict__ src, void* __restrict__ dstv, int64_t elems,
                                              int64_t B, const int64_t* __restrict__ idx, int64_t idx_off, int64_t cap) {
  const int64_t vecs = elems / 16;
  const int64_t total = B * vecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / vecs, v = i - b * vecs;
    int64_t r = idx[b] - idx_off;
    r = r < 0 ? 0 : (r >= cap ? cap - 1 : r);
    const uint4 x = *reinterpret_cast<const uint4*>(src + r * elems + v * 16);
    if (!TO_F32) {
      *reinterpret_cast<uint4*>((uint8_t*)dstv + b * elems + v * 16) = x;
    } else {
      float* d = (float*)dstv + b * elems + v * 16;
      const unsigned wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float4 f;
        f.x = (float)(wds[k] & 0xff);
        f.y = (float)((wds[k] >> 8) & 0xff);
        f.z = (float)((wds[k] >> 16) & 0xff);
        f.w = (float)(wds[k] >> 24);
        *reinterpret_cast<float4*>(d + 4 * k) = f;
      }
    }
  }
}

__global__ void __launch_bounds__(256) jh_gather_kernel(GatherArgs a, int64_t B, const int64_t* __restrict__ idx,
                                                        int64_t idx_off, int64_t cap) {
  const GatherCol c = a.col[blockIdx.y];
  const int64_t elems = c.elems;
  if (c.src_dt == JH_U8) {
    const bool vec = (elems % 16) == 0;
    if (c.dst_dt == JH_U8) {
      if (vec) gather_u8_vec<false>((const uint8_t*)c.src, c.dst, elems, B, idx, idx_off, cap);
      else gather_rows((const uint8_t*)c.src, (uint8_t*)c.dst, elems, B, idx, idx_off, cap);
    } else {
      if (vec) gather_u8_vec<true>((const uint8_t*)c.src, c.dst, elems, B, idx, idx_off, cap);
      else gather_rows((const uint8_t*)c.src, (float*)c.dst, elems, B, idx, idx_off, cap);
    }
  } else if (c.src_dt == JH_F32) {
    gather_rows((const float*)c.src, (float*)c.dst, elems, B, idx, idx_off, cap);
  } else if (c.src_dt == JH_I64) {
    if (c.dst_dt == JH_I64) gather_rows((const int64_t*)c.src, (int64_t*)c.dst, elems, B, idx, idx_off, cap);
    else gather_rows((const int64_t*)c.src, (float*)c.dst, elems, B, idx, idx_off, cap);
  } else if (c.src_dt == JH_F64) {
    if (c.dst_dt == JH_F64) gather_rows((const double*)c.src, (double*)c.dst, elems, B, idx, idx_off, cap);
    else gather_rows((const double*)c.src, (float*)c.dst, elems, B, idx, idx_off, cap);
  } else if (c.src_dt == JH_I32) {
    if (c.dst_dt == JH_I32) gather_rows((const int32_t*)c.src, (int32_t*)c.dst, elems, B, idx, idx_off, cap);
    else gather_rows((const int32_t*)c.src, (float*)c.dst, elems, B, idx, idx_off, cap);
  }
}

JH_EXPORT int jh_store_gather(jh_store* s, int64_t B, const int64_t* d_idx, int64_t idx_offset, int32_t n_sel,
                              const int32_t* sel_cols, void* const* d_out, const int32_t* out_dtype, jh_stream stream) {
  JH_ARG(s && d_idx && sel_cols && d_out && out_dtype);
  JH_ARG(n_sel > 0 && n_sel <= 16);
  if (B == 0) return JH_OK;
  JH_ARG(B > 0);
  GatherArgs a;
  int64_t max_items = 1;
  for (int i = 0; i < n_sel; ++i) {
    const int c = sel_cols[i];
    JH_ARG(c >= 0 && c < s->n_cols);
    const int sdt = s->cols[c].dtype, ddt = out_dtype[i];
    if (!(ddt == sdt || ddt == JH_F32)) return jh_fail(JH_ERR_ARG, "gather: column %d out dtype %d unsupported (stored %d)", c, ddt, sdt);
    a.col[i] = GatherCol{s->dev[c], d_out[i], s->cols[c].elems, sdt, ddt};
    const int64_t per = (sdt == JH_U8 && s->cols[c].elems % 16 == 0) ? s->cols[c].elems / 16 : s->cols[c].elems;
    const int64_t items = (B * per + 255) / 256;
    if (items > max_items) max_items = items;
  }
  // memory-bound: cap at 256 CUs x 8 blocks and grid-stride (guide G11)
  const unsigned gx = (unsigned)(max_items < 2048 ? max_items : 2048);
  JH_LAUNCH(jh_gather_kernel, dim3(gx, n_sel), dim3(256), 0, jh_s(stream), a, B, d_idx, idx_offset, s->capacity);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

JH_EXPORT void* jh_store_col_ptr(jh_store* s, int32_t col) {
  if (!s || col < 0 || col >= s->n_cols) return nullptr;
  return s->dev[col];
}
JH_EXPORT int64_t jh_store_size(const jh_store* s) { return s ? s->counter : -1; }
JH_EXPORT int64_t jh_store_index(const jh_store* s) { return s ? s->index : -1; }
JH_EXPORT int64_t jh_store_capacity(const jh_store* s) { return s ? s->capacity : -1; }
JH_EXPORT int jh_store_set_position(jh_store* s, int64_t index, int64_t counter) {
  JH_ARG(s != nullptr);
  JH_ARG(index >= 0 && index < s->capacity && counter >= 0 && counter <= s->capacity);
  s->index = index;
  s->counter = counter;
  return JH_OK;
}

JH_EXPORT void jh_store_clear(jh_store* s) {
  if (!s) return;
  s->index = 0;
  s->counter = 0;
}
