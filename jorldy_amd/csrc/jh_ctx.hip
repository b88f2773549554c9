// Context, error slot, pinned staging slabs.
#include <stdarg.h>

#include <chrono>

#include "jh_common.h"

std::string& jh_err_slot() {
  static thread_local std::string s;
  return s;
}

int jh_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  jh_err_slot() = buf;
  return code;
}

JH_EXPORT int jh_abi_version(void) { return JH_ABI_VERSION; }
JH_EXPORT const char* jh_last_error(void) { return jh_err_slot().c_str(); }

JH_EXPORT int jh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

JH_EXPORT int jh_ctx_create(int device, jh_ctx** out) {
  JH_ARG(out != nullptr);
  int n = jh_device_count();
  if (n <= 0) return jh_fail(JH_ERR_NODEVICE, "no HIP device visible: libjorldy_hip needs an MI355X (gfx950)");
  JH_ARG(device >= 0 && device < n);
  JH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  JH_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return jh_fail(JH_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  jh_ctx* c = new jh_ctx();
  c->device = device;
  *out = c;
  return JH_OK;
}

JH_EXPORT void jh_ctx_destroy(jh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& s : ctx->slabs) {
    if (s.ev) {
      (void)hipEventSynchronize(s.ev);
      (void)hipEventDestroy(s.ev);
    }
    if (s.host) (void)hipHostFree(s.host);
  }
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  for (void* b : ctx->retired) (void)hipFree(b);
  delete ctx;
}

JH_EXPORT int jh_ctx_sync(jh_ctx* ctx, jh_stream stream) {
  JH_ARG(ctx != nullptr);
  JH_HIP(hipStreamSynchronize(jh_s(stream)));
  return JH_OK;
}

// Slabs and scratch are shared by every thread that uses this context (learner thread, batched-actor thread, ring
// producers): cursor, in_use / pending flags and the scratch list are guarded by the context's mutex.  A slab handed out
// stays with ONE caller (in_use) until jh_ctx_slab_release: a second thread that wraps the ring skips it instead of
// overwriting -- or, on growth, freeing -- memory the first is still filling (jh_store_stage_begin holds its slab across
// calls).  When every slab is held a new one is appended.
int jh_ctx_slab(jh_ctx* ctx, size_t bytes, jh_pinned_slab** out) {
  std::unique_lock<std::mutex> lock(ctx->mu);
  if (ctx->slabs.empty()) ctx->slabs.resize(jh_ctx::kSlabs);
  const int n = (int)ctx->slabs.size();
  jh_pinned_slab* pick = nullptr;
  for (int k = 0; k < n; ++k) {
    jh_pinned_slab& c = ctx->slabs[(ctx->next_slab + k) % n];
    if (!c.in_use) {
      pick = &c;
      ctx->next_slab = (ctx->next_slab + k + 1) % n;
      break;
    }
  }
  if (!pick) {
    if (n >= jh_ctx::kMaxSlabs) return jh_fail(JH_ERR_STATE, "all %d pinned staging slabs are held (a jh_store_stage_begin without its commit?)", n);
    ctx->slabs.emplace_back();
    pick = &ctx->slabs.back();
  }
  jh_pinned_slab& s = *pick;
  s.in_use = true;
  const bool wait = s.pending;
  s.pending = false;
  lock.unlock();  // the slab is ours now: wait / (re)allocate without holding up the other threads
  auto fail = [&](int rc) {
    std::lock_guard<std::mutex> g(ctx->mu);
    s.in_use = false;
    return rc;
  };
  if (wait) {
    hipError_t se = hipEventSynchronize(s.ev);
    if (se != hipSuccess) {
      return fail(jh_fail(JH_ERR_HIP, "hipEventSynchronize on a staging slab failed: %s", hipGetErrorString(se)));
    }
  }
  if (s.bytes < bytes) {
    if (s.host && hipHostFree(s.host) != hipSuccess) return fail(jh_fail(JH_ERR_HIP, "hipHostFree of a staging slab failed"));
    size_t want = 1 << 16;
    while (want < bytes) want <<= 1;
    s.host = nullptr;
    s.bytes = 0;
    // coherent, device-mapped pinned memory: kernels may read small inputs in place,
    // hipMemcpyAsync from it is a true async DMA.
    if (hipHostMalloc(&s.host, want, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&s.dev, s.host, 0) != hipSuccess) {
      (void)hipGetLastError();
      s.host = nullptr;
      return fail(jh_fail(JH_ERR_HIP, "hipHostMalloc of a %zu-byte staging slab failed", want));
    }
    s.bytes = want;
  }
  if (!s.ev && hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) return fail(jh_fail(JH_ERR_HIP, "hipEventCreate failed"));
  *out = &s;
  return JH_OK;
}

int jh_ctx_slab_release(jh_ctx* ctx, jh_pinned_slab* slab, hipStream_t stream) {
  hipError_t e = hipEventRecord(slab->ev, stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  slab->pending = (e == hipSuccess);
  slab->in_use = false;
  if (e != hipSuccess) return jh_fail(JH_ERR_HIP, "hipEventRecord on a staging slab -> %s", hipGetErrorString(e));
  return JH_OK;
}

// A stream capture that failed (hipErrorStreamCaptureInvalidated) leaves its stream in the capture until somebody ends it; a caller whose
// framework gave up half way (torch.cuda.graph.__exit__ raises out of capture_end) calls this before it reuses or abandons the stream.
// Returns 1 when there was a capture to end, 0 when the stream was not capturing; the sticky error of the failed capture is cleared.
JH_EXPORT int jh_stream_abort_capture(jh_stream stream) {
  hipStream_t st = jh_s(stream);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  int ended = 0;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(st, &g);
    if (g) (void)hipGraphDestroy(g);
    ended = 1;
  }
  (void)hipGetLastError();
  return ended;
}

int jh_ctx_scratch(jh_ctx* ctx, size_t bytes, void** out) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (ctx->scratch_bytes < bytes) {
    // Growing is rare (first calls).  A block that was handed out is NEVER freed before the context dies: its address
    // may be baked into a captured hipGraph (td / c51 / ppo loss partials), and nothing may hipFree / synchronise
    // during a capture.  Old blocks are kept on a list (a few hundred KB in total).
    size_t want = 1 << 16;
    while (want < bytes) want <<= 1;
    void* blk = nullptr;
    JH_HIP(hipMalloc(&blk, want));
    if (ctx->scratch) ctx->retired.push_back(ctx->scratch);
    ctx->scratch = blk;
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return JH_OK;
}

// Host-side arrival wait on device-mapped pinned memory: returns 0 as soon as none of the 32-bit words base[idx[i]]
// equals `sentinel` any more (a finishing kernel overwrote them), 1 after timeout_s.  Pure host spin (pause between polls);
// callers come through ctypes, i.e. WITHOUT the Python GIL -- a Python spin loop would starve the other threads of the
// process.  jh_host_wait_marks: the same for float words and a float sentinel (compared by bit pattern).
JH_EXPORT int jh_host_wait_words(const uint32_t* base, const int32_t* idx, int32_t n, uint32_t sentinel, double timeout_s) {
  if (!base || !idx || n <= 0) return jh_fail(JH_ERR_ARG, "jh_host_wait_words: bad argument");
  const volatile uint32_t* v = base;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 1;; ++spin) {
    bool all = true;
    for (int i = 0; i < n; ++i) all = all && v[idx[i]] != sentinel;
    if (all) return 0;
    __builtin_ia32_pause();
    if ((spin & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return 1;
  }
}
JH_EXPORT int jh_host_wait_marks(const float* base, const int32_t* idx, int32_t n, float sentinel, double timeout_s) {
  uint32_t bits;
  memcpy(&bits, &sentinel, 4);
  return jh_host_wait_words(reinterpret_cast<const uint32_t*>(base), idx, n, bits, timeout_s);
}

// The epoch shuffles of PPO.learn (core/agent/ppo.py:116-118: `idxs = np.arange(M)`, then `np.random.shuffle(idxs)` once per epoch,
// cumulatively in place) drawn from numpy's OWN global MT19937 state, in place, with numpy's own algorithm -- so that every draw the
// reference would make is made, in the same order, from the same generator (the minibatch index lists stay bit-identical and whatever
// consumes np.random afterwards sees the same stream).  numpy: RandomState.shuffle -> _shuffle_raw: for i = n-1 .. 1:
// j = random_interval(i); swap(x[i], x[j]); random_interval(max) = masked rejection sampling on next_uint32 (next_uint64 above 2^32-1).
// `next_uint32` / `next_uint64` / `bitgen_state` are the function pointers and the state address numpy publishes through
// BitGenerator.ctypes.  12 us per np.random.shuffle call of 1024 indices in Python -> ~3 us here; three epochs sat between the
// two graph launches of learn() with the GPU idle.
JH_EXPORT int jh_np_legacy_shuffles(void* bitgen_state, void* next_uint32_fn, void* next_uint64_fn, int64_t n, int32_t epochs,
                                    int64_t* h_perm_out) {
  if (!bitgen_state || !next_uint32_fn || !next_uint64_fn || !h_perm_out || n <= 0 || epochs <= 0)
    return jh_fail(JH_ERR_ARG, "jh_np_legacy_shuffles: bad argument");
  typedef uint32_t (*u32_fn)(void*);
  typedef uint64_t (*u64_fn)(void*);
  const u32_fn next32 = reinterpret_cast<u32_fn>(next_uint32_fn);
  const u64_fn next64 = reinterpret_cast<u64_fn>(next_uint64_fn);
  int64_t* x = h_perm_out;  // epoch e is shuffled in place in its own row, seeded with the previous epoch's result
  for (int64_t i = 0; i < n; ++i) x[i] = i;
  for (int e = 0; e < epochs; ++e) {
    if (e > 0) {
      memcpy(h_perm_out + (size_t)e * n, h_perm_out + (size_t)(e - 1) * n, sizeof(int64_t) * (size_t)n);
      x = h_perm_out + (size_t)e * n;
    }
    for (int64_t i = n - 1; i >= 1; --i) {
      const uint64_t max = (uint64_t)i;
      uint64_t mask = max, value;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
      if (max <= 0xffffffffULL) {
        while ((value = ((uint64_t)next32(bitgen_state) & mask)) > max) {}
      } else {
        while ((value = (next64(bitgen_state) & mask)) > max) {}
      }
      const int64_t t = x[i];
      x[i] = x[value];
      x[value] = t;
    }
  }
  return JH_OK;
}

// Pinned, device-mapped host memory for callers that exchange small per-step data with kernels
// (observations in, actions out) without a memcpy node.
JH_EXPORT int jh_pinned_alloc(jh_ctx* ctx, int64_t bytes, void** host_out, void** dev_out) {
  JH_ARG(ctx && host_out && dev_out && bytes > 0);
  JH_HIP(hipSetDevice(ctx->device));
  void* h = nullptr;
  JH_HIP(hipHostMalloc(&h, (size_t)bytes, hipHostMallocMapped));
  void* d = nullptr;
  JH_HIP(hipHostGetDevicePointer(&d, h, 0));
  memset(h, 0, (size_t)bytes);
  *host_out = h;
  *dev_out = d;
  return JH_OK;
}

JH_EXPORT void jh_pinned_free(void* host) {
  if (host) (void)hipHostFree(host);
}

// ------------------------------------------------------------------------------ PMC calibration
// A streaming read of exactly `bytes` bytes with a chosen access width per lane (4 / 8 / 16 bytes), summed into d_out so that
// nothing is optimised away: the known byte count that rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* are calibrated against
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern"; tools/pmc_calibrate.sh).
template <typename V>
__global__ void __launch_bounds__(256) jh_calib_stream_kernel(const V* __restrict__ src, int64_t n, float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const V v = src[i];
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / 4); ++k) acc += f[k];
  }
  if (acc == 123.456f) out[0] = acc;  // never true for the calibration data; keeps the loads alive
}

JH_EXPORT int jh_calib_stream(jh_ctx* ctx, const void* d_src, int64_t bytes, int32_t width, float* d_out, jh_stream stream) {
  JH_ARG(ctx && d_src && d_out && bytes > 0 && (width == 4 || width == 8 || width == 16) && bytes % width == 0);
  const int64_t n = bytes / width;
  const unsigned grid = 2048;
  if (width == 4) JH_LAUNCH_NAMED("jh_calib_stream_kernel<4>", jh_calib_stream_kernel<float>, dim3(grid), dim3(256), 0, jh_s(stream), (const float*)d_src, n, d_out);
  else if (width == 8) JH_LAUNCH_NAMED("jh_calib_stream_kernel<8>", jh_calib_stream_kernel<float2>, dim3(grid), dim3(256), 0, jh_s(stream), (const float2*)d_src, n, d_out);
  else JH_LAUNCH_NAMED("jh_calib_stream_kernel<16>", jh_calib_stream_kernel<float4>, dim3(grid), dim3(256), 0, jh_s(stream), (const float4*)d_src, n, d_out);
  JH_LAUNCH_CHECK();
  return JH_OK;
}

// ------------------------------------------------------------------------------ kernel profiler
#include <map>
bool g_jh_prof_on = false;
void jh_note_earlier_error(hipError_t e, const char* before) {
  if (e == hipSuccess) return;
  static int shown = 0;
  char buf[256];
  snprintf(buf, sizeof buf, "an earlier HIP error surfaced before launching %s: %s (cleared; it belongs to work enqueued before this call)", before, hipGetErrorString(e));
  jh_err_slot() = buf;
  if (shown < 8) {
    ++shown;
    fprintf(stderr, "[libjorldy_hip] %s\n", buf);
  }
}
int g_jh_prof_repeat = 1;
namespace {
struct ProfRec {
  const char* name;
  hipEvent_t e0, e1;
  int reps;
  double work;
};
std::vector<ProfRec> g_prof;
}  // namespace

void jh_prof_begin(const char* name, hipStream_t st, int reps, double work) {
  ProfRec r{name, nullptr, nullptr, reps, work};
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, st);
  g_prof.push_back(r);
}

void jh_prof_end(hipStream_t st) {
  if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().e1, st);
}

// Record n back-to-back event pairs with nothing in between under the name "__event_pair_overhead":
// the fixed cost that every measured kernel duration includes (subtracted by bench.py).
JH_EXPORT int jh_prof_calibrate(int32_t n, jh_stream stream) {
  if (!g_jh_prof_on) return jh_fail(JH_ERR_STATE, "jh_prof_calibrate: profiling is off");
  for (int i = 0; i < n; ++i) {
    jh_prof_begin("__event_pair_overhead", jh_s(stream), 1, 0.0);
    jh_prof_end(jh_s(stream));
  }
  return JH_OK;
}

JH_EXPORT int jh_prof_enable(int32_t on) {
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  g_jh_prof_on = on != 0;
  g_jh_prof_repeat = on > 1 ? on : 1;
  return JH_OK;
}

// Writes one line per kernel: "<name>\t<launches>\t<total_ms>\t<work>\n" (synchronises the device); work = flops of
// the MFMA kernels that declare it (per launch x launches), 0 otherwise.
JH_EXPORT int jh_prof_report(char* buf, int64_t cap) {
  JH_ARG(buf && cap > 0);
  JH_HIP(hipDeviceSynchronize());
  struct Agg { long n = 0; double ms = 0, work = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    auto& a = agg[r.name];
    a.n += r.reps;
    a.ms += ms;
    a.work += r.work * r.reps;
  }
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.work);
    out += line;
  }
  if ((int64_t)out.size() + 1 > cap) return jh_fail(JH_ERR_ARG, "jh_prof_report: buffer too small (%zu needed)", out.size() + 1);
  memcpy(buf, out.c_str(), out.size() + 1);
  return JH_OK;
}

// Host cores that sit on the same NUMA node / PCIe root as this context's GPU (sysfs local_cpulist of the device's
// PCI function), e.g. "64-127,192-255".  Acting crosses PCIe twice per timestep; from the far socket every crossing
// costs ~1.8 us more (measured: 12.9 vs 9.3 us per timestep), so collectors pin their host thread to these cores.
JH_EXPORT int jh_ctx_local_cpulist(jh_ctx* ctx, char* out, int64_t len) {
  JH_ARG(ctx && out && len > 16);
  char bus[64] = {0};
  JH_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, ctx->device));
  for (char* c = bus; *c; ++c)
    if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');  // sysfs names are lower case
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE* f = fopen(path, "r");
  if (!f) return jh_fail(JH_ERR_STATE, "cannot read %s", path);
  const size_t n = fread(out, 1, (size_t)len - 1, f);
  fclose(f);
  out[n] = 0;
  for (size_t i = 0; i < n; ++i)
    if (out[i] == '\n') out[i] = 0;
  return JH_OK;
}

// ------------------------------------------------------------------------------ device Gaussian draws
// NoisyNet noise (core/network/utils.py:58-60 draws torch.randn per forward) generated on the device by a counter-based
// generator: element i of call c = Box-Muller on splitmix64(seed, c, i).  The call counter lives in DEVICE memory and is
// advanced by the last workgroup to finish (arrival ticket; every workgroup has read the counter before it arrives), so
// a captured hipGraph draws fresh noise on every replay without any host involvement.
// d_state: uint64[4] = {seed, call counter, arrival ticket, 0}.
__global__ void __launch_bounds__(256) jh_normal_fill_kernel(int64_t n, float* __restrict__ out, unsigned long long* __restrict__ state) {
  const unsigned long long seed = state[0], ctr = state[1];
  const int64_t pairs = (n + 1) >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * 256) {
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + ctr * 0xD1B54A32D192ED03ull + (unsigned long long)i * 0x8CB92BA72F3D8DD7ull;
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    const float u1 = (float)((unsigned)(x >> 40) + 1u) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)((unsigned)(x & 0xFFFFFFull)) * (1.0f / 16777216.0f);  // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.2831853071795865f * u2, &sn, &cs);
    out[2 * i] = r * cs;
    if (2 * i + 1 < n) out[2 * i + 1] = r * sn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(state + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(state + 2, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(state + 1, ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

JH_EXPORT int jh_normal_fill(jh_ctx* ctx, int64_t n, float* d_out, uint64_t* d_state, jh_stream stream) {
  JH_ARG(ctx && d_out && d_state && n > 0);
  const int64_t pairs = (n + 1) / 2;
  int64_t nb = (pairs + 255) / 256;
  if (nb > 512) nb = 512;
  JH_LAUNCH(jh_normal_fill_kernel, dim3((unsigned)nb), dim3(256), 0, jh_s(stream), n, d_out, (unsigned long long*)d_state);
  JH_LAUNCH_CHECK();
  return JH_OK;
}
