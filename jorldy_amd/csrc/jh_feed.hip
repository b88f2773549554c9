// Device-resident actor -> replay feed for lockstep batched actors (SURVEY.md §8f rank 2 with ranks 1 and 3):
// frame-stack de-duplication and the n-step assembly happen in HBM, on the acting stream.
//
// Reference: every actor's env wrapper keeps a sliding stack of the last C frames (core/env/atari.py:145-149: the new
// frame is appended, the oldest dropped; reset refills the stack), Ape-X's interact_callback keeps a deque of n + 1
// steps per actor and emits {state_t, action_t, reward[n], done[n], next_state = state_{t+n}} with the actor-side
// priority |G_n - q_t| (core/agent/ape_x.py:174-199), and the learner receives both full stacks of every transition
// through a queue (2 x C x H x W bytes per transition: 56 KB at Atari shapes).
//
// Here the N actors of one BatchedValueActors tick have ALREADY uploaded their stacks for the batched forward.  Per tick:
//   1 jh_feed_compare_kernel   is this tick's stack the previous one shifted by a frame?  (bytewise: exact, needs no
//                              cooperation from the env; a reset or any other discontinuity just compares unequal)
//   2 jh_feed_append_kernel    shifted: ONE new plane goes into the actor's private plane ring and the stack's C slot
//                              numbers are the previous ones shifted; otherwise all C planes are appended
//   3 jh_feed_emit_kernel      rolling (reward, done, q, action) of the last n + 1 ticks; folds the n-step return onto
//                              q_{t+n} in the reference's float32 operation order, emits slot numbers of state_t and
//                              state_{t+n}, action_t, reward[n], done[n] and the float64 priority
// The emitted rows are what the frame-de-duplicating replay store keeps per transition (2 C slot numbers instead of 2 C
// planes, frame_dedup.py), so sampling rebuilds the stacks with the same row-gather kernel.  Nothing but the tick's
// rewards and done flags (2 N floats) crosses PCIe a second time.
#include <atomic>

#include "jh_common.h"

struct jh_feed {
  jh_ctx* ctx = nullptr;
  int N = 0, C = 0, n = 0, L = 0;
  int64_t plane = 0, R = 0, window = 0, Th = 0;
  std::atomic<int64_t> tick{0};  // advanced by the producer thread (jh_feed_emit), read by jh_feed_state from others
  float gamma = 0.f;
  void* block = nullptr;
  size_t block_bytes = 0;
  int64_t *ids = nullptr, *cur = nullptr, *alloc_hist = nullptr, *act = nullptr;
  float *rew = nullptr, *done = nullptr, *q = nullptr;
  int32_t *neq = nullptr, *flags = nullptr;
  bool pushed = false;  // this tick's stacks are in (jh_feed_push_*), jh_feed_emit closes the tick
};

namespace {
__global__ void __launch_bounds__(256) jh_feed_compare_kernel(const uint8_t* __restrict__ obs, const uint8_t* __restrict__ prev, int C,
                                                              int64_t plane, int32_t* __restrict__ neq) {
  const int a = blockIdx.x, c = blockIdx.y;  // plane c of the new stack against plane c + 1 of the previous one
  const uint8_t* x = obs + ((size_t)a * C + c) * plane;
  const uint8_t* y = prev + ((size_t)a * C + c + 1) * plane;
  int diff = 0;
  const int64_t n16 = ((((uintptr_t)x | (uintptr_t)y) & 15) == 0) ? plane >> 4 : 0;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  const uint4* y4 = reinterpret_cast<const uint4*>(y);
  for (int64_t i = threadIdx.x; i < n16; i += 256) {
    const uint4 u = x4[i], v = y4[i];
    diff |= (u.x != v.x) | (u.y != v.y) | (u.z != v.z) | (u.w != v.w);
  }
  for (int64_t i = (n16 << 4) + threadIdx.x; i < plane; i += 256) diff |= x[i] != y[i];
  if (__syncthreads_or(diff) && threadIdx.x == 0) atomicOr(neq + a, 1);
}

struct FeedAppend {
  const uint8_t* obs;
  uint8_t* pool;
  const int32_t* neq;
  const int64_t *cur_in, *ids_prev;
  int64_t *cur_out, *ids_new, *alloc_hist;
  int32_t* flags;
  int N, C;
  int64_t plane, R, window, Th, tick;
};

__global__ void __launch_bounds__(256) jh_feed_append_kernel(FeedAppend g) {
  const int a = blockIdx.x, c = blockIdx.y;
  const bool fresh = g.neq[a] != 0;
  const int64_t c0 = g.cur_in[a];
  int64_t dst = -1;
  if (fresh) dst = (int64_t)a * g.R + (c0 + c) % g.R;
  else if (c == g.C - 1) dst = (int64_t)a * g.R + c0 % g.R;
  if (dst >= 0) {
    const uint8_t* x = g.obs + ((size_t)a * g.C + c) * g.plane;
    uint8_t* y = g.pool + (size_t)dst * g.plane;
    const int64_t n16 = ((((uintptr_t)x | (uintptr_t)y) & 15) == 0) ? g.plane >> 4 : 0;
    for (int64_t i = threadIdx.x; i < n16; i += 256) reinterpret_cast<uint4*>(y)[i] = reinterpret_cast<const uint4*>(x)[i];
    for (int64_t i = (n16 << 4) + threadIdx.x; i < g.plane; i += 256) y[i] = x[i];
  }
  if (threadIdx.x != 0) return;
  g.ids_new[(size_t)a * g.C + c] = dst >= 0 ? dst : g.ids_prev[(size_t)a * g.C + c + 1];
  if (c == 0) {
    const int64_t c1 = c0 + (fresh ? g.C : 1);
    g.cur_out[a] = c1;
    g.alloc_hist[(size_t)a * g.Th + g.tick % g.Th] = c1;
    // every plane a stored transition can still reference was allocated during the last `window` ticks: they must all
    // fit the actor's ring, or a live row would decode to newer frames
    const int64_t then = g.tick >= g.window ? g.alloc_hist[(size_t)a * g.Th + (g.tick - g.window) % g.Th] : 0;
    if (c1 - then > g.R) atomicOr(g.flags, 1);
  }
}

// Frame mode: the env hands over only the NEWEST plane of every actor (+ a reset flag).  Block (a, c) rebuilds plane c of
// the actor's stack for the acting forward -- from the new frame (c == C - 1, or every c after a reset:
// core/env/atari.py:112 tiles the first frame) or from the plane pool (slot numbers of the previous stack shifted by one) --
// and block (a, C - 1) appends the new frame to the actor's plane ring.  One plane per actor and tick, always.
struct FeedFrames {
  const uint8_t* frames;   // [N][plane]
  const uint8_t* reset;    // [N] (device-visible pinned memory)
  uint8_t* pool;
  uint8_t* stack_out;      // [N][C][plane]
  const int64_t *cur_in, *ids_prev;
  int64_t *cur_out, *ids_new, *alloc_hist;
  int32_t* flags;
  int N, C, first;
  int64_t plane, R, window, Th, tick;
};

__global__ void __launch_bounds__(256) jh_feed_frames_kernel(FeedFrames g) {
  const int a = blockIdx.x, c = blockIdx.y;
  const bool fresh = g.first || g.reset[a] != 0;
  const int64_t c0 = g.cur_in[a];
  const int64_t slot = (int64_t)a * g.R + c0 % g.R;  // where this tick's frame goes
  const bool from_frame = fresh || c == g.C - 1;
  const int64_t src_slot = from_frame ? slot : g.ids_prev[(size_t)a * g.C + c + 1];
  const uint8_t* x = from_frame ? g.frames + (size_t)a * g.plane : g.pool + (size_t)src_slot * g.plane;
  uint8_t* y = g.stack_out + ((size_t)a * g.C + c) * g.plane;
  uint8_t* z = c == g.C - 1 ? g.pool + (size_t)slot * g.plane : nullptr;
  const int64_t n16 = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)(z ? z : y)) & 15) == 0) ? g.plane >> 4 : 0;
  for (int64_t i = threadIdx.x; i < n16; i += 256) {
    const uint4 v = reinterpret_cast<const uint4*>(x)[i];
    reinterpret_cast<uint4*>(y)[i] = v;
    if (z) reinterpret_cast<uint4*>(z)[i] = v;
  }
  for (int64_t i = (n16 << 4) + threadIdx.x; i < g.plane; i += 256) {
    const uint8_t v = x[i];
    y[i] = v;
    if (z) z[i] = v;
  }
  if (threadIdx.x != 0) return;
  g.ids_new[(size_t)a * g.C + c] = src_slot;
  if (c == 0) {
    const int64_t c1 = c0 + 1;
    g.cur_out[a] = c1;
    g.alloc_hist[(size_t)a * g.Th + g.tick % g.Th] = c1;
    const int64_t then = g.tick >= g.window ? g.alloc_hist[(size_t)a * g.Th + (g.tick - g.window) % g.Th] : 0;
    if (c1 - then > g.R) atomicOr(g.flags, 1);
  }
}

struct FeedEmit {
  int N, C, n, L;
  int64_t tick;
  float gamma;
  double prio_eps;
  const float *h_rew, *h_done, *d_q;  // this tick's (h_*: device-visible pinned memory)
  const int64_t* d_action;
  float *rew, *done, *q;  // rolling [L][N]
  int64_t* act;
  const int64_t* ids;  // [L][N][C]
  int64_t *state_ids, *next_ids, *action_out;
  float* reward_out;
  uint8_t* done_out;
  double* prio_out;
};

__global__ void __launch_bounds__(64) jh_feed_emit_kernel(FeedEmit g) {
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= g.N) return;
  const int L = g.L, N = g.N;
  const int i = (int)(g.tick % L);
  const float q_new = g.d_q[a];
  g.rew[(size_t)i * N + a] = g.h_rew[a];
  g.done[(size_t)i * N + a] = g.h_done[a];
  g.q[(size_t)i * N + a] = q_new;
  g.act[(size_t)i * N + a] = g.d_action[a];
  if (g.tick + 1 < L) return;  // the deque is not full yet (ape_x.py:177)
  const int o0 = (i + 1) % L;  // oldest
  float tq = q_new;            // tmp_buffer[-1]["q"]
  for (int s = 1; s <= g.n; ++s) {  // second newest .. oldest (ape_x.py:189-193), float32 like the reference's arrays
    const int k = (i + L - s) % L;
    const float keep = __fmul_rn(__fsub_rn(1.0f, g.done[(size_t)k * N + a]), g.gamma);
    tq = __fadd_rn(g.rew[(size_t)k * N + a], __fmul_rn(keep, tq));
  }
  g.prio_out[a] = (double)fabsf(__fsub_rn(tq, g.q[(size_t)o0 * N + a])) + g.prio_eps;
  g.action_out[a] = g.act[(size_t)o0 * N + a];
  for (int s = 0; s < g.n; ++s) {  // oldest .. second newest
    const int k = (o0 + s) % L;
    g.reward_out[(size_t)a * g.n + s] = g.rew[(size_t)k * N + a];
    g.done_out[(size_t)a * g.n + s] = g.done[(size_t)k * N + a] != 0.f ? 1 : 0;
  }
  for (int c = 0; c < g.C; ++c) {
    g.state_ids[(size_t)a * g.C + c] = g.ids[((size_t)o0 * N + a) * g.C + c];
    g.next_ids[(size_t)a * g.C + c] = g.ids[((size_t)i * N + a) * g.C + c];
  }
}
}  // namespace

JH_EXPORT int jh_feed_create(jh_ctx* ctx, int32_t n_actors, int32_t C, int64_t plane_bytes, int32_t n_step, float gamma,
                             int64_t planes_per_actor, int64_t window_ticks, jh_feed** out) {
  JH_ARG(ctx && out && n_actors > 0 && C > 0 && plane_bytes > 0 && n_step > 0 && window_ticks > 0);
  JH_ARG(planes_per_actor >= 2 * (int64_t)C);
  jh_feed* f = new jh_feed();
  f->ctx = ctx; f->N = n_actors; f->C = C; f->n = n_step; f->L = n_step + 1; f->plane = plane_bytes; f->R = planes_per_actor;
  f->window = window_ticks; f->Th = window_ticks + 1; f->gamma = gamma;
  const size_t N = n_actors, L = f->L;
  const size_t b_ids = sizeof(int64_t) * L * N * C, b_cur = sizeof(int64_t) * 2 * N, b_hist = sizeof(int64_t) * N * (size_t)f->Th,
               b_act = sizeof(int64_t) * L * N, b_f = sizeof(float) * L * N, b_neq = sizeof(int32_t) * N;
  const size_t total = b_ids + b_cur + b_hist + b_act + 3 * b_f + b_neq + 64;
  f->block_bytes = total;
  hipError_t e = hipMalloc(&f->block, total);
  if (e != hipSuccess) {
    delete f;
    return jh_fail(JH_ERR_HIP, "jh_feed_create: hipMalloc(%zu) -> %s", total, hipGetErrorString(e));
  }
  e = hipMemset(f->block, 0, total);
  if (e != hipSuccess) {
    (void)hipFree(f->block);
    delete f;
    return jh_fail(JH_ERR_HIP, "jh_feed_create: hipMemset -> %s", hipGetErrorString(e));
  }
  char* p = static_cast<char*>(f->block);
  f->ids = reinterpret_cast<int64_t*>(p); p += b_ids;
  f->cur = reinterpret_cast<int64_t*>(p); p += b_cur;
  f->alloc_hist = reinterpret_cast<int64_t*>(p); p += b_hist;
  f->act = reinterpret_cast<int64_t*>(p); p += b_act;
  f->rew = reinterpret_cast<float*>(p); p += b_f;
  f->done = reinterpret_cast<float*>(p); p += b_f;
  f->q = reinterpret_cast<float*>(p); p += b_f;
  f->neq = reinterpret_cast<int32_t*>(p); p += b_neq;
  f->flags = reinterpret_cast<int32_t*>(p);
  *out = f;
  return JH_OK;
}

JH_EXPORT void jh_feed_destroy(jh_feed* f) {
  if (!f) return;
  if (f->block) (void)hipFree(f->block);
  delete f;
}

// First half of a tick, stack mode: the actors' stacks as they were uploaded for the forward.
JH_EXPORT int jh_feed_push_stacks(jh_feed* f, const uint8_t* d_obs, const uint8_t* d_prev_obs, uint8_t* d_pool, jh_stream stream) {
  JH_ARG(f && d_obs && d_pool && !f->pushed);
  JH_ARG(f->tick == 0 || d_prev_obs != nullptr);
  hipStream_t st = jh_s(stream);
  const int N = f->N, C = f->C, L = f->L;
  const int64_t t = f->tick;
  // first tick: every plane is new; C == 1: "shifted by a frame" holds trivially (nothing shared, one new plane)
  JH_HIP(hipMemsetAsync(f->neq, t == 0 ? 1 : 0, sizeof(int32_t) * (size_t)N, st));
  if (t > 0 && C > 1) {
    JH_LAUNCH(jh_feed_compare_kernel, dim3(N, C - 1), dim3(256), 0, st, d_obs, d_prev_obs, C, f->plane, f->neq);
    JH_LAUNCH_CHECK();
  }
  const int slot = (int)(t % L), prev = (int)((t + L - 1) % L);
  FeedAppend ga{};
  ga.obs = d_obs; ga.pool = d_pool; ga.neq = f->neq;
  ga.cur_in = f->cur + (size_t)(t & 1) * N; ga.cur_out = f->cur + (size_t)((t + 1) & 1) * N;
  ga.ids_prev = f->ids + (size_t)prev * N * C; ga.ids_new = f->ids + (size_t)slot * N * C;
  ga.alloc_hist = f->alloc_hist; ga.flags = f->flags;
  ga.N = N; ga.C = C; ga.plane = f->plane; ga.R = f->R; ga.window = f->window; ga.Th = f->Th; ga.tick = t;
  JH_LAUNCH(jh_feed_append_kernel, dim3(N, C), dim3(256), 0, st, ga);
  JH_LAUNCH_CHECK();
  f->pushed = true;
  return JH_OK;
}

// First half of a tick, frame mode: only the newest plane of every actor (device) + the env's reset flags (host);
// d_stack_out receives the rebuilt stacks [N][C][plane] for the acting forward.
JH_EXPORT int jh_feed_push_frames(jh_feed* f, const uint8_t* d_frames, const uint8_t* h_reset, uint8_t* d_pool, uint8_t* d_stack_out,
                                  jh_stream stream) {
  JH_ARG(f && d_frames && h_reset && d_pool && d_stack_out && !f->pushed);
  hipStream_t st = jh_s(stream);
  const int N = f->N, C = f->C, L = f->L;
  const int64_t t = f->tick;
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(f->ctx, (size_t)N, &slab);
  if (rc) return rc;
  memcpy(slab->host, h_reset, (size_t)N);
  const int slot = (int)(t % L), prev = (int)((t + L - 1) % L);
  FeedFrames g{};
  g.frames = d_frames; g.reset = static_cast<const uint8_t*>(slab->dev); g.pool = d_pool; g.stack_out = d_stack_out;
  g.cur_in = f->cur + (size_t)(t & 1) * N; g.cur_out = f->cur + (size_t)((t + 1) & 1) * N;
  g.ids_prev = f->ids + (size_t)prev * N * C; g.ids_new = f->ids + (size_t)slot * N * C;
  g.alloc_hist = f->alloc_hist; g.flags = f->flags;
  g.N = N; g.C = C; g.first = t == 0; g.plane = f->plane; g.R = f->R; g.window = f->window; g.Th = f->Th; g.tick = t;
  JH_LAUNCH(jh_feed_frames_kernel, dim3(N, C), dim3(256), 0, st, g);
  JH_LAUNCH_CHECK();
  rc = jh_ctx_slab_release(f->ctx, slab, st);
  if (rc) return rc;
  f->pushed = true;
  return JH_OK;
}

// Second half of a tick: the action taken / its Q (device) and the env's answer (host) -> n-step rows (see the header).
JH_EXPORT int jh_feed_emit(jh_feed* f, const int64_t* d_action, const float* d_q, const float* h_reward, const float* h_done,
                           double prio_eps, int64_t* d_state_ids, int64_t* d_next_ids, int64_t* d_action_out, float* d_reward_out,
                           uint8_t* d_done_out, double* d_prio_out, int32_t* emitted, jh_stream stream) {
  JH_ARG(f && d_action && d_q && h_reward && h_done && emitted && f->pushed);
  JH_ARG(d_state_ids && d_next_ids && d_action_out && d_reward_out && d_done_out && d_prio_out);
  hipStream_t st = jh_s(stream);
  const int N = f->N, C = f->C, L = f->L;
  const int64_t t = f->tick;
  jh_pinned_slab* slab = nullptr;
  int rc = jh_ctx_slab(f->ctx, sizeof(float) * 2 * (size_t)N, &slab);
  if (rc) return rc;
  memcpy(slab->host, h_reward, sizeof(float) * (size_t)N);
  memcpy(static_cast<float*>(slab->host) + N, h_done, sizeof(float) * (size_t)N);
  FeedEmit ge{};
  ge.N = N; ge.C = C; ge.n = f->n; ge.L = L; ge.tick = t; ge.gamma = f->gamma; ge.prio_eps = prio_eps;
  ge.h_rew = static_cast<const float*>(slab->dev); ge.h_done = static_cast<const float*>(slab->dev) + N;
  ge.d_q = d_q; ge.d_action = d_action;
  ge.rew = f->rew; ge.done = f->done; ge.q = f->q; ge.act = f->act; ge.ids = f->ids;
  ge.state_ids = d_state_ids; ge.next_ids = d_next_ids; ge.action_out = d_action_out; ge.reward_out = d_reward_out;
  ge.done_out = d_done_out; ge.prio_out = d_prio_out;
  JH_LAUNCH(jh_feed_emit_kernel, dim3((N + 63) / 64), dim3(64), 0, st, ge);
  JH_LAUNCH_CHECK();
  rc = jh_ctx_slab_release(f->ctx, slab, st);
  if (rc) return rc;
  *emitted = t + 1 >= L ? N : 0;
  f->tick = t + 1;
  f->pushed = false;
  return JH_OK;
}

JH_EXPORT int jh_feed_tick(jh_feed* f, const uint8_t* d_obs, const uint8_t* d_prev_obs, uint8_t* d_pool, const int64_t* d_action,
                           const float* d_q, const float* h_reward, const float* h_done, double prio_eps, int64_t* d_state_ids,
                           int64_t* d_next_ids, int64_t* d_action_out, float* d_reward_out, uint8_t* d_done_out,
                           double* d_prio_out, int32_t* emitted, jh_stream stream) {
  int rc = jh_feed_push_stacks(f, d_obs, d_prev_obs, d_pool, stream);
  if (rc) return rc;
  return jh_feed_emit(f, d_action, d_q, h_reward, h_done, prio_eps, d_state_ids, d_next_ids, d_action_out, d_reward_out, d_done_out,
                      d_prio_out, emitted, stream);
}

JH_EXPORT int jh_feed_state(jh_feed* f, int32_t* h_flags, int64_t* h_planes_written, jh_stream stream) {
  JH_ARG(f && h_flags && h_planes_written);
  hipStream_t st = jh_s(stream);
  std::vector<int64_t> cur(f->N);
  const int64_t tick = f->tick.load(std::memory_order_acquire);  // ONE snapshot: another thread may be closing a tick right now
  JH_HIP(hipMemcpyAsync(h_flags, f->flags, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  JH_HIP(hipMemcpyAsync(cur.data(), f->cur + (size_t)(tick & 1) * f->N, sizeof(int64_t) * (size_t)f->N, hipMemcpyDeviceToHost, st));
  JH_HIP(hipStreamSynchronize(st));
  int64_t s = 0;
  for (int64_t v : cur) s += v;
  *h_planes_written = s;
  return JH_OK;
}

// Checkpoint of the feed's own state (a resumable device-fed replay): the device block (slot numbers of the
// rolling stacks, per-actor plane cursors and their per-tick history, the rolling action / reward / done / q windows, flags) + the
// tick counter.  With the plane pool, the store's rows and the sum tree (saved by their owners) a feed created with the SAME
// geometry continues exactly where this one stood.  Both calls synchronise `stream`; jh_feed_save must not race a tick.
JH_EXPORT int64_t jh_feed_state_bytes(const jh_feed* f) { return f ? (int64_t)f->block_bytes + 16 : 0; }

JH_EXPORT int jh_feed_save(jh_feed* f, void* h_out, int64_t bytes, jh_stream stream) {
  JH_ARG(f && h_out && bytes == jh_feed_state_bytes(f));
  if (f->pushed) return jh_fail(JH_ERR_STATE, "jh_feed_save between the two halves of a tick");
  int64_t head[2] = {f->tick.load(), (int64_t)f->block_bytes};
  memcpy(h_out, head, 16);
  JH_HIP(hipMemcpyAsync((char*)h_out + 16, f->block, f->block_bytes, hipMemcpyDeviceToHost, jh_s(stream)));
  JH_HIP(hipStreamSynchronize(jh_s(stream)));
  return JH_OK;
}

JH_EXPORT int jh_feed_load(jh_feed* f, const void* h_in, int64_t bytes, jh_stream stream) {
  JH_ARG(f && h_in && bytes == jh_feed_state_bytes(f));
  int64_t head[2];
  memcpy(head, h_in, 16);
  if (head[1] != (int64_t)f->block_bytes) return jh_fail(JH_ERR_ARG, "jh_feed_load: the saved feed has another geometry (%lld vs %zu state bytes)", (long long)head[1], f->block_bytes);
  JH_HIP(hipMemcpyAsync(f->block, (const char*)h_in + 16, f->block_bytes, hipMemcpyHostToDevice, jh_s(stream)));
  JH_HIP(hipStreamSynchronize(jh_s(stream)));
  f->tick.store(head[0]);
  f->pushed = false;
  return JH_OK;
}
