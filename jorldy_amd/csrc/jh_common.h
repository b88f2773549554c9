// Shared internals of libjorldy_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/jorldy_hip.h"

#define JH_EXPORT extern "C" __attribute__((visibility("default")))

std::string& jh_err_slot();
int jh_fail(int code, const char* fmt, ...);

#define JH_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return jh_fail(JH_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)

#define JH_ARG(cond)                                                                    \
  do {                                                                                  \
    if (!(cond)) return jh_fail(JH_ERR_ARG, "%s:%d bad argument: %s", __FILE__, __LINE__, #cond); \
  } while (0)

#define JH_LAUNCH_CHECK()                                                                      \
  do {                                                                                         \
    hipError_t _e = hipGetLastError();                                                         \
    if (_e != hipSuccess)                                                                      \
      return jh_fail(JH_ERR_HIP, "%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
// Disabled by default: JH_LAUNCH is then exactly hipLaunchKernelGGL.  jh_prof_enable(R > 1): kernels launched
// through JH_LAUNCH_IDEM (idempotent: same inputs -> same outputs, no counters advanced) run R times back to back
// inside ONE event pair, so the pair's fixed cost (~4 us, more than most kernels here) is amortised and the
// average is the kernel's duration in a dependent chain -- comparable with rocprofv3's kernel-trace average.
extern bool g_jh_prof_on;
extern int g_jh_prof_repeat;
// A sticky / asynchronous HIP error left behind by earlier work: clearing it keeps THIS launch's check meaningful, but it must not vanish --
// it goes to jh_last_error() and (the first few times) to stderr, attributed to "before <launch>".
void jh_note_earlier_error(hipError_t e, const char* before);
void jh_prof_begin(const char* name, hipStream_t st, int reps, double work);
void jh_prof_end(hipStream_t st);
#define JH_LAUNCH_WORK(NAME, WORK, REPS, KERNEL, GRID, BLOCK, LDS, ST, ...)  \
  do {                                                                       \
    const int _reps = g_jh_prof_on ? (REPS) : 1;                             \
    if (g_jh_prof_on) jh_prof_begin(NAME, ST, _reps, WORK);                  \
    jh_note_earlier_error(hipGetLastError(), NAME); /* an error of EARLIER work (a capture torch abandoned, another library's launch) is not this launch's: cleared, but no longer silently */ \
    for (int _i = 0; _i < _reps; ++_i) hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__); \
    if (g_jh_prof_on) jh_prof_end(ST);                                       \
  } while (0)
#define JH_LAUNCH_NAMED(NAME, KERNEL, GRID, BLOCK, LDS, ST, ...) JH_LAUNCH_WORK(NAME, 0.0, 1, KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__)
// idempotent kernel with a declared amount of work (flops) per launch
#define JH_LAUNCH_IDEM(NAME, WORK, KERNEL, GRID, BLOCK, LDS, ST, ...) JH_LAUNCH_WORK(NAME, WORK, g_jh_prof_repeat, KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__)
#define JH_LAUNCH(KERNEL, GRID, BLOCK, LDS, ST, ...) JH_LAUNCH_NAMED(#KERNEL, KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__)

static inline hipStream_t jh_s(jh_stream s) { return reinterpret_cast<hipStream_t>(s); }

static inline size_t jh_dtype_size(int dt) {
  switch (dt) {
    case JH_U8: return 1;
    case JH_F32: return 4;
    case JH_I64: return 8;
    case JH_F64: return 8;
    case JH_I32: return 4;
    default: return 0;
  }
}

// A pinned (host-coherent, device-mapped) slab that is reused round-robin; the event guards
// reuse while an async copy / kernel that reads it may still be in flight.
struct jh_pinned_slab {
  void* host = nullptr;
  void* dev = nullptr;  // device-visible alias of `host`
  size_t bytes = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;  // release event recorded, not yet known to have fired
  bool in_use = false;   // handed out by jh_ctx_slab, not yet released (guarded by jh_ctx::mu)
};

struct jh_ctx {
  int device = 0;
  static constexpr int kSlabs = 8;      // initial ring; grows (never shrinks) when every slab is held by some caller
  static constexpr int kMaxSlabs = 64;
  std::deque<jh_pinned_slab> slabs;     // deque: growing keeps the addresses callers hold stable
  int next_slab = 0;
  // small device scratch for reductions (partials) -- grows on demand; outgrown blocks stay alive (captured graphs)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  std::vector<void*> retired;
  std::mutex mu;  // slabs + scratch are shared by the threads of a process (learner, batched actors, ring producers)
};

// Get a pinned slab of at least `bytes` (waits for its previous use to drain).  The slab belongs to the caller -- any thread
// of the process -- until jh_ctx_slab_release; slabs other callers still hold are skipped, a fresh one is added when all are held.
int jh_ctx_slab(jh_ctx* ctx, size_t bytes, jh_pinned_slab** out);
// Mark the slab busy until everything enqueued on `stream` so far has executed.
int jh_ctx_slab_release(jh_ctx* ctx, jh_pinned_slab* slab, hipStream_t stream);
int jh_ctx_scratch(jh_ctx* ctx, size_t bytes, void** out);

// ---------------------------------------------------------------- object layouts (shared between TUs)
struct jh_store {
  jh_ctx* ctx = nullptr;
  int64_t capacity = 0;
  int n_cols = 0;
  std::vector<jh_col_desc> cols;
  std::vector<void*> dev;         // device column bases
  std::vector<size_t> row_bytes;  // bytes per transition per column
  int64_t index = 0;              // buffer_index
  int64_t counter = 0;            // buffer_counter
  // staged push state
  jh_pinned_slab* staged = nullptr;
  int64_t staged_n = 0;
  std::vector<size_t> staged_off;
};

int jh_store_append(jh_store* s, int64_t n, const void* const* cols, hipMemcpyKind kind, hipStream_t st);
int jh_store_stage_commit_extra(jh_store* s, int n_extra, const void* const* x_src, void* const* x_dst, const int64_t* x_bytes, hipStream_t st);
bool jh_store_commit_is_one_launch(const jh_store* s, int64_t n);
int jh_store_stage_abort(jh_store* s, hipStream_t st);
int jh_store_stage_commit_gated(jh_store* s, int n_extra, const void* const* x_src, void* const* x_dst, const int64_t* x_bytes, const unsigned* gate,
                                unsigned gate_val, hipStream_t st);

struct jh_cartpole {
  int W = 0;
  std::vector<double> s;  // [W][4]: x, x_dot, theta, theta_dot
  std::vector<int64_t> t;
  std::vector<uint64_t> rng;
};

// synthetic continuous-control env (jh_env.hip): stands in for MuJoCo Hopper at config.ppo.mujoco shapes
struct jh_control {
  int W = 0, S = 0, A = 0;
  std::vector<double> s;  // [W][S]
  std::vector<int64_t> t;
  std::vector<uint64_t> rng;
};

void jh_cartpole_obs_rows(const jh_cartpole* e, int r0, int r1, float* h_obs);
void jh_cartpole_step_rows(jh_cartpole* e, int r0, int r1, const int64_t* h_action, float* h_next_obs, float* h_reward, uint8_t* h_done);
struct jh_control;
void jh_control_obs_rows(const jh_control* e, int r0, int r1, float* h_obs);
void jh_control_step_rows(jh_control* e, int r0, int r1, const float* h_action, float* h_next_obs, float* h_reward, uint8_t* h_done);

struct jh_pponet {
  jh_ctx* ctx = nullptr;
  int S = 0, H = 0, A = 0, cont = 0, max_rows = 0;
  int n_out = 0;  // head outputs: A + 1 (discrete) | 2 A + 1 (continuous)
  int gld = 8;    // row width of g_all: 8, or n_out rounded up to 4 beyond 8 outputs (separate-call / tiled paths only)
  int64_t n_params = 0;
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;  // borrowed flat buckets
  // offsets into the flat buckets (state_dict order)
  int64_t o_w1, o_b1, o_w2, o_b2, o_wh0, o_bh0, o_wh1, o_bh1, o_wv, o_bv;
  // owned workspaces
  float *h1 = nullptr, *h2 = nullptr, *dh1 = nullptr, *dh2 = nullptr;
  float* g_all = nullptr;     // [max_rows][gld] packed head gradients (A-operand of the dW_heads GEMM)
  float* dv2 = nullptr;       // [min(max_rows, 1024)] value gradients of the critic's second branch (data-parallel exact critic) + 8 floats:
  float* stats_tmp = nullptr; // the loss kernel's local statistics row between jh_pponet_ppo_update_dp_begin and _end
  int max_act_rows = 0;
  // acting exchange area: pinned host memory mapped into the device address space
  float *obs_pin_h = nullptr, *obs_pin_d = nullptr;        // [max_act_rows][S] observations (host writes, kernel reads)
  float *part_pin_h = nullptr, *part_pin_d = nullptr;      // [H/16][max_act_rows][8] partial head outputs (kernel writes)
  unsigned *flag_pin_h = nullptr, *flag_pin_d = nullptr;   // [tiles] per-tile sequence words
  unsigned act_seq = 0;
  float *act_out_h = nullptr, *act_out_d = nullptr;        // [max_act_rows][n_out] raw heads of nets with more than 8 outputs (pinned + mapped)
  uint64_t act_seed = 0, act_ctr = 0;  // host-side counter-based sampling stream
  float* fwd_part = nullptr;      // [H/16][max_rows][8] per-column-tile partial head outputs (jh_ppo_mb.hip forward)
  float* part_w1 = nullptr;       // [min(max_rows,1024)/16][H*S + H] per-row-tile partial (dW1 | db1) sums
  float* ssq_part = nullptr;      // [(H/32)^2 + H/32] sums of squares of the gradient tiles written by jh_pmb_bwd's workgroups
  float* norm_partial = nullptr;  // [kNormSlots]
  int norm_slots = 0;             // > 0: the last backward left the global norm's sums of squares in norm_partial[0 .. norm_slots) (jh_mlp.hip: NormJob)
  float* hyper = nullptr;         // device: {lr, beta1, beta2, eps, step, bc1, bc2_sqrt, _}
  float* upd_ws = nullptr;    // jh_pponet_ppo_update_rows: raw heads, their gradients, the second critic branch, loss partials, {w1, w2}, ticket
  size_t upd_floats = 0;
  float* xg = nullptr;        // [max_rows][S] gathered observation rows (B operand of dW1 on the tiled engine)
  float* tg_ws = nullptr;
  size_t tg_ws_floats = 0;
  unsigned* tg_cnt = nullptr;
  int tg_cnt_slots = 0;
};

// ---------------------------------------------------------------- host-side action sampling (PPO.act, ppo.py:55-69)
// Counter-based splitmix64 streams keyed by (net seed, timestep counter, env row[, dim]): the same actions
// whichever acting path (persistent kernel / one launch per step) produced the raw heads.
static inline uint64_t jh_mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline double jh_u01_of(uint64_t key) { return (double)(jh_mix64(key) >> 11) * (1.0 / 9007199254740992.0); }

// Categorical(softmax(z)).sample() by inverse CDF (torch.multinomial(pi, 1)); argmax when !training
static inline int64_t jh_sample_discrete(const jh_pponet* n, const float* z, int wq, int training) {
  const int A = n->A;
  int act = 0;
  float mx = z[0];
  for (int k = 1; k < A; ++k)
    if (z[k] > mx) { mx = z[k]; act = k; }
  if (!training) return act;
  float e[40], se = 0.f;  // (jh_pponet_create: <= 40 head outputs)
  for (int k = 0; k < A; ++k) { e[k] = expf(z[k] - mx); se += e[k]; }
  const float u = (float)jh_u01_of(n->act_seed * 0x100000001B3ull + n->act_ctr * 0x9E3779B97F4A7C15ull + (uint64_t)wq) * se;
  float c = 0.f;
  act = A - 1;
  for (int k = 0; k < A; ++k) {
    c += e[k];
    if (u < c) { act = k; break; }
  }
  return act;
}

// z = (mu_raw[A], log_std_raw[A]): mu = clamp(mu_raw, -5, 5), std = exp(tanh(log_std_raw)) (policy_value.py:54-56);
// action = tanh(Normal(mu, std).sample()) when training (Box-Muller on two counter-based uniforms), tanh(mu) otherwise
static inline void jh_sample_continuous(const jh_pponet* n, const float* z, int wq, int training, float* action) {
  const int A = n->A;
  for (int k = 0; k < A; ++k) {
    const float mu = fminf(fmaxf(z[k], -5.f), 5.f);
    float v = mu;
    if (training) {
      const float sd = expf(tanhf(z[A + k]));
      const uint64_t key = n->act_seed * 0x100000001B3ull + n->act_ctr * 0x9E3779B97F4A7C15ull + (uint64_t)wq * 64u + (uint64_t)k;
      const double u1 = 1.0 - jh_u01_of(key * 2 + 1), u2 = jh_u01_of(key * 2 + 2);  // u1 in (0, 1]
      v = mu + sd * (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    }
    action[k] = tanhf(v);
  }
}

// ---------------------------------------------------------------- optimizer hyper block (device memory, JH_HY_FLOATS floats)
// torch.optim derives (1 - beta) and the bias corrections in Python DOUBLE arithmetic and hands the results to its fp32 kernels as
// scalars; computing them from the fp32-rounded betas instead ((1.f - 0.999f) = 9.99987e-4) puts a 1.3e-5 relative error into
// exp_avg_sq -- found by the float64 optimizer tests of round 4.  The block therefore carries the betas as doubles next to the fp32
// values, and the derived scalars are computed in double on the host (1 - beta) and by ONE device thread per step (bias corrections).
enum {
  JH_HY_LR = 0, JH_HY_B1 = 1, JH_HY_B2 = 2, JH_HY_EPS = 3, JH_HY_STEP = 4,
  JH_HY_BC1 = 5,       // jh_pponet: 1 - beta1^t          | jh_rbnet: centered flag
  JH_HY_BC2S = 6,      // jh_pponet: sqrt(1 - beta2^t)
  JH_HY_OMB1 = 7,      // (float)(1.0 - beta1)  (RMSprop: 1 - alpha)
  JH_HY_OMB2 = 8,      // (float)(1.0 - beta2)
  JH_HY_B1D = 10,      // beta1 as a double (two floats, 8-byte aligned)
  JH_HY_B2D = 12,      // beta2 as a double
  JH_HY_FLOATS = 16
};
static inline void jh_hyper_fill(float* h, double lr, double beta1, double beta2, double eps, double step) {
  h[JH_HY_LR] = (float)lr; h[JH_HY_B1] = (float)beta1; h[JH_HY_B2] = (float)beta2; h[JH_HY_EPS] = (float)eps; h[JH_HY_STEP] = (float)step;
  h[JH_HY_BC1] = 0.f; h[JH_HY_BC2S] = 0.f;
  h[JH_HY_OMB1] = (float)(1.0 - beta1); h[JH_HY_OMB2] = (float)(1.0 - beta2); h[9] = 0.f;
  memcpy(h + JH_HY_B1D, &beta1, 8); memcpy(h + JH_HY_B2D, &beta2, 8);
  h[14] = h[15] = 0.f;
}

// ---------------------------------------------------------------- device helpers (wave = 64)
#ifdef __HIPCC__
// Adam's bias corrections for step t the way torch computes them: 1 - beta^t and its square root in DOUBLE, then fp32.
__device__ __forceinline__ void jh_adam_bias_corrections(const float* hyper, float t, float& bc1, float& bc2s) {
  const double b1 = *reinterpret_cast<const double*>(hyper + JH_HY_B1D), b2 = *reinterpret_cast<const double*>(hyper + JH_HY_B2D);
  bc1 = (float)(1.0 - pow(b1, (double)t));
  bc2s = (float)sqrt(1.0 - pow(b2, (double)t));
}
// jh_pponet: advance the step counter and store the step's bias corrections (ONE thread of ONE kernel per optimizer step)
__device__ __forceinline__ void jh_adam_advance(float* hyper) {
  const float t = hyper[JH_HY_STEP] + 1.f;
  float bc1, bc2s;
  jh_adam_bias_corrections(hyper, t, bc1, bc2s);
  hyper[JH_HY_STEP] = t;
  hyper[JH_HY_BC1] = bc1;
  hyper[JH_HY_BC2S] = bc2s;
}
__device__ __forceinline__ float jh_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double jh_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float jh_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float jh_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double jh_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide reductions for blockDim.x <= 1024 (<= 16 waves).  `red` is 16 floats of LDS.
// Deterministic: fixed shuffle tree inside a wave, then wave partials summed in wave order.
template <typename T, typename Op>
__device__ __forceinline__ T jh_block_reduce(T v, T* red, Op op, T ident) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  __syncthreads();  // protect `red` from a previous use
  if (lane == 0) red[wid] = v;
  __syncthreads();
  T r = ident;
  for (int w = 0; w < nw; ++w) r = op(r, red[w]);
  return r;
}
struct JhAdd { template <typename T> __device__ T operator()(T a, T b) const { return a + b; } };
struct JhMax { template <typename T> __device__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct JhMin { template <typename T> __device__ T operator()(T a, T b) const { return a < b ? a : b; } };
#endif
