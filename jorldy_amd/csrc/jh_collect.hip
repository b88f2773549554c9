// Native sync-mode rollout collection for PPO (discrete policy on CartPole, continuous policy on the synthetic
// control env) -- the whole of
// DistributedManager.run + Actor.run (manager/distributed_manager.py:26-31,76-92) for W workers and
// T steps in ONE call, no Python per step:
//   for t in 0..T-1:
//     jh_pponet_act_discrete      -> observations into device-mapped pinned memory, ONE launch (fused
//                                    MLP forward, partial heads + per-tile flags written back into
//                                    pinned memory), host polls the flags and samples the actions
//     jh_cartpole_step            -> host physics for all W envs, auto-reset
//     transition (s, a, r, s', d) -> pinned staging slab of the rollout store, worker-major
//   one hipMemcpyAsync per column moves the W*T transitions into the GPU-resident store.
// Weights never leave HBM (the acting network IS the learner's), so BaseAgent.sync_out / sync_in
// (core/agent/base.py:75-85) and the per-iteration state_dict broadcast disappear.
#include <chrono>

#include "jh_common.h"

// jh_persist.hip: one persistent acting kernel per rollout
struct jh_persist;
int jh_persist_create(jh_pponet* n, jh_persist** out);
void jh_persist_destroy(jh_persist* p);
int jh_persist_begin(jh_persist* p, int W, int T, hipStream_t st, int split = 0);
unsigned jh_persist_publish(jh_persist* p, int W, const float* h_obs);
unsigned jh_persist_seq(const jh_persist* p);
void jh_persist_publish_rows(jh_persist* p, int r0, int r1, const float* h_obs, unsigned tag);
int jh_persist_collect_range(jh_persist* p, int r0, int n_rows, unsigned tag, float* h_heads);
int jh_persist_collect(jh_persist* p, int W, unsigned tag, float* h_heads);
int jh_persist_collect_rows(jh_persist* p, const int* rows, int n_rows, unsigned tag, float* h_heads);
int jh_persist_heads(const jh_persist* p);
void jh_persist_abort(jh_persist* p);
bool jh_persist_gave_up(const jh_persist* p);
void jh_persist_dump_debug(jh_persist* p, int T);

namespace {
struct Level {  // 2^k W forked envs: row 2 i + a = env i of the level above after action a
  void* env = nullptr;
  int S = 0;
  std::vector<float> next, obs, rw;
  std::vector<uint8_t> dn;
  std::vector<int64_t> act;
  void init(int rows, int S_) {  // (keeps its buffers across runs: resize is a no-op from the second run on)
    S = S_;
    next.resize((size_t)S * rows); obs.resize((size_t)S * rows); rw.resize(rows); dn.resize(rows); act.resize(rows);
    for (int i = 0; i < rows; ++i) act[i] = i & 1;
  }
};
// dst rows 2 i + a <- src row i stepped with action a (auto-reset included: the row then holds the reset state)
// -> the env table's first error (jh_env_vtbl: obs / step return a negative JH_ERR_* and the run stops -- ADVICE r5: the speculative
// steps dropped it, and a failing user env kept feeding stale rows into the store)
inline int fork_step(const jh_env_vtbl& vt, Level& dst, const void* src, int n_src) {
  for (int i = 0; i < n_src; ++i) { vt.copy_row(dst.env, 2 * i, src, i); vt.copy_row(dst.env, 2 * i + 1, src, i); }
  int rc = vt.step(dst.env, 0, 2 * n_src, dst.act.data(), dst.next.data(), dst.rw.data(), dst.dn.data());
  if (rc) return rc;
  return vt.obs(dst.env, 0, 2 * n_src, dst.obs.data());
}
inline void copy_level_row(const jh_env_vtbl& vt, Level& d, int di, const Level& s_, int si) {
  vt.copy_row(d.env, di, s_.env, si);
  const size_t S = (size_t)d.S;
  memcpy(&d.next[S * (size_t)di], &s_.next[S * (size_t)si], sizeof(float) * S);
  memcpy(&d.obs[S * (size_t)di], &s_.obs[S * (size_t)si], sizeof(float) * S);
  d.rw[di] = s_.rw[si];
  d.dn[di] = s_.dn[si];
}
}  // namespace

struct jh_collector {
  jh_ctx* ctx = nullptr;
  jh_pponet* net = nullptr;
  // the host envs behind a table of functions (include/jorldy_hip.h: jh_env_vtbl; round 5): the two built-in envs
  // (jh_cartpole_*, jh_control_*) are two such tables, a C user plugs any other with jh_collector_create_env
  jh_env_vtbl vt{};
  void* env = nullptr;
  jh_store* store = nullptr;
  int W = 0, S = 0, A = 0;
  bool cont = false;
  int col_state = 0, col_action = 1, col_reward = 2, col_next = 3, col_done = 4;
  std::vector<float> obs, next_obs, reward, heads, act_f;
  std::vector<int64_t> act_i;
  std::vector<uint8_t> done;
  double dbg_publish = 0.0, dbg_wait = 0.0;  // JH_COLLECT_DEBUG
  long dbg_n = 0;
  jh_persist* persist = nullptr;  // null: one launch per timestep
  int mode = 1;                   // 1: persistent acting kernel (default), 0: launch per step (JH_COLLECT_PERSISTENT=0)
  // lookahead = 2 (discrete two-action envs that can be forked on the host, 3 W <= 32 rows): every PCIe round trip carries each
  // env's state AND both successor states, and serves two timesteps (run_loop_lookahead).  1: one timestep per round trip.
  int lookahead = 1;
  // round 6: 32 rows of an env that steps row ranges (the built-in envs) are exchanged as two INDEPENDENT halves of 16 (run_loop_split): one half's round
  // trip through the acting kernel runs under the other half's sampling, env steps and bookkeeping.  JH_COLLECT_SPLIT=0: one exchange of 32 rows.
  bool split = false;
  Level lv[4];  // L1, L2, L3 and the one-step exchange's scratch level
  void *spec = nullptr, *spec2 = nullptr, *spec3 = nullptr;  // 2 W / 4 W / 8 W scratch envs (vt.fork_alloc): successors one, two and three steps ahead
  // acting-time capture (jh_collector_set_capture): device destinations of the raw heads / values of the states acted on
  float *cap_h0 = nullptr, *cap_h1 = nullptr, *cap_v = nullptr, *cap_nv = nullptr;
  int64_t cap_rows = 0;
  // up to two more plain copies riding in the commit launch (jh_collector_set_ride_along): device-visible source -> device
  const void* ride_src[2] = {nullptr, nullptr};
  void* ride_dst[2] = {nullptr, nullptr};
  int64_t ride_bytes[2] = {0, 0};
  void* run = nullptr;            // RunState of the run in progress (jh_collector_run, or between _begin and _loop)
  unsigned *gate_h = nullptr, *gate_d = nullptr, gate_seq = 0;  // flag word that releases an early-enqueued commit launch
  int prelaunched_T = 0;          // steps of a persistent kernel already enqueued by jh_collector_prelaunch (0: none)
  double t_act = 0, t_env = 0, t_total = 0;  // host seconds: waiting for actions / stepping envs / whole runs
  double t_first = 0, t_extra = 0, t_commit = 0;  // of t_act: the rollout's first step (kernel start-up) and the value-only query; the commit launch
  int64_t runs = 0;
  int64_t steps = 0;
};

struct RunState;
static void run_state_alloc(jh_collector* c);
static void run_state_free(jh_collector* c);

static int cart_obs(void* e, int32_t r0, int32_t r1, float* o);
static int ctl_obs(void* e, int32_t r0, int32_t r1, float* o);

static int collector_create(jh_ctx* ctx, jh_pponet* net, const jh_env_vtbl* vt, void* env, jh_store* store, const int32_t* cols,
                            jh_collector** out) {
  JH_ARG(ctx && net && vt && env && store && cols && out);
  JH_ARG(vt->obs && vt->step && vt->W > 0 && vt->S > 0 && vt->A > 0);
  const int W = vt->W, S = vt->S, A = vt->A;
  const bool cont = vt->continuous != 0;
  JH_ARG(net->S == S && net->A == A && (net->cont != 0) == cont);
  JH_ARG(W <= net->max_act_rows);
  for (int i = 0; i < 5; ++i) JH_ARG(cols[i] >= 0 && cols[i] < store->n_cols);
  JH_ARG(store->cols[cols[0]].dtype == JH_F32 && store->cols[cols[0]].elems == S);
  if (cont) JH_ARG(store->cols[cols[1]].dtype == JH_F32 && store->cols[cols[1]].elems == A);
  else JH_ARG(store->cols[cols[1]].dtype == JH_I64 && store->cols[cols[1]].elems == 1);
  JH_ARG(store->cols[cols[2]].dtype == JH_F32 && store->cols[cols[2]].elems == 1);
  JH_ARG(store->cols[cols[3]].dtype == JH_F32 && store->cols[cols[3]].elems == S);
  JH_ARG(store->cols[cols[4]].dtype == JH_U8 && store->cols[cols[4]].elems == 1);
  JH_HIP(hipSetDevice(ctx->device));
  jh_collector* c = new jh_collector();
  c->ctx = ctx; c->net = net; c->vt = *vt; c->env = env; c->store = store; c->W = W; c->S = S; c->A = A; c->cont = cont;
  c->col_state = cols[0]; c->col_action = cols[1]; c->col_reward = cols[2]; c->col_next = cols[3]; c->col_done = cols[4];
  if (const char* e = getenv("JH_COLLECT_PERSISTENT")) c->mode = atoi(e);
  if (c->mode == 1 && W <= 32 && W * S <= 512) {  // jh_persist.hip: two row tiles of 16, up to four poll instructions of 128 observation granules
    if (jh_persist_create(net, &c->persist) != JH_OK) c->persist = nullptr;  // unsupported width: one launch per step
  }
  // Two timesteps per exchange is a CAPABILITY of the env (it can be forked on the host: fork_alloc / fork_free / copy_row given) with
  // two discrete actions, not a type: any env that offers it gets it; one that cannot be copied runs one timestep per exchange
  const bool forkable = vt->fork_alloc && vt->fork_free && vt->copy_row;
  if (c->persist && forkable && !cont && A == 2 && 3 * W <= 32 && 3 * W * S <= 128) {
    const char* e = getenv("JH_COLLECT_LOOKAHEAD");
    c->lookahead = e ? (atoi(e) >= 2 ? 2 : 1) : 2;
    if (c->lookahead == 2) {
      c->spec = vt->fork_alloc(env, 2 * W); c->spec2 = vt->fork_alloc(env, 4 * W); c->spec3 = vt->fork_alloc(env, 8 * W);
      if (!c->spec || !c->spec2 || !c->spec3) c->lookahead = 1;
    }
  }
  // two independent halves per timestep: 32 rows, one timestep per exchange, an env that steps row ranges of ITSELF (the vtbl's contract promises that for
  // scratch envs only: the built-in envs do)
  {
    const char* e = getenv("JH_COLLECT_SPLIT");
    c->split = c->persist && c->lookahead != 2 && W == 32 && (vt->obs == cart_obs || vt->obs == ctl_obs) && !(e && atoi(e) == 0);
  }
  c->obs.resize((size_t)S * W);
  c->next_obs.resize((size_t)S * W);
  c->act_i.resize(W);
  c->act_f.resize((size_t)A * W);
  c->reward.resize(W);
  c->done.resize(W);
  c->heads.resize(16 * (size_t)W);
  run_state_alloc(c);
  *out = c;
  return JH_OK;
}

// ---- the two built-in envs as function tables
static int cart_obs(void* e, int32_t r0, int32_t r1, float* o) { jh_cartpole_obs_rows((const jh_cartpole*)e, r0, r1, o); return JH_OK; }
static int cart_step(void* e, int32_t r0, int32_t r1, const void* a, float* nx, float* rw, uint8_t* dn) {
  jh_cartpole_step_rows((jh_cartpole*)e, r0, r1, (const int64_t*)a, nx, rw, dn);
  return JH_OK;
}
static void* cart_fork_alloc(void*, int32_t rows) {
  jh_cartpole* e = nullptr;
  return jh_cartpole_create(rows, 0, &e) == JH_OK ? e : nullptr;
}
static void cart_fork_free(void* e) { jh_cartpole_destroy((jh_cartpole*)e); }
static void cart_copy_row(void* d_, int32_t di, const void* s_, int32_t si) {
  jh_cartpole* d = (jh_cartpole*)d_;
  const jh_cartpole* s = (const jh_cartpole*)s_;
  memcpy(&d->s[4 * (size_t)di], &s->s[4 * (size_t)si], sizeof(double) * 4);
  d->t[di] = s->t[si];
  d->rng[di] = s->rng[si];
}
static int ctl_obs(void* e, int32_t r0, int32_t r1, float* o) {
  const jh_control* c = (const jh_control*)e;
  if (r0 < 0 || r1 > c->W || r0 >= r1) return jh_fail(JH_ERR_ARG, "jh_control: rows %d..%d of %d", r0, r1, c->W);
  jh_control_obs_rows(c, r0, r1, o);
  return JH_OK;
}
static int ctl_step(void* e, int32_t r0, int32_t r1, const void* a, float* nx, float* rw, uint8_t* dn) {
  jh_control* c = (jh_control*)e;
  if (r0 < 0 || r1 > c->W || r0 >= r1) return jh_fail(JH_ERR_ARG, "jh_control: rows %d..%d of %d", r0, r1, c->W);
  jh_control_step_rows(c, r0, r1, (const float*)a, nx, rw, dn);
  return JH_OK;
}

JH_EXPORT int jh_collector_create(jh_ctx* ctx, jh_pponet* net, jh_cartpole* env, jh_store* store,
                                  const int32_t* cols /* state, action, reward, next_state, done */,
                                  jh_collector** out) {
  JH_ARG(env != nullptr);
  const jh_env_vtbl vt{env->W, 4, 2, 0, cart_obs, cart_step, cart_fork_alloc, cart_fork_free, cart_copy_row};
  return collector_create(ctx, net, &vt, env, store, cols, out);
}

// The same collector for a continuous-action policy on the synthetic control env (config.ppo.mujoco shapes):
// store columns state f32[S], action f32[A], reward f32[1], next_state f32[S], done u8[1].
JH_EXPORT int jh_collector_create_control(jh_ctx* ctx, jh_pponet* net, jh_control* env, jh_store* store, const int32_t* cols,
                                          jh_collector** out) {
  JH_ARG(env != nullptr);
  const jh_env_vtbl vt{env->W, env->S, env->A, 1, ctl_obs, ctl_step, nullptr, nullptr, nullptr};
  return collector_create(ctx, net, &vt, env, store, cols, out);
}

// ANY host env behind the table (a C user's simulator, a test's callbacks): see include/jorldy_hip.h
JH_EXPORT int jh_collector_create_env(jh_ctx* ctx, jh_pponet* net, const jh_env_vtbl* vt, void* env, jh_store* store, const int32_t* cols,
                                      jh_collector** out) {
  return collector_create(ctx, net, vt, env, store, cols, out);
}

JH_EXPORT void jh_collector_destroy(jh_collector* c) {
  if (!c) return;
  if (c->persist) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipDeviceSynchronize();
    jh_persist_destroy(c->persist);
  }
  if (c->gate_h) (void)hipHostFree(c->gate_h);
  if (c->spec) c->vt.fork_free(c->spec);
  if (c->spec2) c->vt.fork_free(c->spec2);
  if (c->spec3) c->vt.fork_free(c->spec3);
  run_state_free(c);
  delete c;
}

// Acting-time capture.  PPO.learn starts with two no-grad passes of the SAME network over the rollout it was just handed
// (core/agent/ppo.py:83-94: pi, value = network(state); next_value = network(next_state)[-1]); in sync mode the weights the
// actors acted with are the learner's, so those numbers already existed when the actions were sampled.  With capture set the
// collector keeps, per transition row (worker-major, like the store): the raw policy head(s) of state_t, V(state_t), and
// V(next_state_t) = V(state_{t+1}) (the value acted on one step later; the rollout's last step gets one extra value-only
// query).  Where done_t is set, next_state_t is the terminal observation and state_{t+1} the reset one: GAE multiplies that
// entry by (1 - done_t) = 0 (ppo.py:96), so it never enters a result.  At the end of jh_collector_run the captured block is
// copied to d_h0 [rows][A], d_h1 [rows][A] (continuous policies: log_std_raw; NULL otherwise), d_value [rows], d_next_value [rows]
// in the SAME launch that commits the rollout rows.  rows must equal W * T of the runs that follow; d_value == NULL switches
// capture off.  Values differ from the learner's own pass only by fp32 summation order (~1e-7).
JH_EXPORT int jh_collector_set_capture(jh_collector* c, float* d_h0, float* d_h1, float* d_value, float* d_next_value, int64_t rows) {
  JH_ARG(c != nullptr);
  if (!d_value) { c->cap_h0 = c->cap_h1 = c->cap_v = c->cap_nv = nullptr; c->cap_rows = 0; return JH_OK; }
  JH_ARG(d_h0 && d_next_value && rows > 0 && (!c->cont || d_h1));
  c->cap_h0 = d_h0; c->cap_h1 = c->cont ? d_h1 : nullptr; c->cap_v = d_value; c->cap_nv = d_next_value; c->cap_rows = rows;
  return JH_OK;
}

// Two more copies for the commit launch of the NEXT run only (slot 0 / 1; one-shot: the launch that performs a copy clears its
// slot; bytes == 0 clears a slot before that): the learner's inputs that
// change between learn() calls but are known before the rollout ends -- the minibatch index lists of the coming epochs (drawn
// ahead, np_rng.Predraw) and the decayed learning rate -- go from device-mapped pinned memory to their device buffers inside the
// launch that exists anyway, instead of as hipMemcpyAsync calls of their own (SDMA hand-offs of ~10 us each between the learner's
// last kernel and the next rollout's acting kernel).  The source is read when the commit kernel runs (end of the run).
JH_EXPORT int jh_collector_set_ride_along(jh_collector* c, int32_t slot, const void* d_src_mapped, void* d_dst, int64_t bytes) {
  JH_ARG(c != nullptr && slot >= 0 && slot < 2 && bytes >= 0 && (bytes == 0 || (d_src_mapped && d_dst)));
  c->ride_src[slot] = d_src_mapped; c->ride_dst[slot] = d_dst; c->ride_bytes[slot] = bytes;
  return JH_OK;
}

// exchanges with the acting kernel per run: T timesteps (two per exchange with lookahead) + the value-only query of a capturing run
static int collector_steps(const jh_collector* c, int T) {
  const int extra = (c->cap_v && c->cap_rows == (int64_t)c->W * T) ? 1 : 0;
  return (c->lookahead == 2 ? (T + 1) / 2 : T) + extra;
}
static int collector_rows(const jh_collector* c) { return c->lookahead == 2 ? 3 * c->W : c->W; }

// Enqueue the persistent acting kernel of the NEXT jh_collector_run(T) now (e.g. right behind the learner's last launch): it starts
// when the stream reaches it, loads the then-current weights and waits for the first observations (bounded: ~0.2 s, after which it
// exits and the run falls back to a fresh launch).  Takes the kernel's launch + start-up latency out of the host's critical path
// between learn() and the next rollout.  Nothing else may be enqueued on `stream` until that run (it would wait for the rollout).
JH_EXPORT int jh_collector_prelaunch(jh_collector* c, int32_t T, jh_stream stream) {
  JH_ARG(c != nullptr && T > 0);
  if (!c->persist || c->prelaunched_T) return JH_OK;
  const int steps = collector_steps(c, T);
  int rc = jh_persist_begin(c->persist, collector_rows(c), steps, jh_s(stream), c->split);
  if (rc == JH_OK) c->prelaunched_T = steps;
  return rc;
}

// ---- one run = prepare (staging slabs, acting kernel) -> host loop -> commit (rows + captured block + ride-along copies, ONE launch).
// jh_collector_run does the three in that order.  jh_collector_begin / jh_collector_loop enqueue the commit launch FIRST, gated by a
// flag word in device-mapped pinned memory that the host loop sets when its last row is written: whatever the caller enqueues next
// (the learner's graph) then runs the instant the rollout ends, instead of after the host has returned, crossed back into Python
// and launched it (~45 us of idle GPU per iteration at config.ppo.cartpole shapes).
struct RunState {
  bool active = false, cap = false, persistent = false, early = false;
  int T = 0, steps = 0;
  int64_t n = 0;
  void* cols[16] = {};
  jh_pinned_slab* cap_slab = nullptr;
  float *ch0 = nullptr, *ch1 = nullptr, *cv = nullptr, *cnv = nullptr;
};
static RunState& run_state(jh_collector* c) { return *static_cast<RunState*>(c->run); }
static void run_state_alloc(jh_collector* c) { c->run = new RunState(); }
static void run_state_free(jh_collector* c) { delete static_cast<RunState*>(c->run); c->run = nullptr; }

static int run_commit(jh_collector* c, RunState& r, int rc_in, const unsigned* wait_flag, unsigned wait_val, hipStream_t st) {
  // commit what was staged (keeps the store consistent) and hand the capture slab back.  A run that FAILED (an env error, the acting
  // kernel gone with work queued behind it) appends nothing: its staging rows are half written
  if (rc_in != JH_OK && wait_flag == nullptr) {
    (void)jh_store_stage_abort(c->store, st);
    if (r.cap_slab) (void)jh_ctx_slab_release(c->ctx, r.cap_slab, st);
    r.cap_slab = nullptr;
    for (int q = 0; q < 2; ++q) { c->ride_src[q] = nullptr; c->ride_dst[q] = nullptr; c->ride_bytes[q] = 0; }
    return rc_in;
  }
  const int A = c->A;
  const int64_t n = r.n;
  const void* xs[6]; void* xd[6]; int64_t xb[6]; int k = 0;
  if (r.cap && rc_in == JH_OK) {
    const char* dev0 = (const char*)r.cap_slab->dev;
    auto job = [&](const float* h, float* d, size_t floats) { xs[k] = dev0 + ((const char*)h - (const char*)r.cap_slab->host); xd[k] = d; xb[k] = (int64_t)(sizeof(float) * floats); ++k; };
    job(r.ch0, c->cap_h0, (size_t)n * A);
    if (c->cont) job(r.ch1, c->cap_h1, (size_t)n * A);
    job(r.cv, c->cap_v, (size_t)n);
    job(r.cnv, c->cap_nv, (size_t)n);
  }
  // ride-along copies are ONE-SHOT: consumed by this commit launch (a registration that persisted across runs kept
  // pointing at index-list buffers the agent had meanwhile reallocated, and re-delivered a stale learning rate)
  for (int q = 0; q < 2; ++q) {
    if (rc_in == JH_OK && c->ride_bytes[q] > 0) { xs[k] = c->ride_src[q]; xd[k] = c->ride_dst[q]; xb[k] = c->ride_bytes[q]; ++k; }
    c->ride_src[q] = nullptr; c->ride_dst[q] = nullptr; c->ride_bytes[q] = 0;
  }
  const int rc2 = jh_store_stage_commit_gated(c->store, k, xs, xd, xb, wait_flag, wait_val, st);
  if (r.cap_slab) (void)jh_ctx_slab_release(c->ctx, r.cap_slab, st);
  r.cap_slab = nullptr;
  return rc_in ? rc_in : rc2;
}

static int run_prepare(jh_collector* c, int T, hipStream_t st) {
  if (c->gate_h) {  // did the commit launch of an EARLIER run give up waiting for its release (it copied nothing then)?
    const unsigned lost = __atomic_load_n(c->gate_h + 1, __ATOMIC_ACQUIRE);
    if (lost) {
      __atomic_store_n(c->gate_h + 1, 0u, __ATOMIC_RELEASE);
      return jh_fail(JH_ERR_STATE, "the commit launch of an earlier run (tag %u) timed out waiting for its release: that rollout never reached the store and the "
                                   "learner behind it ran on stale rows -- the agent's weights are not trustworthy", lost);
    }
  }
  RunState& r = run_state(c);
  const int W = c->W, A = c->A;
  r = RunState();
  r.T = T;
  r.n = (int64_t)W * T;
  r.cap = c->cap_v && c->cap_rows == r.n;
  r.steps = collector_steps(c, T);
  int rc = jh_store_stage_begin(c->store, r.n, r.cols);
  if (rc) return rc;
  // captured block in pinned, device-mapped memory: [h0 n*A | h1 n*A (continuous) | value n | next_value n]
  if (r.cap) {
    const size_t fl = (size_t)r.n * A * (c->cont ? 2 : 1) + 2 * (size_t)r.n;
    rc = jh_ctx_slab(c->ctx, sizeof(float) * fl, &r.cap_slab);
    if (rc) { (void)jh_store_stage_commit(c->store, st); return rc; }
    r.ch0 = (float*)r.cap_slab->host;
    r.ch1 = c->cont ? r.ch0 + (size_t)r.n * A : nullptr;
    r.cv = r.ch0 + (size_t)r.n * A * (c->cont ? 2 : 1);
    r.cnv = r.cv + r.n;
  }
  r.persistent = c->persist != nullptr;
  if (r.persistent) {
    const int pre = c->prelaunched_T;
    c->prelaunched_T = 0;
    if (pre != r.steps || jh_persist_gave_up(c->persist)) {  // nothing prelaunched, another length, or it timed out waiting: launch now
      if (pre && !jh_persist_gave_up(c->persist)) {          // a live kernel of another length: stop it first
        jh_persist_abort(c->persist);
        (void)hipStreamSynchronize(st);
      }
      rc = jh_persist_begin(c->persist, collector_rows(c), r.steps, st, c->split);
      if (rc) r.persistent = false;
    }
  }
  r.active = true;
  return JH_OK;
}

static int run_loop_lookahead(jh_collector* c, int training, hipStream_t stream_h, int* t_done);
static int run_loop_split(jh_collector* c, int training, hipStream_t stream_h, int* t_done);

// The host loop.  Returns the first error; the caller commits either way.
static int run_loop(jh_collector* c, int training, hipStream_t stream_h) {
  RunState& r = run_state(c);
  const int W = c->W, S = c->S, A = c->A, T = r.T, steps = T + (r.cap ? 1 : 0);
  const bool cap = r.cap;
  int t_begin = 0;
  if (r.persistent && c->lookahead == 2) {
    const int rc_la = run_loop_lookahead(c, training, stream_h, &t_begin);
    if (rc_la != JH_OK || t_begin >= steps) return rc_la;
    // the acting kernel gave up in the middle of the run: finish the remaining timesteps with one launch per step (below)
    r.persistent = false;
  } else if (r.persistent && c->split) {
    const int rc_sp = run_loop_split(c, training, stream_h, &t_begin);
    if (rc_sp != JH_OK || t_begin >= steps) return rc_sp;
    r.persistent = false;  // (the same)
  }
  jh_stream stream = (jh_stream)stream_h;
  int rc = JH_OK;
  float *ch0 = r.ch0, *ch1 = r.ch1, *cv = r.cv, *cnv = r.cnv;
  float* st = (float*)r.cols[c->col_state];
  int64_t* ac_i = (int64_t*)r.cols[c->col_action];
  float* ac_f = (float*)r.cols[c->col_action];
  float* rw = (float*)r.cols[c->col_reward];
  float* ns = (float*)r.cols[c->col_next];
  uint8_t* dn = (uint8_t*)r.cols[c->col_done];
  bool persistent = r.persistent;
  const int no = c->persist ? jh_persist_heads(c->persist) : 0;  // policy heads + value
  const int n_pol = c->cont ? 2 * A : A;
  std::vector<float> val(W), lg((size_t)W * 2 * A);
  for (int t = t_begin; t < steps; ++t) {
    const bool extra = t == T;  // capture: one value-only query of the states the rollout ended in
    // current state of every env (reset state where it just finished)
    rc = c->vt.obs(c->env, 0, W, c->obs.data());
    if (rc) {
      if (persistent) jh_persist_abort(c->persist);
      return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (persistent) {
      static const bool dbg1 = getenv("JH_COLLECT_DEBUG") != nullptr;
      const auto d0 = dbg1 ? std::chrono::steady_clock::now() : t0;
      const unsigned tag = jh_persist_publish(c->persist, W, c->obs.data());
      const auto d1 = dbg1 ? std::chrono::steady_clock::now() : t0;
      rc = jh_persist_collect(c->persist, W, tag, c->heads.data());
      if (dbg1) {  // JH_COLLECT_DEBUG=1: publish / wait + read of an exchange, summed per collector (printed every 16th run below)
        const auto d2 = std::chrono::steady_clock::now();
        c->dbg_publish += std::chrono::duration<double>(d1 - d0).count();
        c->dbg_wait += std::chrono::duration<double>(d2 - d1).count();
        c->dbg_n += 1;
      }
      if (rc) {  // the kernel gave up (it exits by itself)
        jh_persist_abort(c->persist);
        if (r.early)  // the commit launch and the learner are already queued behind it on this stream: no per-step launches possible
          return jh_fail(JH_ERR_STATE, "the persistent acting kernel gave up at step %d of a run whose commit was enqueued ahead (jh_collector_begin): "
                                       "no observations for ~0.2 s; use jh_collector_run for environments that may stall", t);
        JH_HIP(hipStreamSynchronize(stream_h));  // finish this rollout with per-step launches
        persistent = false;
      } else {
        for (int w = 0; w < W; ++w) {
          const float* hz = c->heads.data() + (size_t)w * no;
          if (!extra) {
            if (c->cont) jh_sample_continuous(c->net, hz, w, training, c->act_f.data() + (size_t)w * A);
            else c->act_i[w] = jh_sample_discrete(c->net, hz, w, training);
          }
          memcpy(lg.data() + (size_t)w * n_pol, hz, sizeof(float) * n_pol);
          val[w] = hz[no - 1];
        }
        if (!extra) c->net->act_ctr += 1;
      }
    }
    if (!persistent) {
      if (r.early) return jh_fail(JH_ERR_STATE, "jh_collector_begin needs the persistent acting kernel (W <= 32, W * S <= 512, JH_COLLECT_PERSISTENT != 0)");
      // (the per-step launch samples even for the value-only query; its actions are not used)
      rc = c->cont ? jh_pponet_act_continuous(c->net, W, c->obs.data(), c->act_f.data(), lg.data(), lg.data() + (size_t)W * A, val.data(), training, stream)
                   : jh_pponet_act_discrete(c->net, W, c->obs.data(), c->act_i.data(), lg.data(), val.data(), training, stream);
      if (rc) return rc;
      if (extra) c->net->act_ctr -= 1;  // the value-only query must not consume a sampling step: same action stream with and without capture
      if (c->cont) {  // [W][A] mu | [W][A] log_std  ->  per-row [mu A | log_std A] like the persistent path
        std::vector<float> tmp(lg);
        for (int w = 0; w < W; ++w) {
          memcpy(lg.data() + (size_t)w * 2 * A, tmp.data() + (size_t)w * A, sizeof(float) * A);
          memcpy(lg.data() + (size_t)w * 2 * A + A, tmp.data() + (size_t)W * A + (size_t)w * A, sizeof(float) * A);
        }
      }
    }
    if (cap) {
      for (int w = 0; w < W; ++w) {
        if (t > 0) cnv[(size_t)w * T + (t - 1)] = val[w];  // V(next_state_{t-1}) = V(state_t)  (masked by done_{t-1} in GAE)
        if (!extra) {
          const size_t row = (size_t)w * T + t;
          cv[row] = val[w];
          memcpy(ch0 + row * A, lg.data() + (size_t)w * n_pol, sizeof(float) * A);
          if (c->cont) memcpy(ch1 + row * A, lg.data() + (size_t)w * n_pol + A, sizeof(float) * A);
        }
      }
    }
    if (extra) {
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      c->t_act += dt;
      c->t_extra += dt;
      break;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (t == 0) c->t_first += std::chrono::duration<double>(t1 - t0).count();
    rc = c->vt.step(c->env, 0, W, c->cont ? (const void*)c->act_f.data() : (const void*)c->act_i.data(), c->next_obs.data(), c->reward.data(), c->done.data());
    if (rc) {
      if (persistent) jh_persist_abort(c->persist);
      return rc;
    }
    for (int w = 0; w < W; ++w) {
      const size_t row = (size_t)w * T + t;  // worker-major: w0 t0..tT-1, w1 ...  (distributed_manager.py:30)
      memcpy(st + S * row, c->obs.data() + (size_t)S * w, sizeof(float) * S);
      memcpy(ns + S * row, c->next_obs.data() + (size_t)S * w, sizeof(float) * S);
      if (c->cont) memcpy(ac_f + A * row, c->act_f.data() + (size_t)A * w, sizeof(float) * A);
      else ac_i[row] = c->act_i[w];
      rw[row] = c->reward[w];
      dn[row] = c->done[w];
    }
    const auto t2 = std::chrono::steady_clock::now();
    c->t_act += std::chrono::duration<double>(t1 - t0).count();
    c->t_env += std::chrono::duration<double>(t2 - t1).count();
    c->steps += 1;
  }
  if (persistent && getenv("JH_PERSIST_DEBUG") && (c->runs % 16) == 15) jh_persist_dump_debug(c->persist, T);
  if (c->dbg_n > 0 && (c->runs % 4) == 3) {
    fprintf(stderr, "[jh_collect] one timestep per exchange, per step: publish %.2f us, wait + read of the heads %.2f us (sampling + bookkeeping + env step: the rest of act / env)\n",
            c->dbg_publish / c->dbg_n * 1e6, c->dbg_wait / c->dbg_n * 1e6);
    c->dbg_publish = c->dbg_wait = 0.0;
    c->dbg_n = 0;
  }
  return JH_OK;
}


// ---- split: the 32 rows of a timestep as two INDEPENDENT exchanges of 16 (round 6).
// One timestep of config.ppo.mujoco's 32 workers was exchange (19.4 us: publication, PCIe polling, 3.9 us of kernel, 48 KB of partial heads back, the host's
// sampling) + env steps and bookkeeping (11 us), one after the other.  The acting kernel's two row tiles are two sets of workgroups that never talk to each
// other, so with tags per row tile (jh_persist.hip: PersistArgs::split) the host runs them out of phase: while half A's observations are on their way through
// the kernel, the host samples, steps and stores half B, publishes B's next observations, and only then comes back for A's heads.  Nothing is speculated: the
// same rows go through the same arithmetic (a row's position inside its tile is unchanged), the sampling stream is keyed by (seed, timestep, row), every env
// row advances on its own state -- the rollout, the captured heads / values and every stored transition are bit-identical to the one-exchange path (tested).
// *t_done: timesteps completed (incl. the value-only query); < steps only when the kernel gave up (the caller finishes with one launch per step).
static int run_loop_split(jh_collector* c, int training, hipStream_t stream_h, int* t_done) {
  RunState& r = run_state(c);
  const int W = c->W, S = c->S, A = c->A, T = r.T, steps = T + (r.cap ? 1 : 0), HR = 16;
  const bool cap = r.cap;
  float *ch0 = r.ch0, *ch1 = r.ch1, *cv = r.cv, *cnv = r.cnv;
  float* st = (float*)r.cols[c->col_state];
  int64_t* ac_i = (int64_t*)r.cols[c->col_action];
  float* ac_f = (float*)r.cols[c->col_action];
  float* rw = (float*)r.cols[c->col_reward];
  float* ns = (float*)r.cols[c->col_next];
  uint8_t* dn = (uint8_t*)r.cols[c->col_done];
  const int no = jh_persist_heads(c->persist);
  const unsigned base = jh_persist_seq(c->persist);  // the kernel's tags are base + 1 .. base + steps, for each half on its own
  bool dead = false;  // the kernel gave up while half 1 of a timestep was owed: that half is finished by hand, then the per-step path takes over
  // test hook (tests/test_agents_gpu.py): JH_COLLECT_TEST_STALL="<timestep>,<half>" makes the host sit idle for 0.35 s in front of that half's read, which is what a
  // stalled environment looks like to the acting kernel (it gives up after ~0.2 s without observations): exercises the hand-over to one launch per step
  int stall_t = -1, stall_h = -1;
  if (const char* e = getenv("JH_COLLECT_TEST_STALL")) (void)sscanf(e, "%d,%d", &stall_t, &stall_h);
  *t_done = 0;
  int rc = c->vt.obs(c->env, 0, W, c->obs.data());
  if (rc) { jh_persist_abort(c->persist); return rc; }
  for (int h = 0; h < 2; ++h) jh_persist_publish_rows(c->persist, h * HR, (h + 1) * HR, c->obs.data(), base + 1);
  for (int t = 0; t < steps; ++t) {
    const bool extra = t == T;  // capture: one value-only query of the states the rollout ended in
    for (int h = 0; h < 2; ++h) {
      const int r0 = h * HR, r1 = r0 + HR;
      const auto t0 = std::chrono::steady_clock::now();
      if (t == stall_t && h == stall_h) {
        // (the OTHER half's workgroups are the ones starved: this half's observations are out already)
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.35) {}
      }
      rc = jh_persist_collect_range(c->persist, r0, HR, base + (unsigned)t + 1, c->heads.data() + (size_t)r0 * no);
      if (rc) {  // the kernel gave up (it exits by itself)
        jh_persist_abort(c->persist);
        if (r.early)
          return jh_fail(JH_ERR_STATE, "the persistent acting kernel gave up at step %d of a run whose commit was enqueued ahead (jh_collector_begin): "
                                       "no observations for ~0.2 s; use jh_collector_run for environments that may stall", t);
        JH_HIP(hipStreamSynchronize(stream_h));
        if (h == 0) { *t_done = t; return JH_OK; }  // both halves stand at step t: the per-step path takes over from here
        // half 1 of step t is still owed (half 0 has moved on): its raw heads from one launch over its 16 rows, sampled below with the rows' own stream keys
        std::vector<float> mu((size_t)HR * A), ls((size_t)HR * A), vv(HR);
        std::vector<float> af((size_t)HR * A);
        std::vector<int64_t> ai(HR);
        const uint64_t ctr = c->net->act_ctr;
        rc = c->cont ? jh_pponet_act_continuous(c->net, HR, c->obs.data() + (size_t)r0 * S, af.data(), mu.data(), ls.data(), vv.data(), 0, (jh_stream)stream_h)
                     : jh_pponet_act_discrete(c->net, HR, c->obs.data() + (size_t)r0 * S, ai.data(), mu.data(), vv.data(), 0, (jh_stream)stream_h);
        if (rc) return rc;
        c->net->act_ctr = ctr;  // (the launch counted a sampling step; this timestep's count comes below)
        dead = true;
        for (int k = 0; k < HR; ++k) {
          float* hz = c->heads.data() + (size_t)(r0 + k) * no;
          memcpy(hz, mu.data() + (size_t)k * A, sizeof(float) * A);
          if (c->cont) memcpy(hz + A, ls.data() + (size_t)k * A, sizeof(float) * A);
          hz[no - 1] = vv[k];
        }
      }
      for (int w = r0; w < r1; ++w) {
        const float* hz = c->heads.data() + (size_t)w * no;
        if (!extra) {
          if (c->cont) jh_sample_continuous(c->net, hz, w, training, c->act_f.data() + (size_t)w * A);
          else c->act_i[w] = jh_sample_discrete(c->net, hz, w, training);
        }
        if (cap) {
          const float val = hz[no - 1];
          if (t > 0) cnv[(size_t)w * T + (t - 1)] = val;  // V(next_state_{t-1}) = V(state_t)  (masked by done_{t-1} in GAE)
          if (!extra) {
            const size_t row = (size_t)w * T + t;
            cv[row] = val;
            memcpy(ch0 + row * A, hz, sizeof(float) * A);
            if (c->cont) memcpy(ch1 + row * A, hz + A, sizeof(float) * A);
          }
        }
      }
      if (h == 1 && !extra) c->net->act_ctr += 1;  // one sampling step per TIMESTEP: both halves drew under the same counter
      const auto t1 = std::chrono::steady_clock::now();
      c->t_act += std::chrono::duration<double>(t1 - t0).count();
      if (extra) { c->t_extra += std::chrono::duration<double>(t1 - t0).count(); continue; }
      if (t == 0) c->t_first += std::chrono::duration<double>(t1 - t0).count();
      rc = c->vt.step(c->env, r0, r1, c->cont ? (const void*)c->act_f.data() : (const void*)c->act_i.data(), c->next_obs.data(), c->reward.data(), c->done.data());
      if (rc) { jh_persist_abort(c->persist); return rc; }
      for (int w = r0; w < r1; ++w) {
        const size_t row = (size_t)w * T + t;  // worker-major: w0 t0..tT-1, w1 ...  (distributed_manager.py:30)
        memcpy(st + S * row, c->obs.data() + (size_t)S * w, sizeof(float) * S);
        memcpy(ns + S * row, c->next_obs.data() + (size_t)S * w, sizeof(float) * S);
        if (c->cont) memcpy(ac_f + A * row, c->act_f.data() + (size_t)A * w, sizeof(float) * A);
        else ac_i[row] = c->act_i[w];
        rw[row] = c->reward[w];
        dn[row] = c->done[w];
      }
      if (t + 1 < steps && !dead) {  // this half's next observations (the reset state where an episode just ended) go out NOW: their round trip runs under the other half's work
        rc = c->vt.obs(c->env, r0, r1, c->obs.data());
        if (rc) { jh_persist_abort(c->persist); return rc; }
        jh_persist_publish_rows(c->persist, r0, r1, c->obs.data(), base + (unsigned)t + 2);
      }
      c->t_env += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    }
    if (!extra) c->steps += 1;
    *t_done = t + 1;
    if (dead) return JH_OK;
  }
  return JH_OK;
}

// ---- lookahead = 2: two timesteps per exchange with the acting kernel.
// An acting step is latency: the host's observations cross PCIe (the GPU polls host memory: ~1.7 us), are relayed on the chip, go
// through ~1.6 us of kernel and come back as partial heads (~0.7 us) -- ~5 us per timestep of which the MFMA work is 0.4 us
// (DESIGN §9.1).  The envs are host objects with two actions: the collector steps COPIES of every env with both actions first and
// publishes three rows per env -- the state s_t and its two possible successors -- in ONE exchange (24 rows x 4 floats = 96 of the
// 128 granules for config.ppo.cartpole's 8 workers; the second 16-row tile is a second set of workgroups, jh_persist.hip).  When the
// heads come back, a_t is sampled from pi(.|s_t) exactly as before, the env takes the successor that action leads to (already
// computed, reward / done / reset included), and a_{t+1} is sampled at once from the heads of THAT row.
// While the GPU works on an exchange the host has nothing to do but poll, so it runs the env model further ahead in that window:
// the four two-step successors of every env and their eight successors (12 env steps per env, ~1.8 us for 8 envs inside a ~6 us
// wait).  When the heads arrive, the next exchange's 24 rows are picked from what is already computed: no env step is left on the
// critical path between two exchanges.
// Same policy evaluations at the visited states (the row position inside an MFMA tile does not change a row's arithmetic:
// bit-identical heads), same counter-based sampling stream, same env RNG streams (one per env, and a fork copies it) -> the
// rollout, the captured heads / values and every stored transition are bit-identical to the one-step-per-exchange path
// (tests/test_agents_gpu.py: lookahead vs JH_COLLECT_LOOKAHEAD=1); what changes is that the GPU also evaluates the successor that
// was not taken (8 wasted rows per exchange) and the host steps env copies that are thrown away.
// *t_done: timesteps (incl. the value-only query) completed; < steps only when the kernel gave up.
static int run_loop_lookahead(jh_collector* c, int training, hipStream_t stream_h, int* t_done) {
  RunState& r = run_state(c);
  const int W = c->W, S = c->S, T = r.T, steps = T + (r.cap ? 1 : 0);
  const bool cap = r.cap;
  void* e = c->env;
  const jh_env_vtbl& vt = c->vt;
  float *ch0 = r.ch0, *cv = r.cv, *cnv = r.cnv;
  float* st = (float*)r.cols[c->col_state];
  int64_t* ac_i = (int64_t*)r.cols[c->col_action];
  float* rw = (float*)r.cols[c->col_reward];
  float* ns = (float*)r.cols[c->col_next];
  uint8_t* dn = (uint8_t*)r.cols[c->col_done];
  const int no = jh_persist_heads(c->persist);  // A logits + value
  const int A = c->A;
  // the levels' buffers live in the collector (they were re-allocated on every run)
  Level &L1 = c->lv[0], &L2 = c->lv[1], &L3 = c->lv[2], &tmp = c->lv[3];
  L1.env = c->spec; L2.env = c->spec2; L3.env = c->spec3;
  L1.init(2 * W, S); L2.init(4 * W, S); L3.init(8 * W, S);
  std::vector<float> pub((size_t)3 * W * S), pub_next((size_t)3 * W * S), hz((size_t)W * no), hz2((size_t)W * no);
  std::vector<int> pick(W);
  std::vector<int64_t> a0(W), a1(W);
  int t = 0;
  *t_done = 0;
  // JH_COLLECT_DEBUG=1: where the host's time per exchange goes (ns, summed over the run, printed every 16th run): fork = stepping the
  // speculative copies, wait = polling for the root rows' heads, sample = both samplings + the chosen successors' heads,
  // publish = assembling and publishing the next exchange, book = transitions / captured heads / moving the envs on
  static const bool dbg = getenv("JH_COLLECT_DEBUG") != nullptr;
  double d_fork = 0, d_wait = 0, d_sample = 0, d_publish = 0, d_book = 0;
  int n_ex = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  // rows 0 .. W-1: the envs' current states; rows W + 2 w + a: the state env w acts on next if it takes action a now
  // an env error anywhere in this loop: stop the acting kernel (it would poll for ~0.2 s otherwise) and return the error -- the caller's
  // commit carries the abort value, nothing of this run reaches the store
  auto env_failed = [&](int rc_env) {
    jh_persist_abort(c->persist);
    return rc_env;
  };
  int rc_env = fork_step(vt, L1, e, W);
  if (!rc_env) rc_env = vt.obs(e, 0, W, pub.data());
  if (rc_env) return env_failed(rc_env);
  memcpy(pub.data() + (size_t)W * S, L1.obs.data(), sizeof(float) * (size_t)2 * W * S);
  auto t0 = std::chrono::steady_clock::now();
  unsigned tag = jh_persist_publish(c->persist, 3 * W, pub.data());
  while (t < steps) {
    const bool extra = t == T;
    const bool two = !extra && t + 1 < T;
    const int t_next = t + (two ? 2 : 1);
    const auto q0 = dbg ? now() : std::chrono::steady_clock::time_point();
    if (!extra) {  // the GPU is busy for ~5 us: run the env model two levels further meanwhile
      rc_env = fork_step(vt, L2, L1.env, 2 * W);
      if (!rc_env && two && t_next < steps) rc_env = fork_step(vt, L3, L2.env, 4 * W);
      if (rc_env) return env_failed(rc_env);
    }
    const auto q1 = dbg ? now() : q0;
    int rc = jh_persist_collect_rows(c->persist, nullptr, W, tag, hz.data());
    const auto q2 = dbg ? now() : q0;
    if (rc) {  // the kernel gave up (it exits by itself)
      jh_persist_abort(c->persist);
      if (r.early)
        return jh_fail(JH_ERR_STATE, "the persistent acting kernel gave up at step %d of a run whose commit was enqueued ahead (jh_collector_begin): "
                                     "no observations for ~0.2 s; use jh_collector_run for environments that may stall", t);
      JH_HIP(hipStreamSynchronize(stream_h));
      *t_done = t;
      return JH_OK;  // the caller finishes the run with one launch per step
    }
    if (!extra) {
      for (int w = 0; w < W; ++w) {
        a0[w] = jh_sample_discrete(c->net, hz.data() + (size_t)w * no, w, training);
        pick[w] = W + 2 * w + (int)a0[w];
      }
      c->net->act_ctr += 1;
      if (two) {  // timestep t + 1: the heads of the chosen successors arrived with the same exchange
        rc = jh_persist_collect_rows(c->persist, pick.data(), W, tag, hz2.data());
        if (rc) return rc;  // (the kernel answers a tag for all rows or for none)
        for (int w = 0; w < W; ++w) a1[w] = jh_sample_discrete(c->net, hz2.data() + (size_t)w * no, w, training);
        c->net->act_ctr += 1;
      }
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (dbg) { d_fork += secs(q0, q1); d_wait += secs(q1, q2); d_sample += secs(q2, t1); ++n_ex; }
    // ---- the next exchange first: its rows are already computed; everything else happens while the GPU works on it
    unsigned tag_next = 0;
    auto t0_next = t1;
    if (!extra && t_next < steps) {
      for (int w = 0; w < W; ++w) {
        const int k1 = 2 * w + (int)a0[w];
        if (two) {
          const int k2 = 2 * k1 + (int)a1[w];
          memcpy(&pub_next[(size_t)S * w], &L2.obs[(size_t)S * k2], sizeof(float) * S);
          memcpy(&pub_next[(size_t)S * (W + 2 * w)], &L3.obs[(size_t)S * 2 * k2], sizeof(float) * 2 * S);
        } else {
          memcpy(&pub_next[(size_t)S * w], &L1.obs[(size_t)S * k1], sizeof(float) * S);
          memcpy(&pub_next[(size_t)S * (W + 2 * w)], &L2.obs[(size_t)S * 2 * k1], sizeof(float) * 2 * S);
        }
      }
      t0_next = std::chrono::steady_clock::now();
      tag_next = jh_persist_publish(c->persist, 3 * W, pub_next.data());
    }
    const auto q3 = dbg ? now() : t1;
    if (dbg) d_publish += secs(t1, q3);
    // ---- bookkeeping of this exchange: captured heads / values, transitions, the envs move on
    if (cap) {
      for (int w = 0; w < W; ++w) {
        const float* z = hz.data() + (size_t)w * no;
        if (t > 0) cnv[(size_t)w * T + (t - 1)] = z[no - 1];  // V(next_state_{t-1}) = V(state_t)  (masked by done_{t-1} in GAE)
        if (!extra) {
          const size_t row = (size_t)w * T + t;
          cv[row] = z[no - 1];
          memcpy(ch0 + row * A, z, sizeof(float) * A);
          if (two) {
            const float* z2 = hz2.data() + (size_t)w * no;
            cnv[row] = z2[no - 1];
            cv[row + 1] = z2[no - 1];
            memcpy(ch0 + (row + 1) * A, z2, sizeof(float) * A);
          }
        }
      }
    }
    if (extra) {
      const double dt = std::chrono::duration<double>(t1 - t0).count();
      c->t_act += dt;
      c->t_extra += dt;
      t = t_next;
      break;
    }
    if (t == 0) c->t_first += std::chrono::duration<double>(t1 - t0).count();
    if (!two) { tmp.env = L3.env; tmp.init(2 * W, S); }  // one-step exchange: L3 is free, the successors move up through it
    for (int w = 0; w < W; ++w) {
      const int k1 = 2 * w + (int)a0[w];
      size_t row = (size_t)w * T + t;  // worker-major (distributed_manager.py:30)
      memcpy(st + S * row, pub.data() + (size_t)S * w, sizeof(float) * S);
      memcpy(ns + S * row, &L1.next[(size_t)S * k1], sizeof(float) * S);
      ac_i[row] = a0[w];
      rw[row] = L1.rw[k1];
      dn[row] = L1.dn[k1];
      if (two) {
        const int k2 = 2 * k1 + (int)a1[w];
        row += 1;
        memcpy(st + S * row, &L1.obs[(size_t)S * k1], sizeof(float) * S);
        memcpy(ns + S * row, &L2.next[(size_t)S * k2], sizeof(float) * S);
        ac_i[row] = a1[w];
        rw[row] = L2.rw[k2];
        dn[row] = L2.dn[k2];
        vt.copy_row(e, w, L2.env, k2);
        if (t_next < steps) {  // (L1's rows of env w are read above before they are overwritten)
          copy_level_row(vt, L1, 2 * w, L3, 2 * k2);
          copy_level_row(vt, L1, 2 * w + 1, L3, 2 * k2 + 1);
        }
      } else {
        vt.copy_row(e, w, L1.env, k1);
        copy_level_row(vt, tmp, 2 * w, L2, 2 * k1);
        copy_level_row(vt, tmp, 2 * w + 1, L2, 2 * k1 + 1);
      }
    }
    if (!two)
      for (int i = 0; i < 2 * W; ++i) copy_level_row(vt, L1, i, tmp, i);
    c->steps += two ? 2 : 1;
    const auto t2 = std::chrono::steady_clock::now();
    if (dbg) d_book += secs(q3, t2);
    c->t_act += std::chrono::duration<double>(t1 - t0).count();
    c->t_env += std::chrono::duration<double>(t2 - t1).count();
    t = t_next;
    tag = tag_next;
    t0 = t0_next;
    pub.swap(pub_next);
  }
  *t_done = t;
  if (dbg && n_ex > 0 && (c->runs % 16) == 15)
    fprintf(stderr, "[jh_collect] host us per exchange (%d exchanges): fork %.2f  wait %.2f  sample %.2f  publish %.2f  book %.2f\n", n_ex, d_fork / n_ex * 1e6,
            d_wait / n_ex * 1e6, d_sample / n_ex * 1e6, d_publish / n_ex * 1e6, d_book / n_ex * 1e6);
  if (getenv("JH_PERSIST_DEBUG") && (c->runs % 16) == 15) jh_persist_dump_debug(c->persist, collector_steps(c, T));
  return JH_OK;
}

// Collect T steps from every env and append the W*T transitions (worker-major) to the store.
JH_EXPORT int jh_collector_run(jh_collector* c, int32_t T, int32_t training, jh_stream stream) {
  JH_ARG(c != nullptr && T > 0);
  if (run_state(c).active) return jh_fail(JH_ERR_STATE, "jh_collector_run inside a jh_collector_begin / jh_collector_loop pair");
  int rc = run_prepare(c, T, jh_s(stream));
  if (rc) return rc;
  rc = run_loop(c, training, jh_s(stream));
  const auto tc = std::chrono::steady_clock::now();
  rc = run_commit(c, run_state(c), rc, nullptr, 0, jh_s(stream));
  c->t_commit += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
  c->runs += 1;
  run_state(c).active = false;
  return rc;
}

// jh_collector_run in two halves with the commit launch enqueued AHEAD of the host loop:
//   jh_collector_begin  staging slabs, the acting kernel (unless prelaunched), then the commit launch -- gated: its workgroups wait
//                       (bounded) for a flag word in device-mapped pinned memory.  The store's row count advances here, so the caller
//                       may enqueue the consumer of the rollout (the learner's graph) right away, behind the commit, on `stream`.
//   jh_collector_loop   the T-step host loop; its last act is the release store of the flag.
// Needs the persistent acting kernel and a commit small enough for the one-launch form (<= 512 KB of rows).  A stalled environment
// (no observations for ~0.2 s) is an ERROR here -- work is already queued behind the acting kernel, so the per-step fallback of
// jh_collector_run does not exist; the flag is still released so that the stream drains.
JH_EXPORT int jh_collector_begin(jh_collector* c, int32_t T, jh_stream stream) {
  JH_ARG(c != nullptr && T > 0);
  if (run_state(c).active) return jh_fail(JH_ERR_STATE, "jh_collector_begin twice without jh_collector_loop");
  if (!c->persist) return jh_fail(JH_ERR_STATE, "jh_collector_begin needs the persistent acting kernel (W <= 32, W * S <= 512, JH_COLLECT_PERSISTENT != 0)");
  if (!c->gate_h) {
    JH_HIP(hipHostMalloc((void**)&c->gate_h, 64, hipHostMallocMapped));
    JH_HIP(hipHostGetDevicePointer((void**)&c->gate_d, c->gate_h, 0));
    c->gate_h[0] = 0;  // the release word
    c->gate_h[1] = 0;  // tag of a run whose commit launch timed out waiting for it (jh_store_copy_cols_kernel)
  }
  int rc = run_prepare(c, T, jh_s(stream));
  if (rc) return rc;
  RunState& r = run_state(c);
  if (!r.persistent || !jh_store_commit_is_one_launch(c->store, r.n)) {  // behave like jh_collector_run from here (commit at the end of the loop)
    r.early = false;
    return JH_OK;
  }
  r.early = true;
  c->gate_seq = (c->gate_seq + 1) & 0x7fffffffu;  // the top bit marks an aborted run
  if (c->gate_seq == 0) c->gate_seq = 1;
  const auto tc = std::chrono::steady_clock::now();
  rc = run_commit(c, r, JH_OK, c->gate_d, c->gate_seq, jh_s(stream));
  c->t_commit += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
  if (rc) {  // nothing is waiting on the flag that will not get it below
    __atomic_store_n(c->gate_h, c->gate_seq, __ATOMIC_RELEASE);
    r.active = false;
  }
  return rc;
}

JH_EXPORT int jh_collector_loop(jh_collector* c, int32_t training, jh_stream stream) {
  JH_ARG(c != nullptr);
  RunState& r = run_state(c);
  if (!r.active) return jh_fail(JH_ERR_STATE, "jh_collector_loop without jh_collector_begin");
  int rc = run_loop(c, training, jh_s(stream));
  if (r.early) {
    // every row, captured value and ride-along source is written: let the commit launch go.  After an ERROR the staging rows are half
    // filled: the abort value makes the launch copy nothing.  The learner's launches that the caller enqueued behind it still
    // run -- on whatever the store held before -- so the agent's weights are NOT trustworthy after this error; the caller must treat the
    // run as failed (the Python collectors raise).
    __atomic_store_n(c->gate_h, rc == JH_OK ? c->gate_seq : (c->gate_seq | 0x80000000u), __ATOMIC_RELEASE);
  } else {
    const auto tc = std::chrono::steady_clock::now();
    rc = run_commit(c, r, rc, nullptr, 0, jh_s(stream));
    c->t_commit += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
  }
  c->runs += 1;
  r.active = false;
  return rc;
}

// More of the same: out[0..5] = per RUN microseconds {first step's action wait (kernel start-up), value-only query, commit launch},
// steady-state action wait per timestep (first step and query excluded), runs, timesteps.
JH_EXPORT int jh_collector_stats_detail(jh_collector* c, double* out6) {
  JH_ARG(c && out6);
  const double r = c->runs > 0 ? (double)c->runs : 1.0;
  out6[0] = c->t_first / r * 1e6;
  out6[1] = c->t_extra / r * 1e6;
  out6[2] = c->t_commit / r * 1e6;
  const double inner = (double)c->steps - (double)c->runs;
  out6[3] = inner > 0 ? (c->t_act - c->t_first - c->t_extra) / inner * 1e6 : 0.0;
  out6[4] = (double)c->runs;
  out6[5] = (double)c->steps;
  return JH_OK;
}

// Diagnostics: host seconds per timestep spent (a) launching + waiting for the actions, (b) stepping
// the envs and writing the transitions.
JH_EXPORT int jh_collector_stats(jh_collector* c, double* act_us_per_step, double* env_us_per_step, int32_t reset) {
  JH_ARG(c != nullptr);
  const double n = c->steps > 0 ? (double)c->steps : 1.0;
  if (act_us_per_step) *act_us_per_step = c->t_act / n * 1e6;
  if (env_us_per_step) *env_us_per_step = c->t_env / n * 1e6;
  if (reset) { c->t_act = c->t_env = c->t_first = c->t_extra = c->t_commit = 0; c->steps = 0; c->runs = 0; }
  return JH_OK;
}
