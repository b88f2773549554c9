// Native sync-mode rollout collection for PPO on CartPole -- the whole of
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
int jh_persist_begin(jh_persist* p, int W, int T, int groups, hipStream_t st);
int jh_persist_step(jh_persist* p, int W, const float* h_obs, int64_t* h_action, int training);
unsigned jh_persist_next_tag(jh_persist* p);
void jh_persist_publish(jh_persist* p, int r0, int r1, const float* h_obs, unsigned tag);
int jh_persist_collect(jh_persist* p, int r0, int r1, unsigned tag, int64_t* h_action, int training);
void jh_persist_end_step(jh_persist* p);
void jh_persist_abort(jh_persist* p);
void jh_persist_dump_debug(jh_persist* p, int T);


struct jh_collector {
  jh_ctx* ctx = nullptr;
  jh_pponet* net = nullptr;
  jh_cartpole* env = nullptr;
  jh_store* store = nullptr;
  int W = 0;
  int col_state = 0, col_action = 1, col_reward = 2, col_next = 3, col_done = 4;
  std::vector<float> obs, next_obs, reward;
  std::vector<int64_t> act;
  std::vector<uint8_t> done;
  jh_persist* persist = nullptr;  // null: one launch per timestep
  int mode = 1;                   // 1: persistent acting kernel (default), 0: launch per step
  int groups = 1;                 // persistent mode: env rows served as this many half-batches per timestep (1 | 2).  Two halves in
                                  // flight (JH_PERSIST_GROUPS=2) measured SLOWER (11.2 vs 9.3 us per timestep): the kernel's work per
                                  // half is latency, not rows, so each half pays a full compute + poll-detection round
  double t_act = 0, t_env = 0, t_total = 0;  // host seconds: waiting for actions / stepping envs / whole runs
  int64_t steps = 0;
};

JH_EXPORT int jh_collector_create(jh_ctx* ctx, jh_pponet* net, jh_cartpole* env, jh_store* store,
                                  const int32_t* cols /* state, action, reward, next_state, done */,
                                  jh_collector** out) {
  JH_ARG(ctx && net && env && store && cols && out);
  JH_ARG(!net->cont && net->S == 4 && net->A == 2);
  JH_ARG(env->W <= net->max_act_rows);
  for (int i = 0; i < 5; ++i) JH_ARG(cols[i] >= 0 && cols[i] < store->n_cols);
  JH_ARG(store->cols[cols[0]].dtype == JH_F32 && store->cols[cols[0]].elems == 4);
  JH_ARG(store->cols[cols[1]].dtype == JH_I64 && store->cols[cols[1]].elems == 1);
  JH_ARG(store->cols[cols[2]].dtype == JH_F32 && store->cols[cols[2]].elems == 1);
  JH_ARG(store->cols[cols[3]].dtype == JH_F32 && store->cols[cols[3]].elems == 4);
  JH_ARG(store->cols[cols[4]].dtype == JH_U8 && store->cols[cols[4]].elems == 1);
  JH_HIP(hipSetDevice(ctx->device));
  jh_collector* c = new jh_collector();
  c->ctx = ctx; c->net = net; c->env = env; c->store = store; c->W = env->W;
  c->col_state = cols[0]; c->col_action = cols[1]; c->col_reward = cols[2]; c->col_next = cols[3]; c->col_done = cols[4];
  if (const char* e = getenv("JH_COLLECT_PERSISTENT")) c->mode = atoi(e);
  if (const char* e = getenv("JH_PERSIST_GROUPS")) c->groups = atoi(e) == 2 ? 2 : 1;
  if (c->W % 2) c->groups = 1;
  if (c->mode == 1 && c->W <= 16 && net->H % 128 == 0 && net->A + 1 <= 3) {
    if (jh_persist_create(net, &c->persist) != JH_OK) c->persist = nullptr;  // e.g. LDS too small: fall back
  }
  c->obs.resize(4 * (size_t)c->W);
  c->act.resize(c->W);
  c->next_obs.resize(4 * (size_t)c->W);
  c->reward.resize(c->W);
  c->done.resize(c->W);
  *out = c;
  return JH_OK;
}

JH_EXPORT void jh_collector_destroy(jh_collector* c) {
  delete c;
}

// Collect T steps from every env and append the W*T transitions (worker-major) to the store.
JH_EXPORT int jh_collector_run(jh_collector* c, int32_t T, int32_t training, jh_stream stream) {
  JH_ARG(c != nullptr && T > 0);
  const int W = c->W;
  const int64_t n = (int64_t)W * T;
  void* cols[16];
  int rc = jh_store_stage_begin(c->store, n, cols);
  if (rc) return rc;
  float* st = (float*)cols[c->col_state];
  int64_t* ac = (int64_t*)cols[c->col_action];
  float* rw = (float*)cols[c->col_reward];
  float* ns = (float*)cols[c->col_next];
  uint8_t* dn = (uint8_t*)cols[c->col_done];
  bool persistent = c->persist != nullptr;
  if (persistent) {
    rc = jh_persist_begin(c->persist, W, T, c->groups, jh_s(stream));
    if (rc) persistent = false;
  }
  auto record = [&](int t, int r0, int r1) {
    for (int w = r0; w < r1; ++w) {
      const size_t row = (size_t)w * T + t;  // worker-major: w0 t0..tT-1, w1 ...  (distributed_manager.py:30)
      memcpy(st + 4 * row, c->obs.data() + 4 * w, sizeof(float) * 4);
      memcpy(ns + 4 * row, c->next_obs.data() + 4 * w, sizeof(float) * 4);
      ac[row] = c->act[w];
      rw[row] = c->reward[w];
      dn[row] = c->done[w];
    }
  };
  int t_start = 0;
  if (persistent && c->groups == 2) {
    // Two half-batches in flight: while the host samples / steps / republishes half A, the kernel works on half B.
    // Per-env trajectories and the sampling stream (keyed by timestep and row) are the same as with one batch.
    const int half = W / 2;
    jh_cartpole_obs(c->env, c->obs.data());
    unsigned tag = jh_persist_next_tag(c->persist);
    jh_persist_publish(c->persist, 0, half, c->obs.data(), tag);
    jh_persist_publish(c->persist, half, W, c->obs.data(), tag);
    int t = 0;
    for (; t < T && persistent; ++t) {
      const unsigned tag_next = t + 1 < T ? jh_persist_next_tag(c->persist) : 0u;
      for (int g = 0; g < 2; ++g) {
        const int r0 = g * half, r1 = r0 + half;
        const auto t0 = std::chrono::steady_clock::now();
        rc = jh_persist_collect(c->persist, r0, r1, tag, c->act.data(), training);
        if (rc) {  // the kernel gave up: finish THIS timestep's remaining rows and the rest of the rollout with per-step launches
          jh_persist_abort(c->persist);
          JH_HIP(hipStreamSynchronize(jh_s(stream)));
          persistent = false;
          rc = jh_pponet_act_discrete(c->net, W - r0, c->obs.data() + 4 * r0, c->act.data() + r0, nullptr, nullptr, training, stream);
          if (rc) { (void)jh_store_stage_commit(c->store, stream); return rc; }
          jh_cartpole_step_rows(c->env, r0, W, c->act.data(), c->next_obs.data(), c->reward.data(), c->done.data());
          record(t, r0, W);
          break;
        }
        const auto t1 = std::chrono::steady_clock::now();
        jh_cartpole_step_rows(c->env, r0, r1, c->act.data(), c->next_obs.data(), c->reward.data(), c->done.data());
        record(t, r0, r1);
        if (t + 1 < T) {
          jh_cartpole_obs_rows(c->env, r0, r1, c->obs.data());  // next state, or the reset state where the episode ended
          jh_persist_publish(c->persist, r0, r1, c->obs.data(), tag_next);
        }
        const auto t2 = std::chrono::steady_clock::now();
        c->t_act += std::chrono::duration<double>(t1 - t0).count();
        c->t_env += std::chrono::duration<double>(t2 - t1).count();
      }
      jh_persist_end_step(c->persist);
      c->steps += 1;
      tag = tag_next;
    }
    t_start = t;  // == T, or (after a fall-back) the first timestep the generic loop below still has to do
  }
  for (int t = t_start; t < T; ++t) {
    jh_cartpole_obs(c->env, c->obs.data());  // current state of every env (reset state where it just finished)
    const auto t0 = std::chrono::steady_clock::now();
    if (persistent) {
      rc = jh_persist_step(c->persist, W, c->obs.data(), c->act.data(), training);
      if (rc) {  // the kernel gave up (it exits by itself): finish this rollout with per-step launches
        jh_persist_abort(c->persist);
        JH_HIP(hipStreamSynchronize(jh_s(stream)));
        persistent = false;
      }
    }
    if (!persistent) {
      rc = jh_pponet_act_discrete(c->net, W, c->obs.data(), c->act.data(), nullptr, nullptr, training, stream);
      if (rc) { (void)jh_store_stage_commit(c->store, stream); return rc; }
    }
    const auto t1 = std::chrono::steady_clock::now();
    rc = jh_cartpole_step(c->env, c->act.data(), c->next_obs.data(), c->reward.data(), c->done.data());
    if (rc) { (void)jh_store_stage_commit(c->store, stream); return rc; }
    record(t, 0, W);
    const auto t2 = std::chrono::steady_clock::now();
    c->t_act += std::chrono::duration<double>(t1 - t0).count();
    c->t_env += std::chrono::duration<double>(t2 - t1).count();
    c->steps += 1;
  }
  if (persistent && getenv("JH_PERSIST_DEBUG") && (c->steps % (64 * T)) == 0) jh_persist_dump_debug(c->persist, T);
  return jh_store_stage_commit(c->store, stream);
}

// Diagnostics: host seconds per timestep spent (a) launching + waiting for the actions, (b) stepping
// the envs and writing the transitions.
JH_EXPORT int jh_collector_stats(jh_collector* c, double* act_us_per_step, double* env_us_per_step, int32_t reset) {
  JH_ARG(c != nullptr);
  const double n = c->steps > 0 ? (double)c->steps : 1.0;
  if (act_us_per_step) *act_us_per_step = c->t_act / n * 1e6;
  if (env_us_per_step) *env_us_per_step = c->t_env / n * 1e6;
  if (reset) { c->t_act = c->t_env = 0; c->steps = 0; }
  return JH_OK;
}
