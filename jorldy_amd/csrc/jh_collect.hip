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
int jh_persist_begin(jh_persist* p, int W, int T, hipStream_t st);
unsigned jh_persist_publish(jh_persist* p, int W, const float* h_obs);
int jh_persist_collect(jh_persist* p, int W, unsigned tag, float* h_heads);
int jh_persist_heads(const jh_persist* p);
void jh_persist_abort(jh_persist* p);
void jh_persist_dump_debug(jh_persist* p, int T);

struct jh_collector {
  jh_ctx* ctx = nullptr;
  jh_pponet* net = nullptr;
  jh_cartpole* cart = nullptr;  // exactly one of the two envs is set
  jh_control* ctl = nullptr;
  jh_store* store = nullptr;
  int W = 0, S = 0, A = 0;
  bool cont = false;
  int col_state = 0, col_action = 1, col_reward = 2, col_next = 3, col_done = 4;
  std::vector<float> obs, next_obs, reward, heads, act_f;
  std::vector<int64_t> act_i;
  std::vector<uint8_t> done;
  jh_persist* persist = nullptr;  // null: one launch per timestep
  int mode = 1;                   // 1: persistent acting kernel (default), 0: launch per step (JH_COLLECT_PERSISTENT=0)
  double t_act = 0, t_env = 0, t_total = 0;  // host seconds: waiting for actions / stepping envs / whole runs
  int64_t steps = 0;
};

static int collector_create(jh_ctx* ctx, jh_pponet* net, jh_cartpole* cart, jh_control* ctl, jh_store* store, const int32_t* cols,
                            jh_collector** out) {
  JH_ARG(ctx && net && (cart || ctl) && store && cols && out);
  const int W = cart ? cart->W : ctl->W, S = cart ? 4 : ctl->S, A = cart ? 2 : ctl->A;
  const bool cont = ctl != nullptr;
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
  c->ctx = ctx; c->net = net; c->cart = cart; c->ctl = ctl; c->store = store; c->W = W; c->S = S; c->A = A; c->cont = cont;
  c->col_state = cols[0]; c->col_action = cols[1]; c->col_reward = cols[2]; c->col_next = cols[3]; c->col_done = cols[4];
  if (const char* e = getenv("JH_COLLECT_PERSISTENT")) c->mode = atoi(e);
  if (c->mode == 1 && W <= 16 && W * S <= 128) {
    if (jh_persist_create(net, &c->persist) != JH_OK) c->persist = nullptr;  // unsupported width: one launch per step
  }
  c->obs.resize((size_t)S * W);
  c->next_obs.resize((size_t)S * W);
  c->act_i.resize(W);
  c->act_f.resize((size_t)A * W);
  c->reward.resize(W);
  c->done.resize(W);
  c->heads.resize(16 * (size_t)W);
  *out = c;
  return JH_OK;
}

JH_EXPORT int jh_collector_create(jh_ctx* ctx, jh_pponet* net, jh_cartpole* env, jh_store* store,
                                  const int32_t* cols /* state, action, reward, next_state, done */,
                                  jh_collector** out) {
  return collector_create(ctx, net, env, nullptr, store, cols, out);
}

// The same collector for a continuous-action policy on the synthetic control env (config.ppo.mujoco shapes):
// store columns state f32[S], action f32[A], reward f32[1], next_state f32[S], done u8[1].
JH_EXPORT int jh_collector_create_control(jh_ctx* ctx, jh_pponet* net, jh_control* env, jh_store* store, const int32_t* cols,
                                          jh_collector** out) {
  return collector_create(ctx, net, nullptr, env, store, cols, out);
}

JH_EXPORT void jh_collector_destroy(jh_collector* c) {
  if (!c) return;
  if (c->persist) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipDeviceSynchronize();
    jh_persist_destroy(c->persist);
  }
  delete c;
}

// Collect T steps from every env and append the W*T transitions (worker-major) to the store.
JH_EXPORT int jh_collector_run(jh_collector* c, int32_t T, int32_t training, jh_stream stream) {
  JH_ARG(c != nullptr && T > 0);
  const int W = c->W, S = c->S, A = c->A;
  const int64_t n = (int64_t)W * T;
  void* cols[16];
  int rc = jh_store_stage_begin(c->store, n, cols);
  if (rc) return rc;
  float* st = (float*)cols[c->col_state];
  int64_t* ac_i = (int64_t*)cols[c->col_action];
  float* ac_f = (float*)cols[c->col_action];
  float* rw = (float*)cols[c->col_reward];
  float* ns = (float*)cols[c->col_next];
  uint8_t* dn = (uint8_t*)cols[c->col_done];
  bool persistent = c->persist != nullptr;
  if (persistent) {
    rc = jh_persist_begin(c->persist, W, T, jh_s(stream));
    if (rc) persistent = false;
  }
  for (int t = 0; t < T; ++t) {
    // current state of every env (reset state where it just finished)
    if (c->cart) jh_cartpole_obs(c->cart, c->obs.data());
    else jh_control_obs(c->ctl, c->obs.data());
    const auto t0 = std::chrono::steady_clock::now();
    if (persistent) {
      const unsigned tag = jh_persist_publish(c->persist, W, c->obs.data());
      rc = jh_persist_collect(c->persist, W, tag, c->heads.data());
      if (rc) {  // the kernel gave up (it exits by itself): finish this rollout with per-step launches
        jh_persist_abort(c->persist);
        JH_HIP(hipStreamSynchronize(jh_s(stream)));
        persistent = false;
      } else {
        const int no = jh_persist_heads(c->persist);
        for (int w = 0; w < W; ++w) {
          if (c->cont) jh_sample_continuous(c->net, c->heads.data() + (size_t)w * no, w, training, c->act_f.data() + (size_t)w * A);
          else c->act_i[w] = jh_sample_discrete(c->net, c->heads.data() + (size_t)w * no, w, training);
        }
        c->net->act_ctr += 1;
      }
    }
    if (!persistent) {
      rc = c->cont ? jh_pponet_act_continuous(c->net, W, c->obs.data(), c->act_f.data(), nullptr, nullptr, training, stream)
                   : jh_pponet_act_discrete(c->net, W, c->obs.data(), c->act_i.data(), nullptr, nullptr, training, stream);
      if (rc) { (void)jh_store_stage_commit(c->store, stream); return rc; }
    }
    const auto t1 = std::chrono::steady_clock::now();
    rc = c->cart ? jh_cartpole_step(c->cart, c->act_i.data(), c->next_obs.data(), c->reward.data(), c->done.data())
                 : jh_control_step(c->ctl, c->act_f.data(), c->next_obs.data(), c->reward.data(), c->done.data());
    if (rc) { (void)jh_store_stage_commit(c->store, stream); return rc; }
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
