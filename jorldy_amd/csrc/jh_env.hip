// Vectorised host-side CartPole-v1 (synthetic stand-in for gym, which cannot be installed in the
// build image).  W environments advance in one call: float64 dynamics (standard cart-pole ODE,
// Euler tau = 0.02), float32 observations, reward shaping of core/env/gym_env.py:78
// (-1 on done, else 0.1) and the auto-reset of Actor.run (manager/distributed_manager.py:91).
// Reset noise comes from one splitmix64 stream per env so the CPU oracle
// (oracle/jorldy_oracle.py: CartPoleOracle) reproduces every trajectory bit for bit.
#include <math.h>

#include "jh_common.h"


namespace {
constexpr double kGrav = 9.8, kMc = 1.0, kMp = 0.1, kLen = 0.5, kFmag = 10.0, kTau = 0.02;
constexpr double kThetaLim = 12.0 * 2.0 * 3.14159265358979323846 / 360.0;
constexpr double kXLim = 2.4;
constexpr int64_t kMaxSteps = 500;

inline double next_u01(uint64_t& st) {
  st += 0x9E3779B97F4A7C15ull;
  uint64_t x = st;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x = x ^ (x >> 31);
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

inline void reset_env(jh_cartpole* e, int w) {
  for (int k = 0; k < 4; ++k) e->s[4 * (size_t)w + k] = -0.05 + 0.1 * next_u01(e->rng[w]);
  e->t[w] = 0;
}
}  // namespace

JH_EXPORT int jh_cartpole_create(int32_t W, uint64_t seed, jh_cartpole** out) {
  JH_ARG(out != nullptr && W > 0);
  jh_cartpole* e = new jh_cartpole();
  e->W = W;
  e->s.assign(4 * (size_t)W, 0.0);
  e->t.assign(W, 0);
  e->rng.resize(W);
  for (int w = 0; w < W; ++w) {
    e->rng[w] = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(w + 1) * 0xBF58476D1CE4E5B9ull;
    reset_env(e, w);
  }
  *out = e;
  return JH_OK;
}

JH_EXPORT void jh_cartpole_destroy(jh_cartpole* e) { delete e; }

JH_EXPORT int jh_cartpole_obs(const jh_cartpole* e, float* h_obs) {
  JH_ARG(e && h_obs);
  for (size_t i = 0; i < 4 * (size_t)e->W; ++i) h_obs[i] = (float)e->s[i];
  return JH_OK;
}

// envs [r0, r1) only (the arrays are indexed by env, i.e. the caller passes the full-width buffers)
void jh_cartpole_obs_rows(const jh_cartpole* e, int r0, int r1, float* h_obs) {
  for (size_t i = 4 * (size_t)r0; i < 4 * (size_t)r1; ++i) h_obs[i] = (float)e->s[i];
}

JH_EXPORT int jh_cartpole_step(jh_cartpole* e, const int64_t* h_action, float* h_next_obs, float* h_reward,
                               uint8_t* h_done) {
  JH_ARG(e && h_action && h_next_obs && h_reward && h_done);
  jh_cartpole_step_rows(e, 0, e->W, h_action, h_next_obs, h_reward, h_done);
  return JH_OK;
}

void jh_cartpole_step_rows(jh_cartpole* e, int r0, int r1, const int64_t* h_action, float* h_next_obs, float* h_reward, uint8_t* h_done) {
  const double total_mass = kMc + kMp, pml = kMp * kLen;
  for (int w = r0; w < r1; ++w) {
    double* s = &e->s[4 * (size_t)w];
    double x = s[0], xd = s[1], th = s[2], thd = s[3];
    const double force = h_action[w] == 1 ? kFmag : -kFmag;
    const double ct = cos(th), st = sin(th);
    const double temp = (force + pml * thd * thd * st) / total_mass;
    const double thacc = (kGrav * st - ct * temp) / (kLen * (4.0 / 3.0 - kMp * ct * ct / total_mass));
    const double xacc = temp - pml * thacc * ct / total_mass;
    x = x + kTau * xd;
    xd = xd + kTau * xacc;
    th = th + kTau * thd;
    thd = thd + kTau * thacc;
    s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
    e->t[w] += 1;
    const bool d = x < -kXLim || x > kXLim || th < -kThetaLim || th > kThetaLim || e->t[w] >= kMaxSteps;
    for (int k = 0; k < 4; ++k) h_next_obs[4 * (size_t)w + k] = (float)s[k];
    h_done[w] = d ? 1 : 0;
    h_reward[w] = d ? -1.0f : 0.1f;  // gym_env.py:78
    if (d) reset_env(e, w);          // distributed_manager.py:91
  }
}

// ------------------------------------------------------------------------------ synthetic continuous control
// MuJoCo is not installable in the build image; config.ppo.mujoco (Hopper-v3: S = 11, A = 3, continuous actions in
// [-1, 1]) gets a deterministic stand-in of the same shapes so that the native collector has an environment to
// drive: float64 dynamics
//     s'_i = 0.95 s_i + 0.05 sum_j P[i][j] a_j + 0.02 sin(s_{(i+1) mod S}),   P[i][j] = 0.5 sin(1.7 (i+1) + 2.3 (j+1))
//     reward = s'_0 + 0.1 - 0.001 |a|^2,   done when |s'_0| > 2 or after 1000 steps,   reset U(-0.05, 0.05)^S
// float32 observations, auto-reset like Actor.run (manager/distributed_manager.py:91).  The CPU oracle
// (oracle/jorldy_oracle.py: ControlOracle) reproduces every trajectory bit for bit.
namespace {
inline void reset_control(jh_control* e, int w) {
  for (int k = 0; k < e->S; ++k) e->s[(size_t)e->S * w + k] = -0.05 + 0.1 * next_u01(e->rng[w]);
  e->t[w] = 0;
}
}  // namespace

JH_EXPORT int jh_control_create(int32_t W, int32_t S, int32_t A, uint64_t seed, jh_control** out) {
  JH_ARG(out != nullptr && W > 0 && S > 0 && S <= 64 && A > 0 && A <= 16);
  jh_control* e = new jh_control();
  e->W = W; e->S = S; e->A = A;
  e->s.assign((size_t)S * W, 0.0);
  e->t.assign(W, 0);
  e->rng.resize(W);
  for (int w = 0; w < W; ++w) {
    e->rng[w] = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(w + 1) * 0xBF58476D1CE4E5B9ull;
    reset_control(e, w);
  }
  *out = e;
  return JH_OK;
}

JH_EXPORT void jh_control_destroy(jh_control* e) { delete e; }

JH_EXPORT int jh_control_obs(const jh_control* e, float* h_obs) {
  JH_ARG(e && h_obs);
  jh_control_obs_rows(e, 0, e->W, h_obs);
  return JH_OK;
}
// rows r0 .. r1-1 of the full arrays (absolute row indexing, like jh_cartpole_*_rows): the collector steps the two halves of a 32-row env apart
void jh_control_obs_rows(const jh_control* e, int r0, int r1, float* h_obs) {
  for (size_t i = (size_t)e->S * r0; i < (size_t)e->S * r1; ++i) h_obs[i] = (float)e->s[i];
}

JH_EXPORT int jh_control_step(jh_control* e, const float* h_action, float* h_next_obs, float* h_reward, uint8_t* h_done) {
  JH_ARG(e && h_action && h_next_obs && h_reward && h_done);
  jh_control_step_rows(e, 0, e->W, h_action, h_next_obs, h_reward, h_done);
  return JH_OK;
}

void jh_control_step_rows(jh_control* e, int r0, int r1, const float* h_action, float* h_next_obs, float* h_reward, uint8_t* h_done) {
  const int S = e->S, A = e->A;
  double nxt[64];
  for (int w = r0; w < r1; ++w) {
    double* s = &e->s[(size_t)S * w];
    const float* a = h_action + (size_t)A * w;
    double a2 = 0.0;
    for (int j = 0; j < A; ++j) a2 += (double)a[j] * (double)a[j];
    for (int i = 0; i < S; ++i) {
      double drive = 0.0;
      for (int j = 0; j < A; ++j) drive += 0.5 * sin(1.7 * (i + 1) + 2.3 * (j + 1)) * (double)a[j];
      nxt[i] = 0.95 * s[i] + 0.05 * drive + 0.02 * sin(s[(i + 1) % S]);
    }
    for (int i = 0; i < S; ++i) s[i] = nxt[i];
    e->t[w] += 1;
    const bool d = s[0] > 2.0 || s[0] < -2.0 || e->t[w] >= 1000;
    for (int i = 0; i < S; ++i) h_next_obs[(size_t)S * w + i] = (float)s[i];
    h_done[w] = d ? 1 : 0;
    h_reward[w] = (float)(s[0] + 0.1 - 0.001 * a2);
    if (d) reset_control(e, w);
  }
}
