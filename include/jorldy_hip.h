/*
 * jorldy_hip.h -- C ABI of libjorldy_hip.so, the MI355X (gfx950) native RL hot path
 * that sits behind JORLDY's core/agent + core/buffer API.
 *
 * The reference (kakaoenterprise/JORLDY, all paths below relative to
 * /root/reference/jorldy/) has NO native code and NO FFI: every function here
 * replaces a piece of pure-Python/numpy/torch-CPU code.  Each entry point cites
 * the reference file:line it stands in for.  INTEGRATION.md shows the ctypes
 * binding a JORLDY maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative jh_status otherwise;
 *     jh_last_error() returns a thread-local, human readable message.
 *   - `d_*` parameters are DEVICE pointers (e.g. torch `tensor.data_ptr()`),
 *     borrowed for the duration of the enqueued work; `h_*` are HOST pointers,
 *     consumed before the call returns (copied into the library's pinned staging).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     work is enqueued on it; no call synchronises unless its comment says so.
 *   - one jh_ctx per GPU; a ctx (and objects created from it) is not thread-safe.
 *   - plain C types only; no torch / C++ types cross this boundary.
 */
#ifndef JORLDY_HIP_H
#define JORLDY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JH_ABI_VERSION 2

typedef enum {
  JH_OK = 0,
  JH_ERR_HIP = -1,      /* a HIP runtime call failed (message has hipGetErrorString) */
  JH_ERR_ARG = -2,      /* invalid argument */
  JH_ERR_STATE = -3,    /* object in the wrong state (e.g. sampling an empty tree) */
  JH_ERR_NOMEM = -4,
  JH_ERR_NODEVICE = -5  /* no usable gfx950 device */
} jh_status;

typedef enum { JH_U8 = 0, JH_F32 = 1, JH_I64 = 2, JH_F64 = 3, JH_I32 = 4 } jh_dtype;

typedef struct jh_ctx jh_ctx;
typedef struct jh_store jh_store;
typedef struct jh_per jh_per;
typedef struct jh_cartpole jh_cartpole;
typedef void* jh_stream;

/* ------------------------------------------------------------------ library / context */
int jh_abi_version(void);
/* End a stream capture that was invalidated and abandoned by its owner (hipStreamEndCapture + destroy whatever graph comes back) and clear
 * the sticky error: 1 = there was a capture to end, 0 = the stream was not capturing.  The host side calls it when torch.cuda.graph raises
 * out of capture_end(), which leaves the capture stream current and still capturing.                                                  */
int jh_stream_abort_capture(jh_stream stream);
const char* jh_last_error(void);
int jh_device_count(void);                       /* 0 when no GPU is visible (never an error) */
int jh_ctx_create(int device, jh_ctx** out);     /* binds the device, allocates pinned staging */
void jh_ctx_destroy(jh_ctx* ctx);
/* Host cores local to this context's GPU (sysfs local_cpulist of its PCI function, e.g. "64-127,192-255"): the
 * collector thread (Actor.run's replacement, manager/distributed_manager.py:76-92) should run there -- acting crosses
 * PCIe twice per timestep and the far socket adds ~1.8 us per crossing.                                   */
int jh_ctx_local_cpulist(jh_ctx* ctx, char* out, int64_t len);
int jh_ctx_sync(jh_ctx* ctx, jh_stream stream);  /* hipStreamSynchronize */
/* Pinned host memory mapped into the device address space (the pinned staging of the collector:
 * observations written by the host are read in place by the acting kernels, actions come back
 * the same way).  *dev_out is the address kernels must use.                                 */
/* Host-side wait for results that kernels write into device-mapped pinned memory (the learn() statistics: the
 * reference reads them with .item() syncs, core/agent/ppo.py:171-184, rainbow.py:241-253): returns 0 once none of
 * base[idx[i]], i < n, equals `sentinel` any more, 1 after timeout_s seconds.  Pure host spin, no HIP call.          */
int jh_host_wait_marks(const float* base, const int32_t* idx, int32_t n, float sentinel, double timeout_s);
/* The same on 32-bit words (e.g. the low words of int64 actions preset to -1 by the host). */
int jh_host_wait_words(const uint32_t* base, const int32_t* idx, int32_t n, uint32_t sentinel, double timeout_s);
/* The epoch shuffles of PPO.learn (core/agent/ppo.py:116-118: idxs = np.arange(M); np.random.shuffle(idxs) per epoch, cumulative)
 * drawn IN PLACE from numpy's global MT19937 with numpy's algorithm (RandomState._shuffle_raw + random_interval), through the
 * state address / next_uint32 / next_uint64 function pointers of BitGenerator.ctypes: h_perm_out [epochs][n] = the index list of
 * every epoch, bit-identical to the Python calls (3 us instead of 12 us per 1024 indices).  Pure host code.                    */
int jh_np_legacy_shuffles(void* bitgen_state, void* next_uint32_fn, void* next_uint64_fn, int64_t n, int32_t epochs,
                          int64_t* h_perm_out);
int jh_pinned_alloc(jh_ctx* ctx, int64_t bytes, void** host_out, void** dev_out);
void jh_pinned_free(void* host);

/* Per-kernel timing with HIP events recorded on the launch stream around every kernel of this
 * library (measurement only; off by default).  jh_prof_report synchronises the device and writes
 * "<kernel>\t<launches>\t<total_ms>\n" lines into buf.                                          */
int jh_prof_enable(int32_t on);
int jh_prof_report(char* buf, int64_t cap);
int jh_prof_calibrate(int32_t n, jh_stream stream); /* n empty event pairs -> "__event_pair_overhead" */
/* Measurement aid: a streaming read of exactly `bytes` device bytes with `width` (4 / 8 / 16) bytes per lane and access -- the
 * known byte count the rocprofv3 HBM counters are calibrated against per access width (tools/pmc_calibrate.sh).              */
int jh_calib_stream(jh_ctx* ctx, const void* d_src, int64_t bytes, int32_t width, float* d_out, jh_stream stream);

/* ------------------------------------------------------------------ transition store
 * GPU-resident struct-of-arrays ring that replaces the list-of-dicts storage of
 *   ReplayBuffer   core/buffer/replay_buffer.py:8-35   (ring + uniform gather)
 *   RolloutBuffer  core/buffer/rollout_buffer.py:6-24  (append, take all, clear)
 * and BaseBuffer.stack_transition core/buffer/base.py:42-56 (AoS->SoA, done once at
 * push time on the host side instead of at every sample).
 * Column c is a device array [capacity][elems[c]] of dtype[c].                      */
typedef struct {
  int32_t dtype;  /* jh_dtype of the stored column */
  int64_t elems;  /* elements per transition */
} jh_col_desc;

int jh_store_create(jh_ctx* ctx, int64_t capacity, int32_t n_cols, const jh_col_desc* cols, jh_store** out);
void jh_store_destroy(jh_store* s);
/* Append n transitions (ring write at buffer_index, wraps; replay_buffer.py:16-23).
 * h_cols[c] points at n*elems[c] contiguous elements of dtype[c].  The data is copied into
 * pinned staging before returning, then moved with hipMemcpyAsync on `stream`.          */
int jh_store_push(jh_store* s, int64_t n, const void* const* h_cols, jh_stream stream);
/* Zero-copy variant for native collectors: get pinned slab pointers for n rows, fill them,
 * then commit (enqueues the async H2D copies).  begin/commit must alternate.             */
int jh_store_stage_begin(jh_store* s, int64_t n, void** h_cols_out);
int jh_store_stage_commit(jh_store* s, jh_stream stream);
/* Same ring append from DEVICE-resident rows already in the stored dtypes (on-device collectors,
 * growing a rollout store): d_cols[c] -> n*elems[c] elements, device-to-device async copies.   */
int jh_store_push_device(jh_store* s, int64_t n, const void* const* d_cols, jh_stream stream);
/* out[c][b][:] = convert(col[c][idx[b] - idx_offset][:]) for the selected columns
 * (replay_buffer.py:25-31 + base.py:42-56 + the fp32 cast of BaseAgent.as_tensor,
 * core/agent/base.py:61-73).  out_dtype[c] is JH_F32 (as_tensor semantics) or the stored
 * dtype (keeps uint8 frames 4x smaller).  d_idx: int64[B]; idx_offset lets PER pass
 * tree-space indices (leaf = idx - (N-1), per_buffer.py:95).                             */
/* Positional writes: row i of the host columns lands in slot h_slots[i] (0 <= slot < capacity); index / counter
 * of the ring are not touched.  Used for the frame pool of the de-duplicated image replay (SURVEY.md §8f rank 2:
 * single 84x84 frames + per-transition frame indices instead of two 4-frame stacks per transition).       */
int jh_store_write_rows(jh_store* s, int64_t n, const int64_t* h_slots, const void* const* h_cols, jh_stream stream);
int jh_store_gather(jh_store* s, int64_t B, const int64_t* d_idx, int64_t idx_offset, int32_t n_sel,
                    const int32_t* sel_cols, void* const* d_out, const int32_t* out_dtype, jh_stream stream);
void* jh_store_col_ptr(jh_store* s, int32_t col);  /* device base of a column (rollout "sample" is a view) */
int64_t jh_store_size(const jh_store* s);          /* buffer_counter */
int64_t jh_store_index(const jh_store* s);         /* buffer_index   */
int64_t jh_store_capacity(const jh_store* s);
void jh_store_clear(jh_store* s);                  /* rollout_buffer.py:19 */
int jh_store_set_position(jh_store* s, int64_t index, int64_t counter); /* checkpoint restore */

/* ------------------------------------------------------------------ prioritized replay
 * Device-resident float64 sum tree, array-heap layout identical to
 * core/buffer/per_buffer.py:7-105: 2N-1 nodes, leaves at [N-1, 2N-2].  All updates
 * are the reference's incremental `+= delta` climbs applied in batch order per node,
 * so the tree stays BIT-IDENTICAL to the reference's numpy tree.                        */
int jh_per_create(jh_ctx* ctx, int64_t capacity, double uniform_sample_prob, jh_per** out);
void jh_per_destroy(jh_per* p);
/* per_buffer.py:19-40: n x add_tree_data at tree_index (wraps independently).
 * h_prio == NULL -> every new leaf gets the current max_priority (per_buffer.py:27-31). */
int jh_per_push(jh_per* p, int64_t n, const double* h_prio, jh_stream stream);
/* The same append with priorities that already live in HBM (float64 [n], device): jh_feed_tick's actor-side priorities. */
int jh_per_push_device(jh_per* p, int64_t n, const double* d_prio, jh_stream stream);
/* per_buffer.py:42-54 applied for b = 0..B-1 in order: d_idx int64[B] TREE-space indices,
 * d_prio float32[B] (prio_dtype JH_F32: the `.item()` of an fp32 tensor, per.py:68-70,
 * rainbow.py:230-231) or float64[B].  Duplicated indices behave sequentially.          */
int jh_per_update(jh_per* p, int64_t B, const int64_t* d_idx, const void* d_prio, int32_t prio_dtype, jh_stream stream);
/* per_buffer.py:70-101.  The three numpy global-RNG draws stay on the host so indices are
 * bit-exact: n_uniform = sum(uniform(B) < usp); h_uniform_slot = randint(counter, n_uniform);
 * h_u = uniform(B - n_uniform) (NOT yet multiplied by the root).  Outputs (device):
 * d_idx int64[B] tree-space, uniform first; d_w64 float64[B] and/or d_w32 float32[B]
 * (either may be NULL) = ((1/counter)/P)^beta / max; d_stats float64[4] =
 * {sampled_p, mean_p, root, max_w_unnormalised}.                                        */
int jh_per_sample(jh_per* p, int64_t B, double beta, int64_t n_uniform, const int64_t* h_uniform_slot,
                  const double* h_u, int64_t* d_idx, double* d_w64, float* d_w32, double* d_stats, jh_stream stream);
/* Blocking read-back of the scalar state (synchronises `stream`). */
/* Sharded PER for data-parallel learners: every rank samples from its own shard with jh_per_sample;
 * jh_per_shard_stats writes {root, count, min priority of that sample} (3 float64) for an all-gather over the ranks,
 * jh_per_weights_sharded recomputes the IS weights of the last sample against the LOGICAL buffer (sum of roots /
 * counts, maximum weight over the global batch = weight of the smallest gathered priority): per_buffer.py:88-94 for
 * one tree holding all shards.  d_all3 float64[n_shards][3] in rank order.                                          */
int jh_per_shard_stats(jh_per* p, int64_t B, double* d_out3, jh_stream stream);
int jh_per_weights_sharded(jh_per* p, int64_t B, double beta, const double* d_all3, int32_t n_shards, double* d_w64,
                           float* d_w32, jh_stream stream);
int jh_per_state(jh_per* p, double* max_priority, double* root, int64_t* tree_index, int64_t* counter, jh_stream stream);
double* jh_per_tree_ptr(jh_per* p);  /* device float64[2N-1] */
int64_t jh_per_tree_size(const jh_per* p);
/* Checkpoint / restore of the whole tree state (blocking; the reference cannot resume its buffer,
 * SURVEY.md §5 -- this is what a complete checkpoint needs).  h_tree: float64[2N-1].   */
int jh_per_load(jh_per* p, const double* h_tree, double max_priority, int64_t tree_index, int64_t counter);
int jh_per_dump(jh_per* p, double* h_tree, jh_stream stream);

/* ------------------------------------------------------------------ PPO math
 * jh_gae: core/agent/ppo.py:95-110.  Inputs float32[W*T], worker-major (row w = one
 * worker's T steps).  delta = r + (1-d)*gamma*V' - V; reverse scan per row that does not
 * bootstrap across the row end; ret = adv + V; if standardize: per-row
 * (adv-mean)/(std_unbiased+1e-7).  One wave per row, wave-shuffle segmented scan.       */
int jh_gae(jh_ctx* ctx, int32_t W, int32_t T, float gamma, float lambda, const float* d_reward, const float* d_done,
           const float* d_value, const float* d_next_value, float* d_adv, float* d_ret, int32_t standardize,
           jh_stream stream);
/* ppo.py:118-125 gathers state[idx], action[idx], adv[idx], ret[idx], value[idx], log_prob_old[idx] inside the
 * minibatch loop; the index lists of all epochs exist before the loop, so the rows are gathered ONCE per learn():
 * row i of d_dst[c] ([n][elems[c]] float32) = row d_idx[i] of d_src[c].  n_cols <= 8.  The minibatch kernels then
 * read consecutive rows (jh_pponet_ppo_update with d_idx = NULL).                                                   */
int jh_ppo_minibatch_rows(jh_ctx* ctx, int64_t n, const int64_t* d_idx, int32_t n_cols, const int32_t* elems,
                          const float* const* d_src, float* const* d_dst, jh_stream stream);
/* ppo.py:112 `ret.mean()`: *d_out = mean of n floats, one workgroup, fixed summation order.                          */
int jh_mean_f32(jh_ctx* ctx, int64_t n, const float* d_x, float* d_out, jh_stream stream);
/* log pi_old(a|s): ppo.py:90-92 `pi.gather(1, action).log()` with pi = exp(log_softmax(logits)). */
int jh_logp_discrete(jh_ctx* ctx, int64_t M, int32_t A, const float* d_logits, const float* d_action, float* d_logp,
                     jh_stream stream);
/* ppo.py:85-88 continuous: per-dimension Normal log-prob of atanh(clamp(a)); takes RAW heads
 * (mu before clamp(+-5), log_std before tanh; core/network/policy_value.py:52-56).      */
int jh_logp_continuous(jh_ctx* ctx, int64_t M, int32_t A, const float* d_mu_raw, const float* d_log_std_raw,
                       const float* d_action, float* d_logp, jh_stream stream);
/* Clipped surrogate + clipped value + entropy, forward AND backward to the head outputs
 * (ppo.py:131-165).  Row i of the minibatch reads the rollout-sized arrays at
 * r = d_idx ? d_idx[i] : i  (the `x[idx]` gathers of ppo.py:122-125 are fused away);
 * d_logits / d_value_pred are minibatch-sized [B][A] / [B].
 * d_stats float32[8] = {loss, actor_loss, critic_loss, entropy_loss, max_ratio, min_prob, c1, c2}. */
int jh_ppo_loss_discrete(jh_ctx* ctx, int32_t B, int32_t A, const float* d_logits, const float* d_value_pred,
                         const int64_t* d_idx, const float* d_action, const float* d_adv, const float* d_ret,
                         const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef,
                         float ent_coef, float* d_grad_logits, float* d_grad_value, float* d_stats,
                         jh_stream stream);
int jh_ppo_loss_continuous(jh_ctx* ctx, int32_t B, int32_t A, const float* d_mu_raw, const float* d_log_std_raw,
                           const float* d_value_pred, const int64_t* d_idx, const float* d_action, const float* d_adv,
                           const float* d_ret, const float* d_value_old, const float* d_logp_old, float eps_clip,
                           float vf_coef, float ent_coef, float* d_grad_mu_raw, float* d_grad_log_std_raw,
                           float* d_grad_value, float* d_stats, jh_stream stream);
/* The same loss for DATA-PARALLEL learners with the critic of ppo.py:147-154 exact over the global minibatch (see
 * jh_pponet_ppo_update_dp_begin): both critic branches' value gradients are kept (d_grad_value | d_dv2 float32[B]), d_critic_sums
 * float32[2] <- this rank's {sum e1, sum e2}, d_stats_local float32[8] <- this rank's statistics.  After the caller's all-reduce (MEAN
 * over the ranks) of d_critic_sums, jh_ppo_critic_select_rows mixes d_grad_value in place with the global branch weights and writes
 * d_stats (actor / entropy terms of this rank, critic terms of the global minibatch).                                            */
int jh_ppo_loss_deferred(jh_ctx* ctx, int32_t continuous, int32_t B, int32_t A, const float* d_head0, const float* d_head1,
                         const float* d_value_pred, const int64_t* d_idx, const float* d_action, const float* d_adv, const float* d_ret,
                         const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef, float ent_coef,
                         float* d_grad_head0, float* d_grad_head1, float* d_grad_value, float* d_dv2, float* d_critic_sums,
                         float* d_stats_local, jh_stream stream);
int jh_ppo_critic_select_rows(jh_ctx* ctx, int32_t B, const float* d_critic_sums, float vf_coef, float ent_coef, float* d_grad_value,
                              const float* d_dv2, const float* d_stats_local, float* d_stats, jh_stream stream);

/* The same losses for a network whose LAST LAYER stacks the heads (the policy-value net on the CNN head as jh_rbnet runs it: policy_value.py:8-22 with
 * head.py:21-61 under it; rows of the last layer = pi [A] | v, or mu [A] | log_std [A] | v): d_heads float32[B][ld] = (head0 [A] | head1 [A] (continuous) |
 * value | padding), gradient back in the same layout (d_grad_heads float32[B][ld], padding columns untouched).  d_dv2 float32[B] + d_critic_sums float32[2]
 * both non-NULL: the deferred critic of jh_ppo_loss_deferred (d_stats is then the LOCAL row; finish with jh_ppo_critic_select_strided on the value column,
 * ldv = ld).  Replaces ppo.py:122-165 for config.ppo.atari / ppo.procgen.                                                                               */
int jh_ppo_loss_packed(jh_ctx* ctx, int32_t continuous, int32_t B, int32_t A, const float* d_heads, int32_t ld, const int64_t* d_idx, const float* d_action,
                       const float* d_adv, const float* d_ret, const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef,
                       float ent_coef, float* d_grad_heads, float* d_dv2, float* d_critic_sums, float* d_stats, jh_stream stream);
int jh_ppo_critic_select_strided(jh_ctx* ctx, int32_t B, const float* d_critic_sums, float vf_coef, float ent_coef, float* d_grad_value, int32_t ldv,
                                 const float* d_dv2, const float* d_stats_local, float* d_stats, jh_stream stream);
/* d_packed float32[rows][ld] -> d_h0 [rows][A], d_h1 [rows][A] (NULL: one head), d_value [rows]: what jh_logp_* / jh_gae read (ppo.py:83-94, once per learn()) */
int jh_heads_unpack(jh_ctx* ctx, int64_t rows, int32_t A, const float* d_packed, int32_t ld, float* d_h0, float* d_h1, float* d_value, jh_stream stream);
/* PPO.act, discrete policy, heads already on the device (ppo.py:63-69): action[w] = Categorical(softmax(d_heads[w][0..A))).sample() by inverse CDF on the
 * counter-based stream (seed, counter = timestep, w) -- the stream of jh_pponet_act_discrete's host-side sampler --, or the first maximum when !training.
 * d_action int64[W]: device or device-mapped host memory.                                                                                                */
int jh_policy_act_discrete(jh_ctx* ctx, int32_t W, int32_t A, const float* d_heads, int32_t ld, uint64_t seed, uint64_t counter, int32_t training,
                           int64_t* d_action, jh_stream stream);

/* ------------------------------------------------------------------ TD losses (DQN family)
 * One kernel for dqn.py:128-141, double.py:28-39, multistep.py:41-50, per.py:54-74,
 * ape_x.py:96-116.  flags: */
#define JH_TD_DOUBLE 1   /* a* = argmax Q_online(s'); bootstrap Q_target(s')[a*]           */
#define JH_TD_PER 2      /* loss = mean(w*td^2), priorities td^alpha; else Huber (beta=1)  */
/* d_q [B][A] online Q(s); d_q_next_online [B][A] (DOUBLE only); d_q_next_target [B][A];
 * d_action float32[B]; d_reward/d_done float32 [B][n] (n = max(n_step,1); n_step==0 is the
 * 1-step form); d_weights float32[B] (PER only).  Outputs: d_grad_q [B][A] (d loss/d Q(s)),
 * d_prio float32[B] (= td^alpha, or |td| when not PER; may be NULL),
 * d_stats float32[4] = {loss, max_Q, mean_td, 0}.                                        */
int jh_td_loss(jh_ctx* ctx, int32_t B, int32_t A, int32_t n_step, int32_t flags, const float* d_q,
               const float* d_q_next_online, const float* d_q_next_target, const float* d_action,
               const float* d_reward, const float* d_done, const float* d_weights, float gamma, float alpha,
               float* d_grad_q, float* d_prio, float* d_stats, jh_stream stream);

/* ------------------------------------------------------------------ C51 / Rainbow
 * Categorical n-step projection + cross-entropy, forward and backward to the online
 * logits (rainbow.py:167-239, c51.py:68-109, logits2Q rainbow.py:285-292).  flags: */
#define JH_C51_DOUBLE 1     /* rainbow: action from argmax Q_online(s'); else target net's own */
#define JH_C51_PER 2        /* rainbow: prio = KL^alpha, loss = mean(w)*mean(KL) (shape-broadcast quirk) */
#define JH_C51_SHIFT_MAX 4  /* c51.py:126-128 subtracts the row max before log_softmax        */
/* d_logit / d_next_logit_online / d_target_logit: [B][A][K]; d_reward/d_done [B][n];
 * outputs d_grad_logit [B][A][K], d_prio float32[B] (NULL ok), d_kl float32[B] (NULL ok),
 * d_stats float32[8] = {loss, max_Q, max_logit, min_logit, mean_kl, 0,0,0}.              */
int jh_c51_loss(jh_ctx* ctx, int32_t B, int32_t A, int32_t K, int32_t n_step, int32_t flags, const float* d_logit,
                const float* d_next_logit_online, const float* d_target_logit, const float* d_action,
                const float* d_reward, const float* d_done, const float* d_weights, float v_min, float v_max,
                float gamma, float alpha, float* d_grad_logit, float* d_prio, float* d_kl, float* d_stats,
                jh_stream stream);

/* Batched acting of the value-net agents: DQN.act / ApeX.act / C51.act / Rainbow.act (core/agent/dqn.py:76-92,
 * ape_x.py:64-77, c51.py:50-66, rainbow.py:140-152) for N actors in one call.  d_logits [N][A][K] are the network's
 * outputs (K = 1: Q values; K > 1: atom logits, Q = expectation under softmax over the support linspace(v_min,
 * v_max, K), rainbow.py:285-292).  Epsilon-greedy per actor with the HOST's draws (h_eps float32[N], h_u float64[N]
 * = np.random.random(), h_rand_action int64[N] = np.random.randint(A); all three NULL: greedy).  Outputs (device):
 * d_action int64[N] (first maximum, like torch.argmax), d_q_taken float32[N] (Q of the action taken; NULL ok),
 * d_q_all float32[N][A] (NULL ok).                                                                                */
int jh_value_act(jh_ctx* ctx, int32_t N, int32_t A, int32_t K, const float* d_logits, float v_min, float v_max,
                 const float* h_eps, const double* h_u, const int64_t* h_rand_action, int64_t* d_action,
                 float* d_q_taken, float* d_q_all, jh_stream stream);

/* ------------------------------------------------------------------ native policy-value MLP
 * The encoder of the PPO configs (core/network/head.py:6-18 MLP head + policy_value.py:8-57):
 * S -> H relu -> H relu -> {A logits | A mu, A log_std} + value, as hand-written kernels
 * (fp32-input MFMA for the H x H contractions).  Parameters, gradients and Adam moments are FLAT
 * fp32 device buckets borrowed from the caller, laid out in the reference's state_dict order:
 *   head.l.weight [H][S], head.l.bias [H], l.weight [H][H], l.bias [H],
 *   pi.weight [A][H], pi.bias [A]                              (discrete)
 *   mu.weight, mu.bias, log_std.weight [A][H], log_std.bias    (continuous)
 *   v.weight [1][H], v.bias [1]
 * so one RCCL all-reduce covers the whole gradient and torch views give state_dict()/ckpt compat. */
typedef struct jh_pponet jh_pponet;
int64_t jh_pponet_param_count(int32_t S, int32_t H, int32_t A, int32_t continuous);
int jh_pponet_create(jh_ctx* ctx, int32_t S, int32_t H, int32_t A, int32_t continuous, int32_t max_rows,
                     float* d_params, float* d_grads, float* d_m, float* d_v, uint64_t seed, jh_pponet** out);
void jh_pponet_destroy(jh_pponet* n);
/* Adam hyper-parameters live in device memory (a captured graph can be replayed while the host
 * anneals lr, core/agent/base.py:93-111).  step >= 0 sets Adam's step counter (checkpoint restore),
 * step < 0 keeps it.  Hyper-parameters are DOUBLES: torch.optim derives (1 - beta) and the bias
 * corrections 1 - beta^t in Python double arithmetic before its fp32 kernels see them, and so does this
 * library ((1.f - 0.999f) would be off by 1.3e-5 relative, visible in exp_avg_sq).                */
int jh_pponet_set_hyper(jh_pponet* n, double lr, double beta1, double beta2, double eps, double step, jh_stream stream);
int jh_pponet_set_lr(jh_pponet* n, double lr, jh_stream stream);
/* Forward of B rows of d_x [*, S] (gathered through d_idx int64[B] when non-NULL: the `state[idx]`
 * of ppo.py:122).  Writes the RAW heads: d_head0 = logits | mu_raw [B][A], d_head1 = log_std_raw
 * (continuous only), d_value [B].  Activations stay in the net for a following backward.     */
int jh_pponet_forward(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, float* d_head0,
                      float* d_head1, float* d_value, jh_stream stream);
/* Backward of the last forward given d(loss)/d(raw heads); OVERWRITES the flat gradient bucket
 * (== optimizer.zero_grad + loss.backward, ppo.py:164-165).                                    */
int jh_pponet_backward(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_g_head0,
                       const float* d_g_head1, const float* d_g_value, jh_stream stream);
/* torch.nn.utils.clip_grad_norm_(max_norm) (skipped if max_norm <= 0) + torch.optim.Adam.step
 * on the flat buckets (ppo.py:166-169).  d_norm_out: optional device float, pre-clip norm.      */
int jh_pponet_adam_step(jh_pponet* n, float max_norm, float* d_norm_out, jh_stream stream);
/* One whole PPO minibatch update (ppo.py:122-169: forward of `state[idx]`, clipped loss forward +
 * backward, encoder backward, clip_grad_norm_, Adam) in 4-5 launches (csrc/jh_ppo_mb.hip): layer 1 is
 * generated in the operand fetch, the heads never exist as tensors (per-column-tile partials are summed
 * by the loss kernel), d(loss)/d(h2) is generated in the operand fetch of its two consumers, and ONE
 * backward grid produces every gradient (dh1 only as the per-row-tile partial sums of dW1 / db1).
 * B <= 1024, hidden_size % 32 == 0.  d_idx may be NULL (rows already gathered: jh_ppo_minibatch_rows).
 * do_adam == 0 stops after the backward with a complete gradient bucket (data-parallel: all-reduce, then
 * jh_pponet_adam_step).  d_stats float32[8] as in jh_ppo_loss_*.                                    */
int jh_pponet_ppo_update(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_action,
                         const float* d_adv, const float* d_ret, const float* d_value_old, const float* d_logp_old,
                         float eps_clip, float vf_coef, float ent_coef, float max_norm, int32_t do_adam, float* d_stats,
                         jh_stream stream);
/* The minibatch update of ppo.py:122-169 for ANY number of rows up to max_rows (config.ppo.mujoco: 2048-row minibatches) in one call: forward, the loss --
 * forward and backward in ONE launch whatever B (both critic branches' value gradients are kept per row, the last workgroup to arrive reduces the
 * partials and leaves the branch weights; the backward's first kernel forms the value gradient) --, backward, [clip_grad_norm_ + Adam unless
 * do_adam == 0].  Bit-identical to jh_pponet_forward -> jh_ppo_loss_discrete / _continuous -> jh_pponet_backward -> jh_pponet_adam_step, one launch less. */
int jh_pponet_ppo_update_rows(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_action, const float* d_adv, const float* d_ret,
                              const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef, float ent_coef, float max_norm,
                              int32_t do_adam, float* d_stats, jh_stream stream);
/* jh_pponet_ppo_update for DATA-PARALLEL learners with the reference's critic exactly: core/agent/ppo.py:147-154 takes
 * max(mean(e1), mean(e2)) over the WHOLE minibatch -- a max of two means, so with the minibatch sharded over ranks the branch is
 * only known after {sum e1, sum e2} have been reduced.  _begin: forward + loss of this rank's B rows; d_critic_sums float32[2] <- this
 * rank's sums.  The caller all-reduces d_critic_sums (MEAN over ranks; every rank holds B rows).  _end: branch weights from the
 * reduced sums, value gradients, backward -> a complete gradient bucket (then: all-reduce MEAN of the bucket, jh_pponet_adam_step).
 * d_stats float32[8] as in jh_ppo_loss_*: actor / entropy terms of this rank's rows, critic terms of the global minibatch.       */
int jh_pponet_ppo_update_dp_begin(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_action, const float* d_adv,
                                  const float* d_ret, const float* d_value_old, const float* d_logp_old, float eps_clip, float vf_coef,
                                  float ent_coef, float* d_critic_sums, jh_stream stream);
int jh_pponet_ppo_update_dp_end(jh_pponet* n, int32_t B, const float* d_x, const int64_t* d_idx, const float* d_critic_sums, float vf_coef,
                                float ent_coef, float* d_stats, jh_stream stream);
/* jh_pponet_ppo_update_dp_end with the ranks' exchange of the critic sums folded into its first launch (jh_peer_*, B <= 256 rows per rank):
 * d_critic_sums = this rank's sums on entry, the ranks' mean on exit.  Declared behind jh_peer below (forward declaration here).  */
struct jh_peer;
int jh_pponet_ppo_update_dp_end_peer(jh_pponet* n, struct jh_peer* peer, int32_t B, const float* d_x, const int64_t* d_idx, float* d_critic_sums,
                                     float vf_coef, float ent_coef, float* d_stats, jh_stream stream);
/* Device address of the optimizer's hyper block (float[8]: lr, beta1, beta2, eps, step, ...), e.g. as the destination of a
 * jh_collector_set_ride_along copy that delivers the next decayed learning rate (base.py:93-111) without a copy of its own. */
void* jh_pponet_hyper_ptr(jh_pponet* n);
/* Sampling stream of the acting calls and the collectors (counter-based: the action of env row w at acting step c is a function
 * of (seed, c, w); the reference samples with torch.multinomial / torch.normal, ppo.py:55-69): read, or with set != 0 restore,
 * {seed, counter} -- what a resumed run needs to continue the same action stream.                                     */
int jh_pponet_act_rng(jh_pponet* n, uint64_t* seed, uint64_t* counter, int32_t set);
/* PPO.act for W envs in one shot (ppo.py:55-69, discrete): ONE kernel launch computes the fused MLP
 * forward and writes per-column-tile partial head outputs + sequence words into device-mapped
 * pinned memory; the host polls them, sums the partials, and does softmax + multinomial (argmax
 * when training == 0).  h_obs [W][S] / h_action [W] / h_logits_out [W][A] / h_value_out [W] are
 * ordinary HOST pointers; BLOCKING: returns when the actions are available.                     */
int jh_pponet_act_discrete(jh_pponet* n, int32_t W, const float* h_obs, int64_t* h_action, float* h_logits_out,
                           float* h_value_out, int32_t training, jh_stream stream);
/* The continuous policy (ppo.py:55-63): h_action [W][A] = tanh(Normal(clamp(mu_raw, -5, 5), exp(tanh(log_std_raw)))
 * .sample()), tanh(mu) when training == 0; h_mu_raw_out / h_log_std_raw_out [W][A] and h_value_out [W] optional.   */
int jh_pponet_act_continuous(jh_pponet* n, int32_t W, const float* h_obs, float* h_action, float* h_mu_raw_out,
                             float* h_log_std_raw_out, float* h_value_out, int32_t training, jh_stream stream);

/* ------------------------------------------------------------------ vectorised host collector
 * Synthetic CartPole-v1 (gym is not installable in the build image): W envs stepped in one
 * call on the host, float64 dynamics, reward shaping of core/env/gym_env.py:78, auto-reset
 * like Actor.run (manager/distributed_manager.py:76-92).  Pure host code.               */
int jh_cartpole_create(int32_t W, uint64_t seed, jh_cartpole** out);
void jh_cartpole_destroy(jh_cartpole* e);
int jh_cartpole_obs(const jh_cartpole* e, float* h_obs /* [W][4] */);
int jh_cartpole_step(jh_cartpole* e, const int64_t* h_action /* [W] */, float* h_next_obs /* [W][4] */,
                     float* h_reward /* [W] */, uint8_t* h_done /* [W] */);

/* Synthetic continuous-control env at config.ppo.mujoco shapes (MuJoCo is not installable in the build image; Hopper-v3
 * is S = 11, A = 3, actions in [-1, 1]): deterministic float64 dynamics (csrc/jh_env.hip), float32 observations,
 * auto-reset like Actor.run (manager/distributed_manager.py:91); mirrored bit for bit by oracle ControlOracle.  */
typedef struct jh_control jh_control;
int jh_control_create(int32_t W, int32_t S, int32_t A, uint64_t seed, jh_control** out);
void jh_control_destroy(jh_control* e);
int jh_control_obs(const jh_control* e, float* h_obs /* [W][S] */);
int jh_control_step(jh_control* e, const float* h_action /* [W][A] */, float* h_next_obs /* [W][S] */,
                    float* h_reward /* [W] */, uint8_t* h_done /* [W] */);

/* ------------------------------------------------------------------ native sync collector
 * DistributedManager.run + Actor.run (manager/distributed_manager.py:26-31,76-92) for every worker
 * and T steps in one call: batched on-GPU acting through device-mapped pinned memory, host env
 * stepping, transitions written worker-major into the rollout store's pinned staging and moved with
 * one hipMemcpyAsync per column.  cols = store column indices of {state, action, reward,
 * next_state, done} (f32[4], i64[1], f32[1], f32[4], u8[1]).                                       */
typedef struct jh_collector jh_collector;

/* The host envs behind the collector as a table of functions (round 5): what `Actor.run` needs from an env
 * (manager/distributed_manager.py:76-92: the current observations, one step with auto-reset) plus, OPTIONALLY, the ability to be
 * copied on the host.  The two built-in envs above are two such tables (jh_collector_create / _create_control build them);
 * jh_collector_create_env plugs any other host simulator -- a C user's, or Python callbacks through ctypes (tests).
 *   obs   rows r0 .. r1-1 of the CURRENT observations (the reset observation where the last step ended an episode) -> h_obs [r1-r0][S]
 *   step  rows r0 .. r1-1 take h_action (int64 [n] for a discrete policy, float [n][A] in [-1, 1] for a continuous one):
 *         h_next_obs [n][S] (the terminal observation where done), h_reward [n], h_done [n]; finished rows reset themselves
 *   both return 0 or a negative JH_ERR_* (the run stops, what was collected is still committed).
 *   fork_alloc / fork_free / copy_row (all three or none): a scratch env of `rows` rows, its release, and "row di of dst becomes an
 *   exact copy of row si of src" including the row's RNG stream.  An env that offers them and has two discrete actions gets TWO
 *   timesteps per acting exchange (the collector steps copies one to three steps ahead while the GPU evaluates the policy; rollouts
 *   are bit-identical to one timestep per exchange); one that does not runs one timestep per exchange.
 * The collector calls obs / step with r0 = 0, r1 = W on the env itself and with arbitrary row ranges on scratch envs only.          */
typedef struct jh_env_vtbl {
  int32_t W, S, A, continuous;
  int (*obs)(void* env, int32_t r0, int32_t r1, float* h_obs);
  int (*step)(void* env, int32_t r0, int32_t r1, const void* h_action, float* h_next_obs, float* h_reward, uint8_t* h_done);
  void* (*fork_alloc)(void* env, int32_t rows);
  void (*fork_free)(void* scratch);
  void (*copy_row)(void* dst, int32_t di, const void* src, int32_t si);
} jh_env_vtbl;
int jh_collector_create_env(jh_ctx* ctx, jh_pponet* net, const jh_env_vtbl* vt, void* env, jh_store* store, const int32_t* cols,
                            jh_collector** out);

int jh_collector_create(jh_ctx* ctx, jh_pponet* net, jh_cartpole* env, jh_store* store, const int32_t* cols,
                        jh_collector** out);
/* The same for a CONTINUOUS policy (PPO.act, ppo.py:55-63: tanh(Normal(mu, std).sample())) on jh_control: the
 * persistent acting kernel returns mu / log_std partials, the host samples; action column f32[A], state f32[S].   */
int jh_collector_create_control(jh_ctx* ctx, jh_pponet* net, jh_control* env, jh_store* store, const int32_t* cols,
                                jh_collector** out);
void jh_collector_destroy(jh_collector* c);
int jh_collector_run(jh_collector* c, int32_t T, int32_t training, jh_stream stream);
/* Acting-time capture: PPO.learn opens with two no-grad passes of the network over the rollout (core/agent/ppo.py:83-94:
 * pi, value = network(state); next_value = network(next_state)[-1]); in sync mode the actors' weights ARE the learner's, so
 * these numbers existed when the actions were sampled.  After jh_collector_set_capture every jh_collector_run of rows = W * T
 * transitions also delivers, in the launch that commits the rows (worker-major like the store): d_h0 [rows][A] raw policy head
 * (logits | mu_raw), d_h1 [rows][A] log_std_raw (continuous; NULL otherwise), d_value [rows] = V(state_t), d_next_value [rows]
 * = V(state_{t+1}) (one extra value-only query after the last step; where done_t is set the entry belongs to the reset state
 * and is multiplied by (1 - done_t) = 0 in GAE, ppo.py:96).  d_value == NULL switches capture off.                         */
int jh_collector_set_capture(jh_collector* c, float* d_h0, float* d_h1, float* d_value, float* d_next_value, int64_t rows);
/* jh_collector_run in two halves, the commit launch enqueued AHEAD of the host loop:
 *   jh_collector_begin  staging, the acting kernel (unless prelaunched), then the commit launch (rows + captured block + ride-along
 *                       copies) -- gated: its workgroups wait, bounded, for a flag word in device-mapped pinned memory.  The store's
 *                       row count advances here: the caller may enqueue the rollout's consumer (the learner's launches) on `stream` now.
 *   jh_collector_loop   the T-step host loop (Actor.run, manager/distributed_manager.py:76-92); its last act releases the flag.
 * The consumer then starts the instant the rollout ends instead of after the host has returned and launched it.  Needs the persistent
 * acting kernel (W <= 32, W * S <= 512) and a one-launch commit (<= 512 KB of rows), else the pair behaves exactly like jh_collector_run.
 * A stalled environment (no observations for ~0.2 s) is an ERROR in this form (work is queued behind the acting kernel, the per-step
 * fallback of jh_collector_run cannot run); the flag is released regardless so that the stream drains.                              */
int jh_collector_begin(jh_collector* c, int32_t T, jh_stream stream);
int jh_collector_loop(jh_collector* c, int32_t training, jh_stream stream);
/* Two more copies (slot 0 / 1; bytes == 0 clears the slot) for the commit launch of every following run: device-visible source
 * (device-mapped pinned memory, jh_pinned_alloc) -> device buffer, read when that launch executes (the end of the run).  For the
 * learner's inputs that change between learn() calls and are known before the rollout ends: the coming epochs' minibatch index
 * lists (ppo.py:116-118, drawn ahead) and the decayed learning rate (base.py:93-111).                                    */
int jh_collector_set_ride_along(jh_collector* c, int32_t slot, const void* d_src_mapped, void* d_dst, int64_t bytes);
/* Enqueue the persistent acting kernel of the NEXT jh_collector_run(T) now, e.g. right behind the learner's last launch: it
 * starts when the stream reaches it, reads the then-current weights and waits (bounded, ~0.2 s) for the first observations.
 * Nothing else may be enqueued on `stream` before that run.  A kernel that timed out is replaced by the run itself.      */
int jh_collector_prelaunch(jh_collector* c, int32_t T, jh_stream stream);
/* Host-side timing of the collection loop (microseconds per timestep): launching + waiting for the
 * actions, and stepping the envs + writing the transitions.                                        */
int jh_collector_stats(jh_collector* c, double* act_us_per_step, double* env_us_per_step, int32_t reset);
/* out6 = {first step's action wait per run (acting kernel start-up), value-only query per run, commit launch per run,
 * steady-state action wait per timestep (both excluded), runs, timesteps}, microseconds; call before a resetting jh_collector_stats. */
int jh_collector_stats_detail(jh_collector* c, double* out6);

/* N(0,1) draws on the device for NoisyNet layers (core/network/utils.py:58-60 draws torch.randn per forward):
 * counter-based (element i of call c = Box-Muller on splitmix64(seed, c, i)); d_state uint64[4] = {seed, call counter,
 * 0, 0} in DEVICE memory, advanced by the kernel itself, so a replayed hipGraph draws fresh noise every time.        */
int jh_normal_fill(jh_ctx* ctx, int64_t n, float* d_out, uint64_t* d_state, jh_stream stream);

/* ------------------------------------------------------------------ native value networks
 * The encoders of the DQN / Rainbow / Ape-X family with their learn()-side network work:
 *   kind 0  rainbow  core/network/rainbow.py:8-94: head -> l -> noisy a1|v1 -> noisy a2, v2 -> dueling over K atoms
 *                    (utils.py:55-107 factorised noisy linear)
 *   kind 3  rainbow with independent Gaussian noise (utils.py:72-79): one draw per weight instead of the outer product
 *   kind 1  dueling  core/network/dueling.py:8-35: head -> l1_a|l1_v -> l2_a, l2_v -> dueling combine  (K = 1)
 *   kind 2  q        core/network/q_network.py:8-20: head -> l -> q                                    (K = 1)
 * on core/network/head.py:6-61 (MLP or Nature-CNN head), plus the three forwards / backward / optimizer step
 * of the agents' learn() (rainbow.py:160-186,236-238; dqn.py:128-147; ape_x.py:96-131).  Convolutions are
 * implicit GEMMs on the fp32 MFMA; uint8 NCHW frames are read straight from the replay store (divide by 255
 * inside the operand fetch, head.py:46).  Parameters live in flat fp32 buckets with a private layout;
 * jh_rbnet_segment describes it (rows x cols, row-major, 16-byte aligned offsets, rows = 0: absent):
 *   0 conv1.weight [32][(c,ky,kx)] | head.l.weight [H][S] (mlp)     1 its bias
 *   2 conv2.weight [64][(ky,kx,c)]   3 bias    4 conv3.weight [64][(ky,kx,c)]   5 bias   (cnn only)
 *   6 l.weight [H][F] (cnn: F ordered (y,x,c))   7 l.bias                         (kinds 0, 2)
 *   8 first stream layer, a|v stacked [2H][in] (in = H, or F for kind 1; rainbow: mu_w^T)   9 sig_w   10 bias [2H]   11 sig_b
 *   12-15 a2 / l2_a / q: weight [A*K][H], sig_w, bias, sig_b        16-19 v2 / l2_v: weight [K][H], sig_w, bias, sig_b
 * (sig_* only for kind 0).                                                                              */
typedef struct jh_rbnet jh_rbnet;
int64_t jh_rbnet_param_count_for(int32_t kind, int32_t head_cnn, int32_t channels_or_state_size, int32_t height, int32_t width,
                                 int32_t hidden, int32_t action_size, int32_t num_support);
/* d_params (online), d_target, d_grads, d_m, d_v: caller-owned flat fp32 device buckets of
 * jh_rbnet_param_count_for(...) floats each (16-byte aligned), borrowed for the net's lifetime.
 * d_m / d_v: Adam exp_avg / exp_avg_sq, or RMSprop grad_avg / square_avg.                               */
int jh_rbnet_create(jh_ctx* ctx, int32_t kind, int32_t head_cnn, int32_t channels_or_state_size, int32_t height, int32_t width,
                    int32_t hidden, int32_t action_size, int32_t num_support, int32_t max_batch, float* d_params,
                    float* d_target, float* d_grads, float* d_m, float* d_v, jh_rbnet** out);
void jh_rbnet_destroy(jh_rbnet* n);
int64_t jh_rbnet_param_count(const jh_rbnet* n);
int32_t jh_rbnet_segment_count(void);
int jh_rbnet_segment(const jh_rbnet* n, int32_t i, int64_t* offset, int32_t* rows, int32_t* cols);
/* Length of one noise set (kind 0): N(0,1) draws in the reference's draw order (utils.py:58-60, layers a1, v1,
 * a2, v2): [e_in a1 H][e_out a1 H][e_in v1 H][e_out v1 H][e_in a2 H][e_out a2 A*K][e_in v2 H][e_out v2 K];
 * kind 3: per layer [eps_w (in x out, row-major like the reference's mu_w)][eps_b (out)]                   */
int64_t jh_rbnet_noise_len(const jh_rbnet* n);
/* Adam: (lr, beta1, beta2, eps).  RMSprop: (lr, alpha, unused, eps) + centered.  step = optimizer step counter. */
int jh_rbnet_set_hyper(jh_rbnet* n, double lr, double beta1_or_alpha, double beta2, double eps, int64_t step, int32_t centered,
                       jh_stream stream);
int jh_rbnet_set_lr(jh_rbnet* n, double lr, jh_stream stream);
/* update_target (dqn.py:162-163, rainbow.py:270-271): target <- online                                  */
int jh_rbnet_sync_target(jh_rbnet* n, jh_stream stream);
/* network(x[, is_train]): rows <= max_batch observations (JH_U8 or JH_F32; NCHW images or [rows][S]),
 * which 0 online / 1 target, d_noise one noise set or NULL (kind 0: is_train = False) -> logits [rows][A][K] */
int jh_rbnet_forward(jh_rbnet* n, int32_t which, const void* d_x, int32_t x_dtype, int32_t rows, const float* d_noise,
                     float* d_logits, jh_stream stream);
/* An on-policy learner's forward (ppo.py:127-135 on the CNN head): network(x) of the online parameters for B <= max_batch rows, activations kept for
 * jh_rbnet_backward (which then takes d(loss)/d(outputs) [B][A][K] of these rows).  Kinds 1 and 2 (no noise).                                  */
int jh_rbnet_forward_keep(jh_rbnet* n, const void* d_x, int32_t x_dtype, int32_t B, float* d_logits, jh_stream stream);
/* The three forwards of learn(): d_x = [state; next_state] (2B rows), d_noise three noise sets (kind 0, else
 * NULL) -> d_logits [3][B][A][K] = online(state), online(next_state), target(next_state)                */
int jh_rbnet_learn_forward(jh_rbnet* n, const void* d_x, int32_t x_dtype, int32_t B, const float* d_noise, float* d_logits,
                           jh_stream stream);
/* jh_rbnet_learn_forward in two halves + the part that does not depend on the batch (rainbow.py:160-186: the three forwards of learn()):
 *   jh_rbnet_prepare_noise  W = mu + sig * eps of the three noisy weight sets (network/utils.py:55-86) for the draw d_noise [3][noise_len]
 *                           -- may run on another stream while the trunk runs (a 12-us launch off the critical path)
 *   jh_rbnet_learn_trunk    head + l of [state; next_state] (online) and next_state (target)
 *   jh_rbnet_learn_heads    the noisy dueling heads of the three forwards -> d_logits [3][B][A][K]; uses the prepared sets when
 *                           jh_rbnet_prepare_noise ran for the same d_noise, else materialises them itself                          */
int jh_rbnet_prepare_noise(jh_rbnet* n, const float* d_noise, jh_stream stream);
int jh_rbnet_learn_trunk(jh_rbnet* n, const void* d_x, int32_t x_dtype, int32_t B, jh_stream stream);
int jh_rbnet_learn_heads(jh_rbnet* n, int32_t B, const float* d_noise, float* d_logits, jh_stream stream);
/* loss.backward() given d(loss)/d(online(state) output) [B][A][K] (from jh_c51_loss / jh_td_loss; NULL after jh_rbnet_c51_step); fills d_grads */
int jh_rbnet_backward(jh_rbnet* n, const float* d_g, jh_stream stream);
/* Rainbow.learn()'s loss step on the network's own stream outputs (rainbow.py:160-235), three launches where the separate calls
 * (jh_rbnet_learn_heads' combine, jh_c51_loss x 2, jh_per_update x 2, jh_rbnet_backward's first kernel) take six:
 *   jh_rbnet_learn_heads_raw  = jh_rbnet_learn_heads up to the advantage / value streams (network/rainbow.py:76-87); no logits yet
 *   jh_rbnet_c51_step         launch 1: dueling combine of the three forwards (rainbow.py:88-93 -> d_logits [3][B][A][K], bit-identical
 *                             to jh_rbnet_learn_heads) + double-Q action + n-step projection + KL + priorities KL^alpha (-> d_prio, d_kl)
 *                             + the gradient pulled back through the combine (kept in the network);
 *                             launch 2: batch statistics (d_stats as jh_c51_loss) + priorities into the leaves d_tree_idx of `per`
 *                             (per_buffer.py:42-54); launch 3: the climb.  per NULL: no write-back.  flags: JH_C51_DOUBLE required.
 *   jh_rbnet_backward(n, NULL, ..) continues from the gradient jh_rbnet_c51_step left.                                               */
int jh_rbnet_learn_heads_raw(jh_rbnet* n, int32_t B, const float* d_noise, jh_stream stream);
int jh_rbnet_c51_step(jh_rbnet* n, jh_per* per, int32_t B, int32_t n_step, int32_t flags, const float* d_action, const float* d_reward,
                      const float* d_done, const float* d_weights, const int64_t* d_tree_idx, float v_min, float v_max, float gamma,
                      float alpha, float* d_logits, float* d_prio, float* d_kl, float* d_stats, jh_stream stream);
/* jh_rbnet_backward that leaves two elementwise tails undone -- d(sigma) = d(mu) * eps of the noisy layers (network/utils.py:60-70) and
 * the sum of conv1's weight-gradient partials -- for jh_rbnet_optim_step to do inside the optimizer's own pass (no clipping) or as the
 * launches they were (clipping).  jh_rbnet_flush_grads completes the gradient bucket for anybody who reads it before the optimizer step
 * (a data-parallel all-reduce).                                                                                                       */
int jh_rbnet_backward_deferred(jh_rbnet* n, const float* d_g, jh_stream stream);
int jh_rbnet_flush_grads(jh_rbnet* n, jh_stream stream);
/* [clip_grad_norm_(max_norm) when max_norm > 0 (ape_x.py:128),] optimizer.step(): optimizer 0 torch.optim.Adam,
 * 1 torch.optim.RMSprop (momentum 0, centered per set_hyper); advances the step counter                 */
int jh_rbnet_optim_step(jh_rbnet* n, int32_t optimizer, float max_norm, jh_stream stream);
int jh_rbnet_adam_step(jh_rbnet* n, jh_stream stream);

/* The dense form of the grouped LDS-tiled fp32 MFMA GEMM every value-network layer runs on (nn.Linear forward /
 * data gradient / weight gradient of the core/network modules): C[M][N] = sum_k A(m,k) B(k,n) with a fused epilogue.
 *   a_kcont != 0: A stored [M][K] (lda), else [K][M];   b_kcont != 0: B stored [N][K] (ldb), else [K][N]
 *   epi 0 none | 1 + bias[n] | 2 relu(. + bias[n]) | 3 zero where aux[m][n] <= 0;  d_rowsum (optional) [M] = sum_k A(m,k) */
int jh_tgemm_dense(jh_ctx* ctx, int32_t M, int32_t N, int32_t K, const float* d_a, int32_t lda, int32_t a_kcont, const float* d_b,
                   int32_t ldb, int32_t b_kcont, float* d_c, int32_t ldc, int32_t epi, const float* d_bias, const float* d_aux,
                   int32_t ldaux, float* d_rowsum, jh_stream stream);
/* Test entry: n (<= 6) independent problems C_j [M][N] = A_j [M][K] B_j [N][K]^T as ONE grouped launch (the shape of the value
 * networks' forward launches: online and target trunks side by side); d_a / d_b / d_c are host arrays of n device pointers.  */
int jh_tgemm_dense_group(jh_ctx* ctx, int32_t n, int32_t M, int32_t N, int32_t K, const float* const* d_a, const float* const* d_b,
                         float* const* d_c, jh_stream stream);

/* Measurement / test hook of the tile engine: per-call-site overrides of the workgroup tile, the K split and the XCD order,
 * "<site id | *>:<TM>x<TN>[:s<splits>][:x<0|1>],..." (TM x TN in 16 x 16 fragments per wave: 2x2 = 64 x 64, 4x2 = 128 x 64, 2x4 = 64 x 128);
 * the same grammar as the environment variable JH_TGEMM_CFG; "" or NULL restores the defaults.  No reference counterpart.  */
int jh_tgemm_set_cfg(const char* cfg);

/* ------------------------------------------------------------------ asynchronous actor -> learner staging
 * Replaces the async path's transport (run_mode.py:212-363 async_distributed_train: Ray actors -> manager
 * process -> multiprocessing trans_queue -> `gather_thread` spinning on flags, process.py:7-31,82-97) for
 * many-actor / one-learner agents (Ape-X, ape_x.py:174-199 actor-side priorities): one bounded lock-free
 * multi-producer / single-consumer ring of transitions in pinned host memory with the replay store's column
 * layout.  Actor threads produce rows (+ priorities); the learner thread drains what is published with
 * hipMemcpyAsync straight from the ring's pinned slots into the device store (+ the sum-tree leaves).
 * ctx may be NULL for a host-only ring (pageable memory, jh_ring_consume_host only).                      */
typedef struct jh_ring jh_ring;
int jh_ring_create(jh_ctx* ctx, int64_t slots, int32_t n_cols, const jh_col_desc* cols, int32_t with_priority, jh_ring** out);
void jh_ring_destroy(jh_ring* r);
/* Any thread.  n <= slots rows, h_cols[c] = n rows of column c, h_prio[n] when the ring carries priorities.
 * Blocks while the ring is full; timeout_ms >= 0 bounds that wait (JH_ERR_STATE, nothing written), < 0 forever. */
int jh_ring_produce(jh_ring* r, int64_t n, const void* const* h_cols, const double* h_prio, int32_t timeout_ms);
/* The consumer thread.  Appends every published row (<= max_rows; <= 0: as many as the store holds) to the device
 * store and, with per != NULL, pushes the leaves (actor priorities, or max_priority for a ring without).
 * Asynchronous on `stream`; slots are recycled when the copies have executed.  *n_out = rows taken.        */
int jh_ring_drain(jh_ring* r, jh_store* s, jh_per* per, int64_t max_rows, jh_stream stream, int64_t* n_out);
/* Consumer thread: recycle the slots of drains whose copies have executed (wait != 0: wait for all of them).  */
int jh_ring_reclaim(jh_ring* r, int32_t wait);
/* Host-side consumer (tests / CPU plumbing): rows are copied to h_out_cols and their slots recycled at once. */
int jh_ring_consume_host(jh_ring* r, int64_t max_rows, void* const* h_out_cols, double* h_prio_out, int64_t* n_out);
int jh_ring_stats(jh_ring* r, int64_t* produced, int64_t* drained, double* producer_wait_ms);

/* ------------------------------------------------------------------ device-resident actor -> replay feed
 * For N lockstep actors whose frame stacks are already in HBM (they were uploaded for the batched acting forward):
 * replaces, per tick, the env wrapper's stack bookkeeping as seen by the replay (core/env/atari.py:145-149: consecutive
 * stacks share C - 1 frames), Ape-X's per-actor n-step deque with actor-side priorities (core/agent/ape_x.py:174-199)
 * and the transfer of both full stacks of every transition to the learner (process.py:82-97, run_mode.py:300-330).
 * Planes go into a caller-owned plane pool [n_actors * planes_per_actor][plane_bytes] (each actor owns a private ring of
 * planes_per_actor slots); a stack is C slot numbers.  A stack that is not its predecessor shifted by one frame (reset,
 * frame skip glitch, ...) simply appends all C planes: the comparison is bytewise, so the stored content is always exact.
 * window_ticks: for how many ticks a plane must survive (buffer rows / n_actors + n_step + C + in-flight ticks); when an
 * actor allocates more than planes_per_actor planes inside that window, bit 0 of the flags word is set (jh_feed_state).
 * jh_feed_tick: d_obs uint8 [N][C][plane_bytes] = the stacks acted on this tick, d_prev_obs = those of the previous tick
 * (NULL on the first), d_action int64 [N] / d_q float [N] = action taken and its Q (jh_value_act), h_reward / h_done
 * float [N] (host) = the env's answer.  From tick n_step on (*emitted = N, else 0) the outputs hold one n-step
 * transition per actor: slot numbers of state_t and state_{t+n} int64 [N][C], action_t int64 [N], reward float [N][n],
 * done uint8 [N][n], priority float64 [N] = |G_n - q_t| + prio_eps, all on the device, enqueued on `stream`.          */
typedef struct jh_feed jh_feed;
int jh_feed_create(jh_ctx* ctx, int32_t n_actors, int32_t C, int64_t plane_bytes, int32_t n_step, float gamma,
                   int64_t planes_per_actor, int64_t window_ticks, jh_feed** out);
void jh_feed_destroy(jh_feed* f);
int jh_feed_tick(jh_feed* f, const uint8_t* d_obs, const uint8_t* d_prev_obs, uint8_t* d_pool, const int64_t* d_action,
                 const float* d_q, const float* h_reward, const float* h_done, double prio_eps, int64_t* d_state_ids,
                 int64_t* d_next_ids, int64_t* d_action_out, float* d_reward_out, uint8_t* d_done_out, double* d_prio_out,
                 int32_t* emitted, jh_stream stream);
/* jh_feed_tick in two halves (the acting forward runs between them):
 *   jh_feed_push_stacks  stack mode: the stacks as they were uploaded for the forward (what jh_feed_tick does first)
 *   jh_feed_push_frames  frame mode: the env hands over only the NEWEST plane of every actor, d_frames uint8 [N][plane_bytes]
 *                        (device), and h_reset uint8 [N] (host; 1 = the env was reset: the stack is that frame C times,
 *                        core/env/atari.py:112); the stacks [N][C][plane_bytes] for the forward are rebuilt into d_stack_out
 *                        by gather from the plane pool.  7 KB instead of 28 KB per env step over PCIe at Atari shapes,
 *                        and the wrapper's np.concatenate of atari.py:147 has no counterpart at all.
 *   jh_feed_emit         the second half: action / Q / reward / done -> n-step rows + priorities
 * A feed uses ONE of the two push modes for its whole life.                                                              */
int jh_feed_push_stacks(jh_feed* f, const uint8_t* d_obs, const uint8_t* d_prev_obs, uint8_t* d_pool, jh_stream stream);
int jh_feed_push_frames(jh_feed* f, const uint8_t* d_frames, const uint8_t* h_reset, uint8_t* d_pool, uint8_t* d_stack_out,
                        jh_stream stream);
int jh_feed_emit(jh_feed* f, const int64_t* d_action, const float* d_q, const float* h_reward, const float* h_done,
                 double prio_eps, int64_t* d_state_ids, int64_t* d_next_ids, int64_t* d_action_out, float* d_reward_out,
                 uint8_t* d_done_out, double* d_prio_out, int32_t* emitted, jh_stream stream);
/* Blocking read of the flags word (bit 0: plane ring overrun) and the number of planes written so far. */
int jh_feed_state(jh_feed* f, int32_t* h_flags, int64_t* h_planes_written, jh_stream stream);
/* Checkpoint of the feed itself (the reference cannot resume a replay at all, core/agent/dqn.py:184-199): rolling stacks' slot
 * numbers, plane cursors + history, rolling action / reward / done / q windows, flags, tick.  Together with the plane pool, the
 * store's rows and the sum tree, a feed of the SAME geometry continues exactly where the saved one stood.  h buffers hold
 * jh_feed_state_bytes(f) bytes; both calls synchronise `stream` and must not race a tick.                               */
int64_t jh_feed_state_bytes(const jh_feed* f);
int jh_feed_save(jh_feed* f, void* h_out, int64_t bytes, jh_stream stream);
int jh_feed_load(jh_feed* f, const void* h_in, int64_t bytes, jh_stream stream);

/* ------------------------------------------------------------------ data-parallel learners: the collective
 * The reference has ONE learner and no collective (its "distributed" mode is Ray actors feeding that learner,
 * manager/distributed_manager.py, process.py); BASELINE.json's north star re-expresses Ape-X's many-actor / one-learner
 * layout as one learner per GPU that averages gradients between backward and clip_grad_norm_ / optimizer.step()
 * (core/agent/ppo.py:164-169, ape_x.py:118-122, rainbow.py:236-238).  This is that step: one RCCL communicator per
 * process (one rank per GPU, xGMI inside a node), bound at run time (dlopen of librccl.so.1: single-GPU users need no RCCL).
 *   jh_comm_unique_id   rank 0 creates the 128-byte id; the caller ships it to the other ranks (any side channel:
 *                       a file, MPI, torch.distributed's store).
 *   jh_comm_create      collective over all ranks (ncclCommInitRank on ctx's device).
 *   jh_comm_allreduce_mean_f32   in place, d_bucket[i] <- mean over ranks (ncclAvg), enqueued on `stream`: the flat
 *                       gradient bucket of jh_pponet_* / jh_rbnet_* between their backward and their optimizer step.
 *                       Capturable into a hipGraph (every rank must then replay in the same order).
 *   jh_comm_broadcast   bytes from `root` to all (identical initial weights / optimizer moments).
 *   jh_comm_allgather_f64   n doubles per rank -> [nranks][n] (sharded PER: {root, count, min sampled priority},
 *                       core/buffer/per_buffer.py:88-94 evaluated for the logical buffer over all shards).          */
#define JH_COMM_ID_BYTES 128
typedef struct jh_comm jh_comm;
int jh_comm_unique_id(void* h_id128);
int jh_comm_create(jh_ctx* ctx, int32_t nranks, int32_t rank, const void* h_id128, jh_comm** out);
void jh_comm_destroy(jh_comm* m);
int jh_comm_info(const jh_comm* m, int32_t* nranks, int32_t* rank);
int jh_comm_allreduce_mean_f32(jh_comm* m, float* d_bucket, int64_t n, jh_stream stream);
int jh_comm_broadcast(jh_comm* m, void* d_buf, int64_t bytes, int32_t root, jh_stream stream);
int jh_comm_allgather_f64(jh_comm* m, const double* d_in, double* d_out, int64_t n, jh_stream stream);

/* ---- the same collectives through peer pointers (round 6): every rank of ONE node maps its peers' arenas (hipIpc) and the gradient
 * bucket's all-reduce is two one-hop exchanges -- reduce-scatter by reading slice `rank` of every peer's bucket, all-gather by reading the
 * peers' reduced slices -- inside two launches of this library, no collective library's launch and no ring (csrc/jh_peer.hip).  Chosen with
 * JH_DP_COLLECTIVE=peer; RCCL (jh_comm_*) stays the default.  No reference counterpart.
 *   jh_peer_create   this rank's arena for buckets of up to max_floats floats
 *   jh_peer_handle   its 64-byte hipIpcMemHandle_t; the caller ships all ranks' handles to every rank (any side channel)
 *   jh_peer_connect  h_handles: nranks x 64 bytes in rank order (the own entry is ignored)
 *   jh_peer_allreduce_mean_f32   in place, d_bucket[i] <- mean over ranks; every slice is summed in rank order by ONE rank: identical bits on
 *                    all ranks.  Capturable (sequence numbers advance on the device).  d_bucket 16-byte aligned.
 *   jh_peer_allreduce_small_f32  <= 16 floats, sum (mean != 0: mean) over ranks in rank order, one single-workgroup launch
 *   jh_peer_status   bounded waits (~2 s) that gave up since creation, completed all-reduces                                        */
#define JH_PEER_HANDLE_BYTES 64
typedef struct jh_peer jh_peer;
int jh_peer_create(jh_ctx* ctx, int32_t nranks, int32_t rank, int64_t max_floats, jh_peer** out);
int jh_peer_handle(jh_peer* p, void* h_handle64);
int jh_peer_connect(jh_peer* p, const void* h_handles);
int jh_peer_allreduce_mean_f32(jh_peer* p, float* d_bucket, int64_t n, jh_stream stream);
int jh_peer_allreduce_small_f32(jh_peer* p, float* d_vals, int32_t n, int32_t mean, jh_stream stream);
int jh_peer_status(jh_peer* p, int32_t* timeouts, int64_t* completed);
void jh_peer_destroy(jh_peer* p);


#ifdef __cplusplus
}
#endif
#endif /* JORLDY_HIP_H */
