// The tail of the PPO loss shared by the loss kernels (jh_ppo.hip) and by the backward's first kernel when it finishes a one-launch loss itself
// (jh_mlp.hip: jh_mlp_heads_bwd_dh_kernel): the partials' reduction in the two-pass order, the statistics row, the critic's branch weights.
#pragma once
#include "jh_common.h"

#define PPO_NPART 6

// One statistics row [8] = two 16-byte granules {loss, actor, critic, entropy} | {max_ratio, min_prob, c1, c2}.  The host may be
// spinning on the row in device-mapped memory (the last update of a learn(), core/agent/ppo.py: _await_mapped_stats): it waits for
// ONE element of EACH granule to change -- critic [2] and c2 [7], means of squares, never the -1 the host arms them with -- and each
// granule is one 16-byte store, which lands whole (MI355X_MICROARCH.md, hand-off granules).  Rounds 1-4 wrote eight scalars with a
// __threadfence_system() in front of [7]: an L2 write-back + invalidate (~2-3.5 us) on the critical path of EVERY minibatch's loss
// launch, for a host that only ever waits on the last one.  A row that is not 16-byte aligned (public API, any pointer) keeps that form.
// (noinline: inlined next to the 16-byte form, hipcc merged the two tails into dwordx4 + dwordx3 + dword stores -- a torn granule)
static __device__ __attribute__((noinline)) void jh_ppo_stats_row_unaligned(float* stats, float loss, float actor, float critic, float entropy, float max_ratio,
                                                                     float min_prob, float c1, float c2) {
  stats[0] = loss; stats[1] = actor; stats[2] = critic; stats[3] = entropy; stats[4] = max_ratio; stats[5] = min_prob; stats[6] = c1;
  __threadfence_system();
  stats[7] = c2;
}
__device__ __forceinline__ static void jh_ppo_stats_row(float* stats, float loss, float actor, float critic, float entropy, float max_ratio, float min_prob,
                                                 float c1, float c2) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  if ((reinterpret_cast<uintptr_t>(stats) & 15) == 0) {
    const f32x4 ga = {loss, actor, critic, entropy}, gb = {max_ratio, min_prob, c1, c2};
    // exactly ONE 16-byte store per granule, whatever the optimizer thinks of its neighbours
    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16" ::"v"(stats), "v"(ga), "v"(gb) : "memory");
  } else {
    jh_ppo_stats_row_unaligned(stats, loss, actor, critic, entropy, max_ratio, min_prob, c1, c2);
  }
}

__device__ __forceinline__ static void ppo_finish_stats(float s_smin, float s_e1, float s_e2, float s_ent, float max_ratio,
                                                 float min_prob, int B, int ent_count, float vf, float ent, float& w1,
                                                 float& w2, float* stats) {
  const float actor = -(s_smin / (float)B);
  const float c1 = s_e1 / (float)B, c2 = s_e2 / (float)B;
  const float critic = fmaxf(c1, c2);
  w1 = c1 > c2 ? 1.f : (c1 == c2 ? 0.5f : 0.f);
  w2 = 1.f - w1;
  const float entropy_loss = -(s_ent / (float)ent_count);
  if (stats) jh_ppo_stats_row(stats, actor + vf * critic + ent * entropy_loss /* ppo.py:158-162 */, actor, critic, entropy_loss, max_ratio, min_prob, c1, c2);
}

// nb <= 64 partials fit one wave: every wave reduces them by itself with the shuffle tree wave 0 of ppo_reduce_partials would run
// (lane b holds 0 + partial b, the other lanes the identity), and the block-level combine of that form adds the other waves'
// identities -- 0.f + tree, which is what is returned here: bit-identical, without the LDS exchange and its 12 barriers.
__device__ __forceinline__ static void ppo_reduce_partials_wave(const float* __restrict__ partial, int nb, float (&t)[6]) {
  const int lane = threadIdx.x & 63;
  const float* p = partial + (size_t)(lane < nb ? lane : 0) * PPO_NPART;
  float q[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) q[k] = p[k];
  const bool mine = lane < nb;
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = mine ? 0.f + q[k] : 0.f;
  t[4] = mine ? fmaxf(-3.4e38f, q[4]) : -3.4e38f;
  t[5] = mine ? fminf(3.4e38f, q[5]) : 3.4e38f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] += __shfl_xor(t[k], o, 64);
    t[4] = fmaxf(t[4], __shfl_xor(t[4], o, 64));
    t[5] = fminf(t[5], __shfl_xor(t[5], o, 64));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = 0.f + t[k];
  t[4] = fmaxf(-3.4e38f, t[4]);
  t[5] = fminf(3.4e38f, t[5]);
}


// What the backward's first kernel needs to finish a one-launch loss of nb <= 64 workgroups by itself (partial == null: somebody else did)
struct PpoFinish {
  const float* partial;  // [nb][PPO_NPART]
  int nb, B, ent_count;
  float vf, ent;
  float* stats;          // [8] or null
};
