// Internal interface of the four- / five-launch PPO minibatch update (jh_ppo_mb.hip), driven by jh_mlp.hip.
#pragma once
#include "jh_common.h"

// Flat list of the head outputs (discrete: A logits, value; continuous: A mu, A log_std, value):
// weight row / bias / weight-gradient row / bias-gradient of output o.
struct PmbHeads {
  const float* w[8];
  const float* b[8];
  float* dw[8];
  float* db[8];
  int n_out;
};

bool jh_pmb_eligible(const jh_pponet* n, int B);
// forward of M rows: n->fwd_part <- per-column-tile partial heads; store_act: n->h1 / n->h2 <- activations.
// h1_in non-null: layer 1 already computed (S > 8), else generated in registers.
int jh_pmb_forward(jh_pponet* n, int M, const float* d_x, const int64_t* d_idx, const PmbHeads& hd, const float* h1_in,
                   bool store_act, hipStream_t st);
int jh_pmb_heads_finish(jh_pponet* n, int M, float* d_head0, float* d_head1, float* d_value, hipStream_t st);
// n->g_all [B][8] (+ n->h1, n->h2 of the last forward) -> gradient bucket except (W1 | b1), whose per-row-tile
// partial sums go to n->part_w1.  emit_ssq: every workgroup that writes gradient tiles also writes their sum of squares to
// n->ssq_part (the fused Adam launch follows with no norm kernel in between)
int jh_pmb_backward(jh_pponet* n, int B, const float* d_x, const int64_t* d_idx, const PmbHeads& hd, bool emit_ssq, hipStream_t st);
// (W1 | b1) gradients <- sum of the partials; with_norm: also the global-norm partials + Adam step advance
int jh_pmb_finalize(jh_pponet* n, int B, bool with_norm, hipStream_t st);
