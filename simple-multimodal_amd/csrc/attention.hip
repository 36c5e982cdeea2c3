// Grouped fused attention for the fusion path (q*scale, QK^T, softmax, P.V of F.multi_head_attention_forward as called at
// models/fusion_layers.py:161-163,204): C-ABI entry points and operand validation.  The kernels are in attention2.hip
// (forward with log-sum-exp; recompute backward: dQ kernel + dK/dV kernel, no atomics, deterministic).  All attentions of
// a MulT stage (6 cross or 3 self, unequal Tq/Tk) go out as ONE launch.  Rounds 1-2 kept superseded kernel generations here
// and in attention3.hip / attention4.hip behind mmf_attn_select_impl; round 3 removed them (git history has them,
// DESIGN.md section 5 their measurements).
#include "mmf_internal.h"
#include <stdlib.h>

namespace {

int validate(const char* who, const mmf_attn_problem* p, int n, int head_dim, bool bwd) {
  if (!p || n <= 0 || n > MMF_ATTN_MAX_PROBLEMS) MMF_FAIL(MMF_E_SHAPE, "%s: num_problems=%d out of range", who, n);
  if (head_dim != 64 && head_dim != 96) MMF_FAIL(MMF_E_UNSUPPORTED, "%s: head_dim=%d (supported: 64, 96)", who, head_dim);
  for (int i = 0; i < n; ++i) {
    const mmf_attn_problem& q = p[i];
    if (q.B <= 0 || q.H <= 0 || q.Tq <= 0 || q.Tk <= 0) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: B=%d H=%d Tq=%d Tk=%d", who, i, q.B, q.H, q.Tq, q.Tk);
    const int w = q.H * head_dim;
    if (q.ldq < w || q.ldk < w || q.ldv < w || q.ldo < w || (q.ldq & 7) || (q.ldk & 7) || (q.ldv & 7) || (q.ldo & 7))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: row strides must be >= H*head_dim and multiples of 8", who, i);
    if (!q.Q || !q.K || !q.V || !q.O || !q.LSE) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: null operand", who, i);
    if (!mmf_aligned16(q.Q) || !mmf_aligned16(q.K) || !mmf_aligned16(q.V) || !mmf_aligned16(q.O))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: Q/K/V/O must be 16-byte aligned", who, i);
    if (bwd) {
      if (!q.dO || !q.delta || !q.dQ || !q.dK || !q.dV) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: null gradient operand", who, i);
      if (!mmf_aligned16(q.dO) || !mmf_aligned16(q.dQ) || !mmf_aligned16(q.dK) || !mmf_aligned16(q.dV))
        MMF_FAIL(MMF_E_ALIGN, "%s[%d]: gradient pointers must be 16-byte aligned", who, i);
    }
  }
  return MMF_OK;
}

}  // namespace

int mmf_attn_fwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s);
int mmf_attn_bwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s);

// Tuning hook kept for ABI stability: one kernel generation is left, so only 0 (automatic) and 2 are accepted.
extern "C" int mmf_attn_select_impl(int impl) {
  if (impl != 0 && impl != 2)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_attn_select_impl: impl=%d (only generation 2 is built: 0 or 2)", impl);
  return MMF_OK;
}

extern "C" int mmf_attn_fwd_grouped_ex(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                       float scale, float dropout_p, const uint64_t* rng_state, uint32_t site,
                                       void* stream) {
  if (int rc = validate("mmf_attn_fwd_grouped", problems, num_problems, head_dim, false)) return rc;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f || (dropout_p > 0.f && !rng_state))
    MMF_FAIL(MMF_E_SHAPE, "mmf_attn_fwd_grouped_ex: dropout needs 0 <= p < 1 and an rng_state");
  return mmf_attn_fwd2_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site,
                              static_cast<hipStream_t>(stream));
}

extern "C" int mmf_attn_bwd_grouped_ex(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                       float scale, float dropout_p, const uint64_t* rng_state, uint32_t site,
                                       void* stream) {
  if (int rc = validate("mmf_attn_bwd_grouped", problems, num_problems, head_dim, true)) return rc;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f || (dropout_p > 0.f && !rng_state))
    MMF_FAIL(MMF_E_SHAPE, "mmf_attn_bwd_grouped_ex: dropout needs 0 <= p < 1 and an rng_state");
  return mmf_attn_bwd2_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site,
                              static_cast<hipStream_t>(stream));
}

extern "C" int mmf_attn_fwd_grouped(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                    float scale, void* stream) {
  return mmf_attn_fwd_grouped_ex(problems, num_problems, head_dim, scale, 0.f, nullptr, 0u, stream);
}

extern "C" int mmf_attn_bwd_grouped(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                    float scale, void* stream) {
  return mmf_attn_bwd_grouped_ex(problems, num_problems, head_dim, scale, 0.f, nullptr, 0u, stream);
}
