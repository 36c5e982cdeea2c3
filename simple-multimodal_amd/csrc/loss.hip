// Two (B, .)-row kernels of the training step's tail and head (round 4): each replaces a chain of stock elementwise launches
// whose work is a few hundred bytes — at these sizes a step pays ~5 us per LAUNCH, so the chains cost 100-200 us each.
//
//   fusion_loss_kernel        reference training/advanced_trainer.py:139-166: CrossEntropy(label_smoothing) over (B, C <= 64) logits
//                             plus weighted scalar terms (0.1 x the three contrastive losses, 0.5 x distillation), value AND the
//                             gradient with respect to the logits in one launch (torch: log_softmax, nll_loss, smoothing
//                             sum, three scalar adds, and the same again backwards: ~35 launches).
//   modality_dropout_kernel   reference models/encoders.py:289-321: per-sample Bernoulli keep masks over the three modalities,
//                             no 1 / (1 - p) rescale, a sample that lost all three gets one back at random; masks drawn from the
//                             build's counter-based hash of (device step state, site, sample) like every other dropout site, the
//                             three (B, d) tensors scaled in the same launch (torch: rand, compare, randint, one_hot, where, any
//                             and three strided copies in front of three row-mask launches).  The same kernel applies a given mask
//                             (backward).
#include "mmf_internal.h"

namespace {

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_MAX_EXTRA = 8;

struct LossArgs {
  const float* logits;
  const long long* targets;
  float* loss;
  float* dlogits;
  const float* extra[LOSS_MAX_EXTRA];
  float extra_w[LOSS_MAX_EXTRA];
  int B, C, ldl, n_extra;
  float smoothing;
};

__global__ __launch_bounds__(LOSS_THREADS)
void fusion_loss_kernel(const LossArgs a) {
  __shared__ float part[LOSS_THREADS / 64];
  const int tid = threadIdx.x;
  const float eps = a.smoothing, invB = 1.f / (float)a.B, invC = 1.f / (float)a.C;
  float acc = 0.f;
  for (int b = tid; b < a.B; b += LOSS_THREADS) {            // one sample per thread: C <= 64 logits
    const float* l = a.logits + (size_t)b * a.ldl;
    float mx = -INFINITY;
    for (int c = 0; c < a.C; ++c) mx = fmaxf(mx, l[c]);
    float se = 0.f, sl = 0.f;
    for (int c = 0; c < a.C; ++c) { se += __expf(l[c] - mx); sl += l[c]; }
    const float lse = mx + __logf(se);
    const int y = (int)a.targets[b];
    const float nll = lse - l[y];                            // -log p_y
    const float smooth = lse - sl * invC;                    // -(1 / C) sum_c log p_c
    acc += (1.f - eps) * nll + eps * smooth;
    if (a.dlogits) {
      float* d = a.dlogits + (size_t)b * a.C;
      for (int c = 0; c < a.C; ++c)
        d[c] = (__expf(l[c] - lse) - ((c == y ? 1.f - eps : 0.f) + eps * invC)) * invB;
    }
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) t += part[w];
    t *= invB;
    for (int j = 0; j < a.n_extra; ++j) t += a.extra_w[j] * a.extra[j][0];
    a.loss[0] = t;
  }
}

struct ModDropArgs {
  const float* x[3];
  float* y[3];
  float* keep;                       // [B][3], written (draw) or read (apply)
  const unsigned long long* rng_state;
  unsigned thresh, site;
  int B, d, draw;
};

// one workgroup per sample
__global__ __launch_bounds__(256)
void modality_dropout_kernel(const ModDropArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float k[3];
  if (a.draw) {
    const unsigned key = mmf_rng_key(*a.rng_state, a.site, 0u);
    bool kp[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) kp[m] = mmf_keep(key, (unsigned)(3 * b + m), a.thresh);
    if (!(kp[0] || kp[1] || kp[2])) kp[mmf_mix32(key ^ (0x51ed270bu + (unsigned)b)) % 3u] = true;   // :308-314: one modality comes back
#pragma unroll
    for (int m = 0; m < 3; ++m) k[m] = kp[m] ? 1.f : 0.f;
    if (tid < 3) a.keep[3 * b + tid] = k[tid];
  } else {
#pragma unroll
    for (int m = 0; m < 3; ++m) k[m] = a.keep[3 * b + m];
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const float* x = a.x[m] + (size_t)b * a.d;
    float* y = a.y[m] + (size_t)b * a.d;
    for (int i = 4 * tid; i < a.d; i += 4 * 256) {
      f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + i);
      v *= k[m];
      *reinterpret_cast<f32x4_t*>(y + i) = v;
    }
  }
}

}  // namespace

extern "C" int mmf_fusion_loss(const float* logits, int ldl, const int64_t* targets, int B, int C, float label_smoothing,
                               const float* const* extra, const float* extra_w, int n_extra, float* loss, float* dlogits,
                               void* stream) {
  if (!logits || !targets || !loss || B <= 0 || C <= 0 || C > 64 || ldl < C)
    MMF_FAIL(MMF_E_SHAPE, "mmf_fusion_loss: B=%d C=%d (1..64) ldl=%d", B, C, ldl);
  if (n_extra < 0 || n_extra > LOSS_MAX_EXTRA || (n_extra && (!extra || !extra_w)))
    MMF_FAIL(MMF_E_SHAPE, "mmf_fusion_loss: n_extra=%d out of range [0,%d]", n_extra, LOSS_MAX_EXTRA);
  if (!(label_smoothing >= 0.f) || label_smoothing >= 1.f) MMF_FAIL(MMF_E_SHAPE, "mmf_fusion_loss: label_smoothing must be in [0, 1)");
  LossArgs a = {};
  a.logits = logits; a.targets = reinterpret_cast<const long long*>(targets); a.loss = loss; a.dlogits = dlogits;
  a.B = B; a.C = C; a.ldl = ldl; a.n_extra = n_extra; a.smoothing = label_smoothing;
  for (int j = 0; j < n_extra; ++j) {
    if (!extra[j]) MMF_FAIL(MMF_E_SHAPE, "mmf_fusion_loss: extra[%d] is null", j);
    a.extra[j] = extra[j]; a.extra_w[j] = extra_w[j];
  }
  hipLaunchKernelGGL(fusion_loss_kernel, dim3(1), dim3(LOSS_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_fusion_loss");
  return MMF_OK;
}

extern "C" int mmf_modality_dropout(const float* const* x, float* const* y, float* keep, int B, int d, float p,
                                    const uint64_t* rng_state, uint32_t site, int draw, void* stream) {
  if (!x || !y || !keep || B <= 0 || d <= 0 || (d & 3)) MMF_FAIL(MMF_E_SHAPE, "mmf_modality_dropout: B=%d d=%d (d %% 4 == 0)", B, d);
  if (draw && (!rng_state || !(p >= 0.f) || p >= 1.f)) MMF_FAIL(MMF_E_SHAPE, "mmf_modality_dropout: drawing needs rng_state and 0 <= p < 1");
  ModDropArgs a = {};
  for (int m = 0; m < 3; ++m) {
    if (!x[m] || !y[m] || !mmf_aligned16(x[m]) || !mmf_aligned16(y[m])) MMF_FAIL(MMF_E_ALIGN, "mmf_modality_dropout: null or misaligned operand %d", m);
    a.x[m] = x[m]; a.y[m] = y[m];
  }
  a.keep = keep; a.rng_state = reinterpret_cast<const unsigned long long*>(rng_state);
  a.thresh = draw ? mmf_drop_thresh(p) : 0u; a.site = site; a.B = B; a.d = d; a.draw = draw;
  hipLaunchKernelGGL(modality_dropout_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_modality_dropout");
  return MMF_OK;
}
