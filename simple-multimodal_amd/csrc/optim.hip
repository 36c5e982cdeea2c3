// Training-step tail on the flat arenas (reference recipe
// training/advanced_trainer.py:85-110,168-182): global gradient norm, clip, AdamW — two streaming kernels
// over the fp32 gradient / master arenas.  The AdamW kernel also writes the bf16 shadow the MFMA GEMMs
// read, so a training step needs no separate fp32 -> bf16 weight cast.  HBM-bound: 16 B/element read
// (p, g, m, v) + 14 B/element written (p, m, v, shadow).
// Hyper-parameters live in a small DEVICE array so that a captured hipGraph can be replayed with a new
// learning rate / bias correction every step (the host updates the array before the replay).
#include "mmf_internal.h"

namespace {

constexpr int OPT_THREADS = 256;

#ifndef MMF_OPT_UNROLL
#define MMF_OPT_UNROLL 2          // 16-byte vectors per thread and stream in flight per iteration
#endif
#ifndef MMF_OPT_NT
#define MMF_OPT_NT 1              // non-temporal loads / stores for what the step touches once (masters, moments, gradients)
#endif
template <typename T> __device__ __forceinline__ T ld_stream(const T* p) {
#if MMF_OPT_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T> __device__ __forceinline__ void st_stream(T* p, const T& v) {
#if MMF_OPT_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

__global__ __launch_bounds__(OPT_THREADS)
void sqnorm_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[OPT_THREADS / 64];
  constexpr int U = 2 * MMF_OPT_UNROLL;
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * OPT_THREADS;
  float s = 0.f;
  int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x;
  for (; i + (U - 1) * stride < nvec; i += U * stride) {     // U loads in flight per thread
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4_t*>(x + (i + u * stride) * 4);
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
  }
  for (; i < nvec; i += stride) {
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (t < n) s += x[t] * x[t];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// hp: [0] lr  [1] beta1  [2] beta2  [3] eps  [4] weight_decay  [5] 1-beta1^t  [6] 1-beta2^t
//     [7] max_grad_norm (<= 0: no clipping)  [8] grad_scale (e.g. 1/world for an un-averaged all-reduce)
__global__ __launch_bounds__(OPT_THREADS)
void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                  float* __restrict__ v, unsigned short* __restrict__ shadow, int64_t n,
                  const float* __restrict__ hp, const float* __restrict__ gnorm_sq) {
  const float lr = hp[0], b1 = hp[1], b2 = hp[2], eps = hp[3], wd = hp[4], bc1 = hp[5], bc2 = hp[6];
  float gs = hp[8];
  if (hp[7] > 0.f && gnorm_sq) {      // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), <= 1
    const float norm = sqrtf(*gnorm_sq) * fabsf(gs);
    gs *= fminf(1.f, hp[7] / (norm + 1e-6f));
  }
  const float step = lr / bc1, rs2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * OPT_THREADS;
  auto update = [&](f32x4_t& pp, const f32x4_t& gg, f32x4_t& mm, f32x4_t& vv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * gs;
      mm[e] = b1 * mm[e] + (1.f - b1) * ge;
      vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
      pp[e] = pp[e] * decay - step * mm[e] / (sqrtf(vv[e]) * rs2 + eps);
    }
  };
  constexpr int U = MMF_OPT_UNROLL;
  int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x;
  for (; i + (U - 1) * stride < nvec; i += U * stride) {      // 4 U loads in flight per thread before the first use
    f32x4_t pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t o = (i + u * stride) * 4;
      pp[u] = ld_stream(reinterpret_cast<const f32x4_t*>(p + o));
      gg[u] = ld_stream(reinterpret_cast<const f32x4_t*>(g + o));
      mm[u] = ld_stream(reinterpret_cast<const f32x4_t*>(m + o));
      vv[u] = ld_stream(reinterpret_cast<const f32x4_t*>(v + o));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t o = (i + u * stride) * 4;
      update(pp[u], gg[u], mm[u], vv[u]);
      st_stream(reinterpret_cast<f32x4_t*>(p + o), pp[u]);
      st_stream(reinterpret_cast<f32x4_t*>(m + o), mm[u]);
      st_stream(reinterpret_cast<f32x4_t*>(v + o), vv[u]);
      const u32x2_t sh = {pack_bf16x2(pp[u][0], pp[u][1]), pack_bf16x2(pp[u][2], pp[u][3])};
      *reinterpret_cast<u32x2_t*>(shadow + o) = sh;              // the next forward's GEMMs read the shadow: plain store
    }
  }
  for (; i < nvec; i += stride) {
    f32x4_t pp = *reinterpret_cast<const f32x4_t*>(p + i * 4);
    const f32x4_t gg = *reinterpret_cast<const f32x4_t*>(g + i * 4);
    f32x4_t mm = *reinterpret_cast<const f32x4_t*>(m + i * 4);
    f32x4_t vv = *reinterpret_cast<const f32x4_t*>(v + i * 4);
    update(pp, gg, mm, vv);
    *reinterpret_cast<f32x4_t*>(p + i * 4) = pp;
    *reinterpret_cast<f32x4_t*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4_t*>(v + i * 4) = vv;
    const u32x2_t sh = {pack_bf16x2(pp[0], pp[1]), pack_bf16x2(pp[2], pp[3])};
    *reinterpret_cast<u32x2_t*>(shadow + i * 4) = sh;
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (t < n) {
      const float ge = g[t] * gs;
      const float mt = b1 * m[t] + (1.f - b1) * ge, vt = b2 * v[t] + (1.f - b2) * ge * ge;
      const float pt = p[t] * decay - step * mt / (sqrtf(vt) * rs2 + eps);
      p[t] = pt; m[t] = mt; v[t] = vt; shadow[t] = f32_to_bf16_bits(pt);
    }
  }
}

// Device-resident step counter + schedule: ONE thread advances the optimiser's step count and derives this step's
// learning rate, beta1 and bias corrections into the hyper-parameter array the update kernel reads, so nothing about
// a step's hyper-parameters crosses the PCIe bus and a captured training step replays with the right values however
// far the host runs ahead (a pinned host buffer rewritten per step is read when the GPU executes the copy, not when
// the host enqueues it).  sched: [0] mode (0 = constant hp[0], 1 = OneCycleLR cos, three_phase = False)
// [1] max_lr [2] total_steps [3] pct_start [4] div_factor [5] final_div_factor [6] cycle_momentum (0 / 1)
// [7] base_momentum [8] max_momentum   (reference recipe training/advanced_trainer.py:102-110: OneCycleLR with its
// defaults, i.e. cycle_momentum=True — Adam's beta1 runs 0.95 -> 0.85 -> 0.95 against the learning rate, and
// torch.optim.Adam forms its bias correction from the CURRENT beta1: 1 - beta1(t)^t).  Arithmetic in double like
// torch.optim.lr_scheduler.OneCycleLR.
__global__ void adamw_advance_kernel(long long* __restrict__ step, float* __restrict__ hp,
                                     const double* __restrict__ sched) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long t = *step + 1;                 // number of the optimiser step about to run, 1-based
  *step = t;
  double b1 = hp[1];
  const double b2 = hp[2];
  if (sched[0] == 1.0) {
    const double max_lr = sched[1], total = sched[2], pct = sched[3];
    const double initial = max_lr / sched[4], minimum = initial / sched[5];
    const double up_end = pct * total - 1.0, down_end = total - 1.0, s = (double)(t - 1);
    const double kPi = 3.14159265358979323846;
    const bool up = s <= up_end || up_end >= down_end;
    const double q = up ? (up_end > 0.0 ? s / up_end : 1.0) : fmin(1.0, (s - up_end) / (down_end - up_end));
    const double w = (cos(kPi * q) + 1.0) / 2.0;                     // anneal(start, end) = end + (start - end) w
    hp[0] = (float)(up ? max_lr + (initial - max_lr) * w : minimum + (max_lr - minimum) * w);
    if (sched[6] != 0.0) {
      const double base_m = sched[7], max_m = sched[8];
      b1 = up ? base_m + (max_m - base_m) * w : max_m + (base_m - max_m) * w;
      hp[1] = (float)b1;
    }
  }
  hp[5] = (float)(1.0 - pow(b1, (double)t));
  hp[6] = (float)(1.0 - pow(b2, (double)t));
}

#ifndef MMF_OPT_GRID
#define MMF_OPT_GRID 2048         // workgroups of the streaming kernels (8 per CU)
#endif
inline int opt_grid(int64_t n) {
  int64_t g = ((n >> 2) + OPT_THREADS - 1) / OPT_THREADS;
  if (g > MMF_OPT_GRID) g = MMF_OPT_GRID;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int mmf_sqnorm_f32(const float* x, int64_t n, float* out, void* stream) {
  if (n <= 0) return MMF_OK;
  if (!x || !out || !mmf_aligned16(x)) MMF_FAIL(MMF_E_ALIGN, "mmf_sqnorm_f32: null or unaligned pointer");
  hipLaunchKernelGGL(sqnorm_kernel, dim3(opt_grid(n)), dim3(OPT_THREADS), 0, static_cast<hipStream_t>(stream), x, n, out);
  MMF_CHECK_LAUNCH("mmf_sqnorm_f32");
  return MMF_OK;
}

extern "C" int mmf_adamw_advance(int64_t* step, float* hparams, const double* sched, void* stream) {
  if (!step || !hparams || !sched) MMF_FAIL(MMF_E_SHAPE, "mmf_adamw_advance: null pointer");
  hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<long long*>(step), hparams, sched);
  MMF_CHECK_LAUNCH("mmf_adamw_advance");
  return MMF_OK;
}

extern "C" int mmf_adamw_step(float* master, const float* grad, float* exp_avg, float* exp_avg_sq,
                              void* shadow_bf16, int64_t n, const float* hparams, const float* gnorm_sq,
                              void* stream) {
  if (n <= 0) return MMF_OK;
  if (!master || !grad || !exp_avg || !exp_avg_sq || !shadow_bf16 || !hparams)
    MMF_FAIL(MMF_E_SHAPE, "mmf_adamw_step: null pointer");
  if (!mmf_aligned16(master) || !mmf_aligned16(grad) || !mmf_aligned16(exp_avg) || !mmf_aligned16(exp_avg_sq) ||
      (reinterpret_cast<uintptr_t>(shadow_bf16) & 7))
    MMF_FAIL(MMF_E_ALIGN, "mmf_adamw_step: arenas must be 16-byte aligned");
  hipLaunchKernelGGL(adamw_kernel, dim3(opt_grid(n)), dim3(OPT_THREADS), 0, static_cast<hipStream_t>(stream),
                     master, grad, exp_avg, exp_avg_sq, static_cast<unsigned short*>(shadow_bf16), n, hparams, gnorm_sq);
  MMF_CHECK_LAUNCH("mmf_adamw_step");
  return MMF_OK;
}
