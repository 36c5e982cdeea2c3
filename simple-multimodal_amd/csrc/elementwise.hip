// HBM-bound streaming helpers of the fusion path: casts, 3-way residual sum
// (models/fusion_layers.py:156-158), mean over T (:166-168) and its backward, column sums (bias
// gradients), relu backward.  All use 16-byte accesses per lane and grid-stride loops capped at
// 2048 workgroups (guide: Guideline 11/13).
#include "mmf_internal.h"

namespace {

constexpr int EW_THREADS = 256;
inline int ew_grid(int64_t nvec) {
  int64_t g = (nvec + EW_THREADS - 1) / EW_THREADS;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// f32 -> bf16.  A lane converts 4 consecutive floats per access (16-B load, 8-B store): every load
// wave-instruction covers 1 KiB contiguous (the first version gave each lane 8 consecutive floats = two 16-B loads
// 32 B apart, i.e. two half-used passes over the same cache lines: 2.5 TB/s on the 206 MB weight arena), and four
// independent accesses are in flight per lane.
__global__ __launch_bounds__(EW_THREADS)
void cast_f32_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n) {
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    f32x4_t a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4_t*>(src + (i + u * stride) * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<u32x2_t*>(dst + (i + u * stride) * 4) = u32x2_t{pack_bf16x2(a[u][0], a[u][1]), pack_bf16x2(a[u][2], a[u][3])};
  }
  for (; i < nvec; i += stride) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src + i * 4);
    *reinterpret_cast<u32x2_t*>(dst + i * 4) = u32x2_t{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
  }
  if (blockIdx.x == 0) {                              // tail (n % 4 elements)
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (t < n) dst[t] = f32_to_bf16_bits(src[t]);
  }
}

// bf16 -> f32 (* scale): a lane widens 4 consecutive values per access (8-B load, 16-B store), so every store
// wave-instruction covers 1 KiB contiguous; four independent accesses in flight per lane.
__global__ __launch_bounds__(EW_THREADS)
void cast_bf16_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int64_t n, const float scale) {
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    u32x2_t w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x2_t*>(src + (i + u * stride) * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<f32x4_t*>(dst + (i + u * stride) * 4) =
          f32x4_t{bf16lo(w[u][0]), bf16hi(w[u][0]), bf16lo(w[u][1]), bf16hi(w[u][1])} * scale;
  }
  for (; i < nvec; i += stride) {
    const u32x2_t w = *reinterpret_cast<const u32x2_t*>(src + i * 4);
    *reinterpret_cast<f32x4_t*>(dst + i * 4) = f32x4_t{bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])} * scale;
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (t < n) dst[t] = bf16_bits_to_f32(src[t]) * scale;
  }
}

__device__ __forceinline__ unsigned add_pk(unsigned a, unsigned b) {
  return pack_bf16x2(bf16lo(a) + bf16lo(b), bf16hi(a) + bf16hi(b));
}
__device__ __forceinline__ unsigned add3_pk(unsigned a, unsigned b, unsigned c) {
  return pack_bf16x2(bf16lo(a) + bf16lo(b) + bf16lo(c), bf16hi(a) + bf16hi(b) + bf16hi(c));
}

__global__ __launch_bounds__(EW_THREADS)
void add3_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                 const unsigned short* __restrict__ c, unsigned short* __restrict__ y, int64_t n) {
  const int64_t nvec = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
    const u32x4_t va = *reinterpret_cast<const u32x4_t*>(a + i * 8);
    const u32x4_t vb = *reinterpret_cast<const u32x4_t*>(b + i * 8);
    u32x4_t o;
    if (c) {
      const u32x4_t vc = *reinterpret_cast<const u32x4_t*>(c + i * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = add3_pk(va[e], vb[e], vc[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = add_pk(va[e], vb[e]);
    }
    *reinterpret_cast<u32x4_t*>(y + i * 8) = o;
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 3) + threadIdx.x;
    if (t < n) {
      float s = bf16_bits_to_f32(a[t]) + bf16_bits_to_f32(b[t]);
      if (c) s += bf16_bits_to_f32(c[t]);
      y[t] = f32_to_bf16_bits(s);
    }
  }
}

// grouped form: up to MMF_ADD3_MAX problems in one launch (MulT's three residual sums x_m + two cross outputs,
// reference :156-158, sit in the single-stream part of the step where every launch is exposed)
struct Add3Args {
  int n;
  int blk_start[MMF_ADD3_MAX + 1];
  mmf_add3_problem p[MMF_ADD3_MAX];
};
__global__ __launch_bounds__(EW_THREADS)
void add3_grouped_kernel(const Add3Args a) {
  int pi = 0;
  while (pi + 1 < a.n && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_add3_problem& P = a.p[pi];
  const unsigned short* __restrict__ x = static_cast<const unsigned short*>(P.a);
  const unsigned short* __restrict__ y = static_cast<const unsigned short*>(P.b);
  const unsigned short* __restrict__ z = static_cast<const unsigned short*>(P.c);
  unsigned short* __restrict__ o = static_cast<unsigned short*>(P.y);
  const int64_t nvec = P.n >> 3;
  const int nblk = a.blk_start[pi + 1] - a.blk_start[pi], blk = (int)blockIdx.x - a.blk_start[pi];
  const int64_t stride = (int64_t)nblk * EW_THREADS;
  for (int64_t i = (int64_t)blk * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
    const u32x4_t va = *reinterpret_cast<const u32x4_t*>(x + i * 8);
    const u32x4_t vb = *reinterpret_cast<const u32x4_t*>(y + i * 8);
    const u32x4_t vc = *reinterpret_cast<const u32x4_t*>(z + i * 8);
    u32x4_t r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = add3_pk(va[e], vb[e], vc[e]);
    *reinterpret_cast<u32x4_t*>(o + i * 8) = r;
  }
  if (blk == 0) {
    const int64_t t = (nvec << 3) + threadIdx.x;
    if (t < P.n) o[t] = f32_to_bf16_bits(bf16_bits_to_f32(x[t]) + bf16_bits_to_f32(y[t]) + bf16_bits_to_f32(z[t]));
  }
}

__global__ __launch_bounds__(EW_THREADS)
void relu_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ y,
                     unsigned short* __restrict__ dx, int64_t n) {
  // sign/zero test on the bf16 pattern: y > 0  <=>  not negative and not +-0
  auto keep = [](unsigned yy) -> unsigned {
    const unsigned lo = yy & 0xffffu, hi = yy >> 16;
    return ((!(lo & 0x8000u) && (lo & 0x7fffu)) ? 0x0000ffffu : 0u) |
           ((!(hi & 0x8000u) && (hi & 0x7fffu)) ? 0xffff0000u : 0u);
  };
  const int64_t nvec = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  const bool aligned = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(dx)) & 15u) == 0;
  if (aligned) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
      const u32x4_t g = *reinterpret_cast<const u32x4_t*>(dy + i * 8);
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(y + i * 8);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = g[e] & keep(v[e]);
      *reinterpret_cast<u32x4_t*>(dx + i * 8) = o;
    }
  }
  const int64_t start = aligned ? (nvec << 3) : 0;
  for (int64_t i = start + (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
    const unsigned short yy = y[i];
    dx[i] = (!(yy & 0x8000u) && (yy & 0x7fffu)) ? dy[i] : (unsigned short)0;
  }
}

template <typename T>
__global__ __launch_bounds__(EW_THREADS)
void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, unsigned thresh, float scale,
                    const unsigned long long* __restrict__ state, unsigned site) {
  const unsigned key = mmf_rng_key(*state, site, 0u);
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
    const bool keep = mmf_keep(key ^ (unsigned)(i >> 32), (unsigned)i, thresh);
    if (sizeof(T) == 4) {
      const float v = reinterpret_cast<const float*>(x)[i];
      reinterpret_cast<float*>(y)[i] = keep ? v * scale : 0.f;
    } else {
      const float v = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(x)[i]);
      reinterpret_cast<unsigned short*>(y)[i] = keep ? f32_to_bf16_bits(v * scale) : (unsigned short)0;
    }
  }
}

// mean over T: grid (B, ceil(d/128)); a workgroup owns 128 columns of one sample: 16 lanes x 8 columns
// (one 256-byte segment per row), 16 row groups walk T, LDS combine.  B * d/128 workgroups (96 for
// B=16, d=768) instead of B * d/512.
__global__ __launch_bounds__(256)
void meanpool_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                         int T, int d, int ldy) {
  __shared__ float red[16][128 + 4];
  const int b = blockIdx.x, cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.y * 128 + cl * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < d) {
    const unsigned short* p = x + (size_t)b * T * d + col;
    for (int t = rg; t < T; t += 16) {
      const u32x4_t w = *reinterpret_cast<const u32x4_t*>(p + (size_t)t * d);
      s[0] += bf16lo(w[0]); s[1] += bf16hi(w[0]); s[2] += bf16lo(w[1]); s[3] += bf16hi(w[1]);
      s[4] += bf16lo(w[2]); s[5] += bf16hi(w[2]); s[6] += bf16lo(w[3]); s[7] += bf16hi(w[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rg][cl * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = blockIdx.y * 128 + threadIdx.x;
    if (c < d) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
      y[(size_t)b * ldy + c] = f32_to_bf16_bits(t / (float)T);
    }
  }
}

__global__ __launch_bounds__(EW_THREADS)
void meanpool_bwd_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx,
                         int B, int T, int d, int lddy) {
  const int dv = d >> 3;
  const int64_t nvec = (int64_t)B * T * dv;
  const float inv = 1.f / (float)T;
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
    const int c = (int)(i % dv);
    const int b = (int)(i / ((int64_t)T * dv));
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(dy + (size_t)b * lddy + c * 8);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16lo(w[e]) * inv, bf16hi(w[e]) * inv);
    *reinterpret_cast<u32x4_t*>(dx + i * 8) = o;
  }
}

// column sums (bias gradients), grouped: one launch covers several (x, out) problems.  A workgroup
// owns 64 rows x 512 columns: lane -> 8 columns, the 4 waves split the rows, LDS combine, one f32
// atomic per column per workgroup (M/64 adders per address, spread over the whole launch).
constexpr int COLSUM_ROWS = 64;
struct ColsumArgs {
  int nprob;
  int blk_start[MMF_COLSUM_MAX_PROBLEMS + 1];
  mmf_colsum_problem p[MMF_COLSUM_MAX_PROBLEMS];
};

__global__ __launch_bounds__(256)
void colsum_kernel(const ColsumArgs a) {
  __shared__ float red[3][512];
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_colsum_problem& P = a.p[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncc = (P.N + 511) / 512;
  const int t = (int)blockIdx.x - a.blk_start[pi];
  const int col = ((t % ncc) * 64 + lane) * 8;
  const int r0 = (t / ncc) * COLSUM_ROWS;
  const int r1 = min(P.M, r0 + COLSUM_ROWS);
  const unsigned short* __restrict__ x = static_cast<const unsigned short*>(P.x);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < P.N) {
    for (int r = r0 + wave; r < r1; r += 4) {
      const u32x4_t w = *reinterpret_cast<const u32x4_t*>(x + (size_t)r * P.ldx + col);
      s[0] += bf16lo(w[0]); s[1] += bf16hi(w[0]); s[2] += bf16lo(w[1]); s[3] += bf16hi(w[1]);
      s[4] += bf16lo(w[2]); s[5] += bf16hi(w[2]); s[6] += bf16lo(w[3]); s[7] += bf16hi(w[3]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave - 1][lane * 8 + e] = s[e];
  }
  __syncthreads();
  if (wave == 0 && col < P.N) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      atomicAdd(P.out + col + e, s[e] + red[0][lane * 8 + e] + red[1][lane * 8 + e] + red[2][lane * 8 + e]);
  }
}

// grouped mean over T + concatenation: problem i pools x_i (B, T_i, d) into columns [i d, (i+1) d) of y (B, ldy);
// one launch for the three modalities (reference models/fusion_layers.py:166-171) instead of three 96-workgroup ones
struct PoolArgs { const unsigned short* x[MMF_POOL_MAX]; unsigned short* g[MMF_POOL_MAX]; int T[MMF_POOL_MAX]; int blk_start[MMF_POOL_MAX + 1]; int n, B, d, ld; };
constexpr int POOL_THREADS = 1024, POOL_RG = POOL_THREADS / 16;   // 64 row groups x 16 column lanes: the 288 workgroups of
// a MulT launch are one per CU, so the rows in flight per CU are what sets the rate (256 threads: 1.8 TB/s)
__global__ __launch_bounds__(POOL_THREADS)
void meanpool_cat_fwd_kernel(const PoolArgs a, unsigned short* __restrict__ y) {
  __shared__ float red[POOL_RG][128 + 4];
  const int i = blockIdx.z, T = a.T[i], d = a.d;
  const int b = blockIdx.x, cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.y * 128 + cl * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < d) {
    const unsigned short* p = a.x[i] + (size_t)b * T * d + col;
    for (int t = rg; t < T; t += POOL_RG) {
      const u32x4_t w = *reinterpret_cast<const u32x4_t*>(p + (size_t)t * d);
      s[0] += bf16lo(w[0]); s[1] += bf16hi(w[0]); s[2] += bf16lo(w[1]); s[3] += bf16hi(w[1]);
      s[4] += bf16lo(w[2]); s[5] += bf16hi(w[2]); s[6] += bf16lo(w[3]); s[7] += bf16hi(w[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rg][cl * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = blockIdx.y * 128 + threadIdx.x;
    if (c < d) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < POOL_RG; ++g) t += red[g][threadIdx.x];
      y[(size_t)b * a.ld + i * d + c] = f32_to_bf16_bits(t / (float)T);
    }
  }
}
// backward: dx_i[b][t][:] = dy[b][i d : (i+1) d] / T_i
__global__ __launch_bounds__(EW_THREADS)
void meanpool_cat_bwd_kernel(const PoolArgs a, const unsigned short* __restrict__ dy) {
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.blk_start[i + 1]) ++i;
  const int T = a.T[i], d = a.d, dv = d >> 3;
  const int64_t nvec = (int64_t)a.B * T * dv;
  const float inv = 1.f / (float)T;
  const int nb = a.blk_start[i + 1] - a.blk_start[i];
  const int64_t stride = (int64_t)nb * EW_THREADS;
  unsigned short* dx = a.g[i];
  for (int64_t v = (int64_t)(blockIdx.x - a.blk_start[i]) * EW_THREADS + threadIdx.x; v < nvec; v += stride) {
    const int c = (int)(v % dv);
    const int b = (int)(v / ((int64_t)T * dv));
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(dy + (size_t)b * a.ld + i * d + c * 8);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16lo(w[e]) * inv, bf16hi(w[e]) * inv);
    *reinterpret_cast<u32x4_t*>(dx + v * 8) = o;
  }
}

// row-strided f32 (rows x cols, ld_src floats between rows) -> contiguous bf16: the column blocks autograd hands back
// from a torch.cat are cast without first being copied contiguous
__global__ __launch_bounds__(EW_THREADS)
void cast_f32_bf16_2d_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols, int ld_src) {
  const int cv = cols >> 2;
  const int64_t nvec = (int64_t)rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * EW_THREADS) {
    const int r = (int)(i / cv), c = (int)(i % cv) * 4;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src + (size_t)r * ld_src + c);
    *reinterpret_cast<u32x2_t*>(dst + (size_t)r * cols + c) = u32x2_t{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
  }
}

// zero up to MMF_ZERO_MAX_RANGES [start, end) float ranges of one buffer in ONE launch (the lazily-zeroed gradient
// arena's unmanaged holes: biases, LayerNorm vectors, torch-produced gradients)
struct ZeroArgs { float* base; int n; long long start[MMF_ZERO_MAX_RANGES]; long long end[MMF_ZERO_MAX_RANGES]; int blk_start[MMF_ZERO_MAX_RANGES + 1]; };
__global__ __launch_bounds__(EW_THREADS)
void zero_ranges_kernel(const ZeroArgs a) {
  int r = 0;
  while (r + 1 < a.n && (int)blockIdx.x >= a.blk_start[r + 1]) ++r;
  const int nb = a.blk_start[r + 1] - a.blk_start[r];
  const long long s0 = a.start[r], e0 = a.end[r];
  // 16-byte stores on the aligned middle, scalar stores on the edges
  const long long s4 = (s0 + 3) & ~3LL, e4 = e0 & ~3LL;
  float* p = a.base;
  const long long stride = (long long)nb * EW_THREADS * 4;
  for (long long i = s4 + ((long long)(blockIdx.x - a.blk_start[r]) * EW_THREADS + threadIdx.x) * 4; i + 4 <= e4; i += stride)
    *reinterpret_cast<f32x4_t*>(p + i) = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.x == a.blk_start[r]) {
    for (long long i = s0 + threadIdx.x; i < (s4 < e0 ? s4 : e0); i += EW_THREADS) p[i] = 0.f;
    for (long long i = (e4 > s4 ? e4 : s4) + threadIdx.x; i < e0; i += EW_THREADS) p[i] = 0.f;
  }
}

}  // namespace

#define EW_PTR_CHECK(name, cond) do { if (!(cond)) MMF_FAIL(MMF_E_ALIGN, name ": null or not 16-byte aligned pointer"); } while (0)

extern "C" int mmf_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (n <= 0) return MMF_OK;
  EW_PTR_CHECK("mmf_cast_f32_to_bf16", src && dst && mmf_aligned16(src) && mmf_aligned16(dst));
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n >> 4)), dim3(EW_THREADS), 0,
                     static_cast<hipStream_t>(stream), src, static_cast<unsigned short*>(dst), n);
  MMF_CHECK_LAUNCH("mmf_cast_f32_to_bf16");
  return MMF_OK;
}

extern "C" int mmf_cast_bf16_to_f32_scaled(const void* src, float* dst, int64_t n, float scale, void* stream) {
  if (n <= 0) return MMF_OK;
  EW_PTR_CHECK("mmf_cast_bf16_to_f32", src && dst && mmf_aligned16(src) && mmf_aligned16(dst));
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n >> 4)), dim3(EW_THREADS), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(src), dst, n, scale);
  MMF_CHECK_LAUNCH("mmf_cast_bf16_to_f32");
  return MMF_OK;
}
extern "C" int mmf_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  return mmf_cast_bf16_to_f32_scaled(src, dst, n, 1.f, stream);
}

extern "C" int mmf_add3_bf16(const void* a, const void* b, const void* c, void* y, int64_t n, void* stream) {
  if (n <= 0) return MMF_OK;
  EW_PTR_CHECK("mmf_add3_bf16", a && b && y && mmf_aligned16(a) && mmf_aligned16(b) && mmf_aligned16(y) &&
               (!c || mmf_aligned16(c)));
  hipLaunchKernelGGL(add3_kernel, dim3(ew_grid(n >> 3)), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(a), static_cast<const unsigned short*>(b),
                     static_cast<const unsigned short*>(c), static_cast<unsigned short*>(y), n);
  MMF_CHECK_LAUNCH("mmf_add3_bf16");
  return MMF_OK;
}

extern "C" int mmf_add3_grouped(const mmf_add3_problem* problems, int num_problems, void* stream) {
  if (!problems || num_problems <= 0 || num_problems > MMF_ADD3_MAX)
    MMF_FAIL(MMF_E_SHAPE, "mmf_add3_grouped: num_problems=%d not in 1..%d", num_problems, MMF_ADD3_MAX);
  Add3Args a;
  a.n = num_problems;
  int total = 0;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_add3_problem& q = problems[i];
    if (q.n <= 0) MMF_FAIL(MMF_E_SHAPE, "mmf_add3_grouped[%d]: n=%lld", i, (long long)q.n);
    EW_PTR_CHECK("mmf_add3_grouped", q.a && q.b && q.c && q.y && mmf_aligned16(q.a) && mmf_aligned16(q.b) &&
                 mmf_aligned16(q.c) && mmf_aligned16(q.y));
    a.blk_start[i] = total;
    total += ew_grid(q.n >> 3);
    a.p[i] = q;
  }
  a.blk_start[num_problems] = total;
  hipLaunchKernelGGL(add3_grouped_kernel, dim3(total), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_add3_grouped");
  return MMF_OK;
}

// y = sum of n (2..MMF_ADDN_MAX) bf16 tensors, f32 accumulate, ONE pass: the gradient of a tensor used n times in
// the forward (MulT's input rows feed two q-projections, two k/v-projections, two residuals and the three-way sum,
// reference models/fusion_layers.py:146-158) instead of n - 1 pairwise adds.
struct AddNArgs { int n; const unsigned short* x[MMF_ADDN_MAX]; };
template <bool OUT_F32>
__global__ __launch_bounds__(EW_THREADS)
void addn_kernel(const AddNArgs a, void* __restrict__ y, int64_t numel) {
  const int64_t nvec = numel >> 3;
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < MMF_ADDN_MAX; ++k) {
      if (k < a.n) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(a.x[k] + i * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += bf16lo(v[e]); acc[2 * e + 1] += bf16hi(v[e]); }
      }
    }
    if (OUT_F32) {
      float* o = static_cast<float*>(y) + i * 8;
      *reinterpret_cast<f32x4_t*>(o) = f32x4_t{acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4_t*>(o + 4) = f32x4_t{acc[4], acc[5], acc[6], acc[7]};
    } else {
      const u32x4_t o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
      *reinterpret_cast<u32x4_t*>(static_cast<unsigned short*>(y) + i * 8) = o;
    }
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 3) + threadIdx.x;
    if (t < numel) {
      float s = 0.f;
      for (int k = 0; k < a.n; ++k) s += bf16_bits_to_f32(a.x[k][t]);
      if (OUT_F32) static_cast<float*>(y)[t] = s; else static_cast<unsigned short*>(y)[t] = f32_to_bf16_bits(s);
    }
  }
}

extern "C" int mmf_addn_bf16(const void* const* xs, int n, void* y, int64_t numel, int out_f32, void* stream) {
  if (!xs || n < 2 || n > MMF_ADDN_MAX || !y || numel <= 0) MMF_FAIL(MMF_E_SHAPE, "mmf_addn_bf16: n=%d (2..%d) numel=%lld", n, MMF_ADDN_MAX, (long long)numel);
  AddNArgs a; a.n = n;
  for (int k = 0; k < MMF_ADDN_MAX; ++k) a.x[k] = nullptr;
  for (int k = 0; k < n; ++k) {
    EW_PTR_CHECK("mmf_addn_bf16", xs[k] && mmf_aligned16(xs[k]));
    a.x[k] = static_cast<const unsigned short*>(xs[k]);
  }
  EW_PTR_CHECK("mmf_addn_bf16", mmf_aligned16(y));
  const int grid = ew_grid(numel >> 3);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (out_f32) hipLaunchKernelGGL(addn_kernel<true>, dim3(grid), dim3(EW_THREADS), 0, s, a, y, numel);
  else         hipLaunchKernelGGL(addn_kernel<false>, dim3(grid), dim3(EW_THREADS), 0, s, a, y, numel);
  MMF_CHECK_LAUNCH("mmf_addn_bf16");
  return MMF_OK;
}

struct AddNGroupArgs { int nprob; int blk_start[MMF_ADDN_GROUP_MAX + 1]; mmf_addn_problem p[MMF_ADDN_GROUP_MAX]; };
__global__ __launch_bounds__(EW_THREADS)
void addn_grouped_kernel(const AddNGroupArgs a) {
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_addn_problem& P = a.p[pi];
  const int64_t nvec = P.numel >> 3;
  const int blk = (int)blockIdx.x - a.blk_start[pi], nblk = a.blk_start[pi + 1] - a.blk_start[pi];
  const int64_t stride = (int64_t)nblk * EW_THREADS;
  unsigned short* y = static_cast<unsigned short*>(P.y);
  for (int64_t i = (int64_t)blk * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < MMF_ADDN_MAX; ++k) {
      if (k < P.n) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(static_cast<const unsigned short*>(P.x[k]) + i * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += bf16lo(v[e]); acc[2 * e + 1] += bf16hi(v[e]); }
      }
    }
    const u32x4_t o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
    *reinterpret_cast<u32x4_t*>(y + i * 8) = o;
  }
  if (blk == 0) {
    const int64_t t = (nvec << 3) + threadIdx.x;
    if (t < P.numel) {
      float s = 0.f;
      for (int k = 0; k < P.n; ++k) s += bf16_bits_to_f32(static_cast<const unsigned short*>(P.x[k])[t]);
      y[t] = f32_to_bf16_bits(s);
    }
  }
}

extern "C" int mmf_addn_grouped(const mmf_addn_problem* problems, int num_problems, void* stream) {
  if (!problems || num_problems <= 0 || num_problems > MMF_ADDN_GROUP_MAX)
    MMF_FAIL(MMF_E_SHAPE, "mmf_addn_grouped: num_problems=%d not in 1..%d", num_problems, MMF_ADDN_GROUP_MAX);
  AddNGroupArgs a; a.nprob = num_problems;
  int total = 0;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_addn_problem& q = problems[i];
    if (q.n < 2 || q.n > MMF_ADDN_MAX || q.numel <= 0) MMF_FAIL(MMF_E_SHAPE, "mmf_addn_grouped[%d]: n=%d (2..%d) numel=%lld", i, q.n, MMF_ADDN_MAX, (long long)q.numel);
    for (int k = 0; k < q.n; ++k) EW_PTR_CHECK("mmf_addn_grouped", q.x[k] && mmf_aligned16(q.x[k]));
    EW_PTR_CHECK("mmf_addn_grouped", q.y && mmf_aligned16(q.y));
    a.blk_start[i] = total;
    total += ew_grid(q.numel >> 3);
    a.p[i] = q;
  }
  a.blk_start[num_problems] = total;
  hipLaunchKernelGGL(addn_grouped_kernel, dim3(total), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_addn_grouped");
  return MMF_OK;
}

extern "C" int mmf_dropout(const void* x, void* y, int64_t n, int is_f32, float p, const uint64_t* rng_state,
                           uint32_t site, void* stream) {
  if (n <= 0) return MMF_OK;
  if (!x || !y || !rng_state || !(p >= 0.f) || p >= 1.f) MMF_FAIL(MMF_E_SHAPE, "mmf_dropout: null pointer or p outside [0,1)");
  const unsigned thresh = mmf_drop_thresh(p);
  const float scale = 1.f / (1.f - (float)thresh * (1.f / 4294967296.f));
  const unsigned long long* st = reinterpret_cast<const unsigned long long*>(rng_state);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (is_f32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, s,
                                 static_cast<const float*>(x), static_cast<float*>(y), n, thresh, scale, st, site);
  else hipLaunchKernelGGL(dropout_kernel<unsigned short>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, s,
                          static_cast<const unsigned short*>(x), static_cast<unsigned short*>(y), n, thresh, scale, st, site);
  MMF_CHECK_LAUNCH("mmf_dropout");
  return MMF_OK;
}

// dx (bf16) = dy * (y > 0) with dy and / or y in f32: the gradient of an f32-output linear with ReLU narrowed and
// masked in one pass (was: cast dy, cast y, mask = three launches of a (B, d)-row chain where a launch is ~5 us)
template <bool DY_F32, bool Y_F32>
__global__ __launch_bounds__(EW_THREADS)
void relu_bwd_mixed_kernel(const void* __restrict__ dy_, const void* __restrict__ y_, unsigned short* __restrict__ dx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
    const float g = DY_F32 ? static_cast<const float*>(dy_)[i]
                           : __uint_as_float((unsigned)static_cast<const unsigned short*>(dy_)[i] << 16);
    const float v = Y_F32 ? static_cast<const float*>(y_)[i]
                          : __uint_as_float((unsigned)static_cast<const unsigned short*>(y_)[i] << 16);
    dx[i] = v > 0.f ? f32_to_bf16_bits(g) : (unsigned short)0;
  }
}

extern "C" int mmf_relu_bwd_mixed(const void* dy, int dy_f32, const void* y, int y_f32, void* dx, int64_t n, void* stream) {
  if (n <= 0) return MMF_OK;
  if (!dy || !y || !dx) MMF_FAIL(MMF_E_SHAPE, "mmf_relu_bwd_mixed: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned short* o = static_cast<unsigned short*>(dx);
  const dim3 grid(ew_grid(n)), block(EW_THREADS);
  if (dy_f32 && y_f32)       hipLaunchKernelGGL((relu_bwd_mixed_kernel<true, true>), grid, block, 0, s, dy, y, o, n);
  else if (dy_f32)           hipLaunchKernelGGL((relu_bwd_mixed_kernel<true, false>), grid, block, 0, s, dy, y, o, n);
  else if (y_f32)            hipLaunchKernelGGL((relu_bwd_mixed_kernel<false, true>), grid, block, 0, s, dy, y, o, n);
  else                       hipLaunchKernelGGL((relu_bwd_mixed_kernel<false, false>), grid, block, 0, s, dy, y, o, n);
  MMF_CHECK_LAUNCH("mmf_relu_bwd_mixed");
  return MMF_OK;
}

extern "C" int mmf_relu_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, void* stream) {
  if (n <= 0) return MMF_OK;
  if (!dy || !y || !dx) MMF_FAIL(MMF_E_SHAPE, "mmf_relu_bwd_bf16: null pointer");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(dy), static_cast<const unsigned short*>(y),
                     static_cast<unsigned short*>(dx), n);
  MMF_CHECK_LAUNCH("mmf_relu_bwd_bf16");
  return MMF_OK;
}

extern "C" int mmf_meanpool_fwd(const void* x, void* y, int B, int T, int d, int ldy, void* stream) {
  if (B <= 0 || T <= 0 || d <= 0 || (d & 7) || (ldy & 7) || ldy < d)
    MMF_FAIL(MMF_E_SHAPE, "mmf_meanpool_fwd: B=%d T=%d d=%d ldy=%d (d, ldy multiples of 8)", B, T, d, ldy);
  EW_PTR_CHECK("mmf_meanpool_fwd", x && y && mmf_aligned16(x) && mmf_aligned16(y));
  hipLaunchKernelGGL(meanpool_fwd_kernel, dim3(B, (d + 127) / 128), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x), static_cast<unsigned short*>(y), T, d, ldy);
  MMF_CHECK_LAUNCH("mmf_meanpool_fwd");
  return MMF_OK;
}

extern "C" int mmf_meanpool_bwd(const void* dy, void* dx, int B, int T, int d, int lddy, void* stream) {
  if (B <= 0 || T <= 0 || d <= 0 || (d & 7) || (lddy & 7) || lddy < d)
    MMF_FAIL(MMF_E_SHAPE, "mmf_meanpool_bwd: B=%d T=%d d=%d lddy=%d (d, lddy multiples of 8)", B, T, d, lddy);
  EW_PTR_CHECK("mmf_meanpool_bwd", dy && dx && mmf_aligned16(dy) && mmf_aligned16(dx));
  hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(ew_grid((int64_t)B * T * (d >> 3))), dim3(EW_THREADS), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(dy),
                     static_cast<unsigned short*>(dx), B, T, d, lddy);
  MMF_CHECK_LAUNCH("mmf_meanpool_bwd");
  return MMF_OK;
}

extern "C" int mmf_colsum_grouped(const mmf_colsum_problem* problems, int num_problems, void* stream) {
  if (!problems || num_problems <= 0 || num_problems > MMF_COLSUM_MAX_PROBLEMS)
    MMF_FAIL(MMF_E_SHAPE, "mmf_colsum_grouped: num_problems=%d out of range", num_problems);
  ColsumArgs a; a.nprob = num_problems;
  int total = 0;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_colsum_problem& p = problems[i];
    if (p.M <= 0 || p.N <= 0 || (p.N & 7) || (p.ldx & 7) || p.ldx < p.N)
      MMF_FAIL(MMF_E_SHAPE, "mmf_colsum_grouped[%d]: M=%d N=%d ldx=%d (N, ldx multiples of 8)", i, p.M, p.N, p.ldx);
    if (!p.x || !p.out || !mmf_aligned16(p.x)) MMF_FAIL(MMF_E_ALIGN, "mmf_colsum_grouped[%d]: null or unaligned pointer", i);
    a.blk_start[i] = total;
    total += ((p.N + 511) / 512) * ((p.M + COLSUM_ROWS - 1) / COLSUM_ROWS);
    a.p[i] = p;
  }
  a.blk_start[num_problems] = total;
  hipLaunchKernelGGL(colsum_kernel, dim3(total), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_colsum_grouped");
  return MMF_OK;
}

extern "C" int mmf_colsum_bf16(const void* x, float* out, int M, int N, int ldx, void* stream) {
  mmf_colsum_problem p;
  p.x = x; p.out = out; p.M = M; p.N = N; p.ldx = ldx;
  return mmf_colsum_grouped(&p, 1, stream);
}

extern "C" int mmf_zero_ranges_f32(float* base, const int64_t* starts, const int64_t* ends, int n, void* stream) {
  if (n <= 0) return MMF_OK;
  if (!base || !starts || !ends || n > MMF_ZERO_MAX_RANGES || (reinterpret_cast<uintptr_t>(base) & 15))
    MMF_FAIL(MMF_E_SHAPE, "mmf_zero_ranges_f32: null / misaligned base or n=%d outside [1,%d]", n, MMF_ZERO_MAX_RANGES);
  ZeroArgs a;
  a.base = base; a.n = n;
  int total = 0;
  for (int r = 0; r < n; ++r) {
    if (ends[r] < starts[r] || starts[r] < 0) MMF_FAIL(MMF_E_SHAPE, "mmf_zero_ranges_f32: bad range %d", r);
    a.start[r] = starts[r]; a.end[r] = ends[r];
    a.blk_start[r] = total;
    long long nb = ((ends[r] - starts[r]) / 4 + EW_THREADS - 1) / EW_THREADS;
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    total += (int)nb;
  }
  a.blk_start[n] = total;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(total), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_zero_ranges_f32");
  return MMF_OK;
}

extern "C" int mmf_meanpool_cat_fwd(const void* const* xs, const int* Ts, int n, void* y, int B, int d, int ldy, void* stream) {
  if (!xs || !Ts || !y || n <= 0 || n > MMF_POOL_MAX || B <= 0 || d <= 0 || (d & 7) || (ldy & 7) || ldy < n * d)
    MMF_FAIL(MMF_E_SHAPE, "mmf_meanpool_cat_fwd: n=%d B=%d d=%d ldy=%d", n, B, d, ldy);
  PoolArgs a = {};
  a.n = n; a.B = B; a.d = d; a.ld = ldy;
  for (int i = 0; i < n; ++i) {
    if (!xs[i] || Ts[i] <= 0 || !mmf_aligned16(xs[i])) MMF_FAIL(MMF_E_SHAPE, "mmf_meanpool_cat_fwd[%d]: bad operand", i);
    a.x[i] = static_cast<const unsigned short*>(xs[i]); a.T[i] = Ts[i];
  }
  hipLaunchKernelGGL(meanpool_cat_fwd_kernel, dim3(B, (d + 127) / 128, n), dim3(POOL_THREADS), 0, static_cast<hipStream_t>(stream), a,
                     static_cast<unsigned short*>(y));
  MMF_CHECK_LAUNCH("mmf_meanpool_cat_fwd");
  return MMF_OK;
}

extern "C" int mmf_meanpool_cat_bwd(const void* dy, void* const* dxs, const int* Ts, int n, int B, int d, int lddy, void* stream) {
  if (!dy || !dxs || !Ts || n <= 0 || n > MMF_POOL_MAX || B <= 0 || d <= 0 || (d & 7) || (lddy & 7) || lddy < n * d)
    MMF_FAIL(MMF_E_SHAPE, "mmf_meanpool_cat_bwd: n=%d B=%d d=%d lddy=%d", n, B, d, lddy);
  PoolArgs a = {};
  a.n = n; a.B = B; a.d = d; a.ld = lddy;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    if (!dxs[i] || Ts[i] <= 0 || !mmf_aligned16(dxs[i])) MMF_FAIL(MMF_E_SHAPE, "mmf_meanpool_cat_bwd[%d]: bad operand", i);
    a.g[i] = static_cast<unsigned short*>(dxs[i]); a.T[i] = Ts[i];
    a.blk_start[i] = total;
    total += ew_grid((int64_t)B * Ts[i] * (d >> 3));
  }
  a.blk_start[n] = total;
  hipLaunchKernelGGL(meanpool_cat_bwd_kernel, dim3(total), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), a,
                     static_cast<const unsigned short*>(dy));
  MMF_CHECK_LAUNCH("mmf_meanpool_cat_bwd");
  return MMF_OK;
}

extern "C" int mmf_cast_f32_to_bf16_2d(const float* src, void* dst, int rows, int cols, int ld_src, void* stream) {
  if (rows <= 0 || cols <= 0) return MMF_OK;
  if (!src || !dst || (cols & 3) || (ld_src & 3) || ld_src < cols || !mmf_aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7))
    MMF_FAIL(MMF_E_ALIGN, "mmf_cast_f32_to_bf16_2d: rows=%d cols=%d ld=%d (cols, ld multiples of 4; 16-byte aligned source)", rows, cols, ld_src);
  hipLaunchKernelGGL(cast_f32_bf16_2d_kernel, dim3(ew_grid((int64_t)rows * (cols >> 2) / 4 + 1)), dim3(EW_THREADS), 0,
                     static_cast<hipStream_t>(stream), src, static_cast<unsigned short*>(dst), rows, cols, ld_src);
  MMF_CHECK_LAUNCH("mmf_cast_f32_to_bf16_2d");
  return MMF_OK;
}
