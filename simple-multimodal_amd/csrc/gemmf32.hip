// fp32-storage parity mode (north_star: "within 1e-3 fp32"): grouped / batched GEMM on the EXACT f32 MFMA
// (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fmaf chain, no reduced-precision path — cdna_hip_programming.md
// "FP32-input MFMA"), the row softmax pair that turns it into the reference's explicit-scores attention
// (torch F.multi_head_attention_forward, need_weights path: models/fusion_layers.py:204 via nn.MultiheadAttention),
// and an fp32 LayerNorm.  Same epilogue contract as the bf16 grouped GEMM (include/mmfusion.h), every operand f32.
// This mode exists to hold tight tolerances against the fp32 oracle; speed is secondary (fp32 MFMA peak is 157 TF,
// 1/16 of bf16): 128 x 128 x 32 block, 4 waves as 2 x 2, each 2 x 2 tiles of 32 x 32, operands staged through
// registers into a [k][row] LDS image (the MFMA's A / B lane maps read it conflict-free), single buffer.
#include "mmf_internal.h"

namespace {

constexpr int F_BM = 128, F_BN = 128, F_BK = 32, F_THREADS = 256, F_LD = F_BM + 4;

struct GemmF32Args {
  int nprob, epi;
  float alpha;
  int nb0, nb1;                                   // batch extents (1, 1 for the grouped form)
  long long sA[2], sB[2], sC[2];                  // element strides of the two batch axes
  int tile_start[MMF_GEMM_MAX_PROBLEMS + 1];
  mmf_gemm_problem p[MMF_GEMM_MAX_PROBLEMS];
};

// stage a 128(row) x 32(k) block of an operand into S[k][row].  KR: the reduction index is the memory ROW
// (element (row, k) at X[k * ld + row]); otherwise k is contiguous (X[row * ld + k]).
template <bool KR>
__device__ __forceinline__ void stage(float (*S)[F_LD], const float* __restrict__ X, int ld, int r0, int rows, int k0, int K,
                                      int tid, float v[16]) {
  if (KR) {
    const int r = tid & 127, kb = tid >> 7;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = kb + 2 * i;
      v[i] = (r0 + r < rows && k0 + k < K) ? X[(size_t)(k0 + k) * ld + r0 + r] : 0.f;
    }
  } else {
    const int k = tid & 31, rb = tid >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = rb + 8 * i;
      v[i] = (r0 + r < rows && k0 + k < K) ? X[(size_t)(r0 + r) * ld + k0 + k] : 0.f;
    }
  }
}
template <bool KR>
__device__ __forceinline__ void commit(float (*S)[F_LD], int tid, const float v[16]) {
  if (KR) {
    const int r = tid & 127, kb = tid >> 7;
#pragma unroll
    for (int i = 0; i < 16; ++i) S[kb + 2 * i][r] = v[i];
  } else {
    const int k = tid & 31, rb = tid >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) S[k][rb + 8 * i] = v[i];
  }
}

template <bool A_KR, bool B_KR>
__global__ __launch_bounds__(F_THREADS)
void gemm_f32_kernel(const GemmF32Args a) {
  __shared__ float As[F_BK][F_LD];
  __shared__ float Bs[F_BK][F_LD];
  const int tiles_total = a.tile_start[a.nprob];
  const int bz = blockIdx.x / tiles_total, t = blockIdx.x % tiles_total;
  int pi = 0;
  while (pi + 1 < a.nprob && t >= a.tile_start[pi + 1]) ++pi;
  const mmf_gemm_problem& P = a.p[pi];
  const int M = P.M, N = P.N, K = P.K;
  const int tn = (N + F_BN - 1) / F_BN;
  const int loc = t - a.tile_start[pi];
  const int m0 = (loc / tn) * F_BM, n0 = (loc % tn) * F_BN;
  const int i0 = bz / a.nb1, i1 = bz % a.nb1;
  const float* A = static_cast<const float*>(P.A) + i0 * a.sA[0] + i1 * a.sA[1];
  const float* Bm = static_cast<const float*>(P.B) + i0 * a.sB[0] + i1 * a.sB[1];
  float* C = static_cast<float*>(P.C) + i0 * a.sC[0] + i1 * a.sC[1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lk = lane >> 5;
  f32x16_t acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  const bool do_colsum = A_KR && (a.epi & MMF_EPI_COLSUM_A) && n0 == 0;
  float csum = 0.f;

  float va[16], vb[16];
  for (int k0 = 0; k0 < K; k0 += F_BK) {
    stage<A_KR>(As, A, P.lda, m0, M, k0, K, tid, va);
    stage<B_KR>(Bs, Bm, P.ldb, n0, N, k0, K, tid, vb);
    __syncthreads();                                   // the previous step's fragment reads are done
    commit<A_KR>(As, tid, va);
    commit<B_KR>(Bs, tid, vb);
    __syncthreads();
    if (do_colsum && tid < F_BM) {
#pragma unroll
      for (int k = 0; k < F_BK; ++k) csum += As[k][tid];
    }
#pragma unroll
    for (int kk = 0; kk < F_BK; kk += 2) {
      float af[2], bf[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) af[x] = As[kk + lk][wm * 64 + x * 32 + li];
#pragma unroll
      for (int y = 0; y < 2; ++y) bf[y] = Bs[kk + lk][wn * 64 + y * 32 + li];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[x], bf[y], acc[x][y], 0, 0, 0);
    }
  }
  if (do_colsum && tid < F_BM && m0 + tid < M) atomicAdd(const_cast<float*>(P.bias) + m0 + tid, csum);

  // epilogue: D[i = A row (m)][j = B row (n)], n on the lanes
  const float* aux = static_cast<const float*>(P.aux);
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int n = n0 + wn * 64 + y * 32 + li;
      if (n >= N) continue;
      const float bias = (a.epi & MMF_EPI_BIAS) ? P.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= M) continue;
        float v = acc[x][y][r] + bias;
        if (a.epi & MMF_EPI_RELU) v = fmaxf(v, 0.f);
        if (a.epi & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) {
          const float ax = aux[(size_t)m * P.ldaux + n];
          if (a.epi & MMF_EPI_MASK_AUX) v = ax > 0.f ? v : 0.f;
          v *= a.alpha;
          if (a.epi & MMF_EPI_ADD_AUX) v += ax;
        } else {
          v *= a.alpha;
        }
        float* c = C + (size_t)m * P.ldc + n;
        if (a.epi & MMF_EPI_ACCUM) v += *c;
        *c = v;
      }
    }
}

// ---- row softmax pair (explicit-scores attention) ----------------------------------------------------------
// in place: S[row][0..cols) <- softmax(scale * S[row][:]); one wave per row
__global__ __launch_bounds__(256)
void softmax_rows_kernel(float* __restrict__ S, long long rows, int cols, float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* s = S + row * cols;
  float mx = -3.0e38f;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, s[c] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) { const float e = expf(s[c] * scale - mx); s[c] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane; c < cols; c += 64) s[c] *= inv;
}
// in place: dP[row][:] <- scale * P[row][:] * (dP[row][:] - sum_c dP[row][c] P[row][c])
__global__ __launch_bounds__(256)
void softmax_bwd_rows_kernel(const float* __restrict__ Pm, float* __restrict__ dP, long long rows, int cols, float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = Pm + row * cols;
  float* d = dP + row * cols;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += d[c] * p[c];
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) d[c] = scale * p[c] * (d[c] - dot);
}

// ---- fp32 LayerNorm (one wave per row) ----------------------------------------------------------------------
__global__ __launch_bounds__(256)
void ln_f32_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ g,
                       const float* __restrict__ b, float* __restrict__ mean, float* __restrict__ rstd, int rows, int d, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + (size_t)row * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c];
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) { const float t = xr[c] - mu; q += t * t; }
  const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
  for (int c = lane; c < d; c += 64) y[(size_t)row * d + c] = (xr[c] - mu) * rs * g[c] + b[c];
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
// dx per row; dgamma / dbeta by one atomic add per (row block of 64 rows, column): a workgroup owns 64 rows
__global__ __launch_bounds__(256)
void ln_f32_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ g,
                       const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                       float* __restrict__ dg, float* __restrict__ db, int rows, int d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = blockIdx.x * 64;
  for (int rr = wave; rr < 64; rr += 4) {
    const int row = r0 + rr;
    if (row >= rows) break;
    const float* xr = x + (size_t)row * d;
    const float* dr = dy + (size_t)row * d;
    const float mu = mean[row], rs = rstd[row];
    float c1 = 0.f, c2 = 0.f;
    for (int c = lane; c < d; c += 64) { const float gy = dr[c] * g[c]; c1 += gy; c2 += gy * (xr[c] - mu) * rs; }
    c1 = wave_sum(c1) / (float)d; c2 = wave_sum(c2) / (float)d;
    for (int c = lane; c < d; c += 64) dx[(size_t)row * d + c] = rs * (dr[c] * g[c] - c1 - (xr[c] - mu) * rs * c2);
  }
  for (int c = threadIdx.x; c < d; c += 256) {
    float sg = 0.f, sb = 0.f;
    for (int rr = 0; rr < 64 && r0 + rr < rows; ++rr) {
      const int row = r0 + rr;
      const float dv = dy[(size_t)row * d + c];
      sg += dv * (x[(size_t)row * d + c] - mean[row]) * rstd[row];
      sb += dv;
    }
    atomicAdd(dg + c, sg);
    atomicAdd(db + c, sb);
  }
}

int check_problem(const char* who, const mmf_gemm_problem& q, int i, int layout, int epi) {
  if (!q.A || !q.B || !q.C || q.M <= 0 || q.N <= 0 || q.K <= 0) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: M=%d N=%d K=%d or null operand", who, i, q.M, q.N, q.K);
  if ((epi & MMF_EPI_BIAS) && (epi & MMF_EPI_COLSUM_A)) MMF_FAIL(MMF_E_SHAPE, "%s: BIAS and COLSUM_A are exclusive", who);
  if ((epi & (MMF_EPI_BIAS | MMF_EPI_COLSUM_A)) && !q.bias) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: bias missing", who, i);
  if ((epi & MMF_EPI_COLSUM_A) && layout != MMF_GEMM_TN) MMF_FAIL(MMF_E_SHAPE, "%s: COLSUM_A needs the TN layout", who);
  if ((epi & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) && !q.aux) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: aux missing", who, i);
  if (epi & MMF_EPI_DROPOUT) MMF_FAIL(MMF_E_UNSUPPORTED, "%s: the fp32 parity mode has no dropout epilogue", who);
  return MMF_OK;
}

int launch_f32(const char* who, const mmf_gemm_problem* problems, int n, int layout, int epi, float alpha, int nb0, int nb1,
               const int64_t* sA, const int64_t* sB, const int64_t* sC, hipStream_t s) {
  if (!problems || n <= 0 || n > MMF_GEMM_MAX_PROBLEMS) MMF_FAIL(MMF_E_SHAPE, "%s: num_problems=%d out of range", who, n);
  if (layout < 0 || layout > 2) MMF_FAIL(MMF_E_SHAPE, "%s: layout %d", who, layout);
  GemmF32Args a; a.nprob = n; a.epi = epi; a.alpha = alpha; a.nb0 = nb0; a.nb1 = nb1;
  for (int k = 0; k < 2; ++k) { a.sA[k] = sA ? sA[k] : 0; a.sB[k] = sB ? sB[k] : 0; a.sC[k] = sC ? sC[k] : 0; }
  int total = 0;
  for (int i = 0; i < n; ++i) {
    if (int rc = check_problem(who, problems[i], i, layout, epi)) return rc;
    a.tile_start[i] = total;
    total += ((problems[i].M + F_BM - 1) / F_BM) * ((problems[i].N + F_BN - 1) / F_BN);
    a.p[i] = problems[i];
  }
  a.tile_start[n] = total;
  const long long grid = (long long)total * nb0 * nb1;
  if (grid <= 0 || grid > 0x7fffffffLL) MMF_FAIL(MMF_E_SHAPE, "%s: grid of %lld workgroups", who, grid);
  const dim3 g((unsigned)grid), b(F_THREADS);
  switch (layout) {
    case MMF_GEMM_NT: hipLaunchKernelGGL((gemm_f32_kernel<false, false>), g, b, 0, s, a); break;
    case MMF_GEMM_NN: hipLaunchKernelGGL((gemm_f32_kernel<false, true>), g, b, 0, s, a); break;
    default:          hipLaunchKernelGGL((gemm_f32_kernel<true, true>), g, b, 0, s, a); break;
  }
  MMF_CHECK_LAUNCH(who);
  return MMF_OK;
}

}  // namespace

extern "C" int mmf_gemm_f32_grouped(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                                    float alpha, void* stream) {
  return launch_f32("mmf_gemm_f32_grouped", problems, num_problems, layout, epilogue, alpha, 1, 1, nullptr, nullptr, nullptr,
                    static_cast<hipStream_t>(stream));
}

extern "C" int mmf_gemm_f32_batched(const mmf_gemm_problem* problem, int layout, int epilogue, float alpha, int nb0, int nb1,
                                    const int64_t strideA[2], const int64_t strideB[2], const int64_t strideC[2], void* stream) {
  if (nb0 <= 0 || nb1 <= 0 || !strideA || !strideB || !strideC) MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_f32_batched: batch %d x %d", nb0, nb1);
  if (epilogue & (MMF_EPI_COLSUM_A | MMF_EPI_BIAS | MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX))
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_f32_batched: only alpha / ACCUM epilogues (bias and aux are not batch-strided)");
  return launch_f32("mmf_gemm_f32_batched", problem, 1, layout, epilogue, alpha, nb0, nb1, strideA, strideB, strideC,
                    static_cast<hipStream_t>(stream));
}

extern "C" int mmf_softmax_rows_f32(float* S, int64_t rows, int cols, float scale, void* stream) {
  if (!S || rows <= 0 || cols <= 0 || (rows + 3) / 4 > 0x7fffffffLL) MMF_FAIL(MMF_E_SHAPE, "mmf_softmax_rows_f32: rows=%lld cols=%d", (long long)rows, cols);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), S,
                     (long long)rows, cols, scale);
  MMF_CHECK_LAUNCH("mmf_softmax_rows_f32");
  return MMF_OK;
}

extern "C" int mmf_softmax_bwd_rows_f32(const float* P, float* dP, int64_t rows, int cols, float scale, void* stream) {
  if (!P || !dP || rows <= 0 || cols <= 0 || (rows + 3) / 4 > 0x7fffffffLL) MMF_FAIL(MMF_E_SHAPE, "mmf_softmax_bwd_rows_f32: rows=%lld cols=%d", (long long)rows, cols);
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), P, dP,
                     (long long)rows, cols, scale);
  MMF_CHECK_LAUNCH("mmf_softmax_bwd_rows_f32");
  return MMF_OK;
}

extern "C" int mmf_layernorm_f32_fwd(const float* x, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                                     int rows, int d, float eps, void* stream) {
  if (!x || !y || !gamma || !beta || !mean || !rstd || rows <= 0 || d <= 0) MMF_FAIL(MMF_E_SHAPE, "mmf_layernorm_f32_fwd: rows=%d d=%d", rows, d);
  hipLaunchKernelGGL(ln_f32_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, gamma, beta, mean,
                     rstd, rows, d, eps);
  MMF_CHECK_LAUNCH("mmf_layernorm_f32_fwd");
  return MMF_OK;
}

extern "C" int mmf_layernorm_f32_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd,
                                     float* dx, float* dgamma, float* dbeta, int rows, int d, void* stream) {
  if (!x || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0 || d <= 0)
    MMF_FAIL(MMF_E_SHAPE, "mmf_layernorm_f32_bwd: rows=%d d=%d", rows, d);
  hipLaunchKernelGGL(ln_f32_bwd_kernel, dim3((rows + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), x, dy, gamma, mean, rstd,
                     dx, dgamma, dbeta, rows, d);
  MMF_CHECK_LAUNCH("mmf_layernorm_f32_bwd");
  return MMF_OK;
}
