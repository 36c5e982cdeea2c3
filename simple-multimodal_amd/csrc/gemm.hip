// Grouped bf16 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), f32 accumulate.
//
// One launch = up to MMF_GEMM_MAX_PROBLEMS independent problems (the six cross-modal blocks'
// projections / FFNs of MulT, models/fusion_layers.py:146-153, are issued together so the grid
// has >> 256 workgroups even though each problem is small).  The problem table travels by value
// in the kernel arguments, so a launch is self-contained and hipGraph-capturable.
//
// Tiling: 128(m) x 128(n) x 64(k) per 256-thread workgroup, 4 waves as 2(m) x 2(n), each wave a
// 64 x 64 block = 4 x 4 MFMA tiles of 16 x 16.  The n index rides the MFMA *row* (register) axis
// and m the *column* (lane) axis, so each lane ends up with 4 consecutive n of one output row and
// the epilogue stores 8 B (bf16) / 16 B (f32) per lane.
//
// Operand tiles live in LDS in one of two images, chosen by how the operand lies in HBM:
//   KC ("k contiguous", [idx][k], 128-B rows):  x[m][k], W[n][k].  Read with ds_read_b128;
//       16-B chunk index XOR (row & 7) spreads a fragment read over the banks (guide T2).
//   KR ("k is the row", [k][idx], 256-B rows):  W[k][n] in dgrad, dy[k][m] / x[k][n] in wgrad.
//       Read with ds_read_b64_tr_b16 (hardware transpose); chunk index XOR
//       ((k&3)<<2 | (k>>2)&3) is image (b) of guide T10 (conflict-free for these reads).
// HBM -> LDS goes through registers (global_load_dwordx4 -> ds_write_b128), double-buffered:
// tile t+1's loads are issued before tile t's MFMAs and written to the other buffer after them,
// one barrier per k-step.
#include "mmf_internal.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
[[maybe_unused]] constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
[[maybe_unused]] constexpr int NTHREADS = 256;

struct GemmArgs {
  int nprob;
  int epi;
  int tile_start[MMF_GEMM_MAX_PROBLEMS + 1];
  mmf_gemm_problem p[MMF_GEMM_MAX_PROBLEMS];
};

// ---- LDS images -------------------------------------------------------------------------------
__device__ __forceinline__ int kc_off(int row, int chunk) {          // [128][64] bf16, 128-B rows
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}
__device__ __forceinline__ int kr_off(int krow, int chunk) {         // [64][128] bf16, 256-B rows
  return krow * 256 + ((chunk ^ (((krow & 3) << 2) | ((krow >> 2) & 3))) << 4);
}

// Stage one operand tile HBM -> registers.  KR == false: rows are the m/n index (count `rows`),
// columns are k.  KR == true: rows are k, columns the m/n index.  Out-of-range chunks read as 0.
#ifdef MMF_LEGACY_KERNELS   // first-generation 128x128 register-staged kernel: built with `make LEGACY=1` for A/B runs only
template <bool KR>
__device__ __forceinline__ void stage_load(u32x4_t (&r)[4], const unsigned short* __restrict__ base,
                                           int ld, int idx0, int idx_count, int k0, int K, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NTHREADS * i;
    int row, col, row_lim, col_lim;
    if (!KR) { row = idx0 + (c >> 3); col = k0 + ((c & 7) << 3); row_lim = idx_count; col_lim = K; }
    else     { row = k0 + (c >> 4);   col = idx0 + ((c & 15) << 3); row_lim = K; col_lim = idx_count; }
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < row_lim && col < col_lim)
      v = *reinterpret_cast<const u32x4_t*>(base + (size_t)row * ld + col);
    r[i] = v;
  }
}

template <bool KR>
__device__ __forceinline__ void stage_store(const u32x4_t (&r)[4], char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NTHREADS * i;
    const int off = KR ? kr_off(c >> 4, c & 15) : kc_off(c >> 3, c & 7);
    *reinterpret_cast<u32x4_t*>(tile + off) = r[i];
  }
}

// One MFMA operand fragment for the 16 indices [idx0, idx0+16) and k-substep ks (32 k values):
// lane l holds element e = 0..7 -> (idx0 + (l & 15), k = 32 ks + 8 (l >> 4) + e).
template <bool KR>
__device__ __forceinline__ bf16x8_t read_frag(const char* tile, int idx0, int ks, int lane) {
  if (!KR) {
    const int row = idx0 + (lane & 15);
    const int chunk = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8_t*>(tile + kc_off(row, chunk));
  } else {
    const int kb = ks * 32 + ((lane >> 4) << 3);
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int chunk = (idx0 >> 3) + (p >> 1);
    const int sub = (p & 1) << 3;
    s16x4_t lo = lds_read_tr16(tile + kr_off(kb + q, chunk) + sub);
    s16x4_t hi = lds_read_tr16(tile + kr_off(kb + 4 + q, chunk) + sub);
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

template <bool A_KR, bool B_KR, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 2)
void gemm_grouped_kernel(const GemmArgs args) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // [buf][A|B]

  // ---- which problem / tile am I --------------------------------------------------------------
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < args.nprob && bid >= args.tile_start[pi + 1]) ++pi;
  const mmf_gemm_problem& P = args.p[pi];
  const int M = P.M, N = P.N, K = P.K;
  const int tiles_m = (M + BM - 1) / BM;
  const int t = bid - args.tile_start[pi];
  const int m0 = (t % tiles_m) * BM;
  const int n0 = (t / tiles_m) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  const unsigned short* __restrict__ Ag = static_cast<const unsigned short*>(P.A);
  const unsigned short* __restrict__ Bg = static_cast<const unsigned short*>(P.B);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4_t ra[4], rb[4];
  const int nk = (K + BK - 1) / BK;

  // wgrad only: bias gradient = column sums of dy = sums over k of the A operand.  The n-tile-0
  // workgroup of every m-tile adds up the A chunks it stages anyway (a thread always stages the
  // same 8 columns: chunk id = tid + 256 i keeps (id & 15)), so db costs no extra HBM traffic.
  const bool do_colsum = A_KR && (args.epi & MMF_EPI_COLSUM_A) && n0 == 0;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto colsum_acc = [&]() {
    if (A_KR && do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[2 * e] += bf16lo(ra[i][e]); cs[2 * e + 1] += bf16hi(ra[i][e]); }
    }
  };

  stage_load<A_KR>(ra, Ag, P.lda, m0, M, 0, K, tid);
  stage_load<B_KR>(rb, Bg, P.ldb, n0, N, 0, K, tid);
  colsum_acc();                                   // (adds where the registers are consumed anyway,
  stage_store<A_KR>(ra, smem, tid);               //  never right behind the loads: that would stall)
  stage_store<B_KR>(rb, smem + TILE_BYTES, tid);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {                                   // issue next tile's HBM loads early
      stage_load<A_KR>(ra, Ag, P.lda, m0, M, (kt + 1) * BK, K, tid);
      stage_load<B_KR>(rb, Bg, P.ldb, n0, N, (kt + 1) * BK, K, tid);
    }
    const char* sA = smem + cur * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fm[4], fn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fn[i] = read_frag<B_KR>(sB, wn + i * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) fm[i] = read_frag<A_KR>(sA, wm + i * 16, ks, lane);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[tn], fm[tm], acc[tn][tm], 0, 0, 0);
    }
    if (more) {                                   // other buffer: its last readers passed the
      char* dA = smem + (cur ^ 1) * 2 * TILE_BYTES;   // barrier that ended the previous k-step
      colsum_acc();
      stage_store<A_KR>(ra, dA, tid);
      stage_store<B_KR>(rb, dA + TILE_BYTES, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  if (A_KR && do_colsum) {           // 16 row-groups (tid >> 4) x 16 column chunks (tid & 15) -> 128 sums
    float* red = reinterpret_cast<float*>(smem);           // main loop ended on a barrier: LDS is free
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 128 && m0 + tid < M) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g * 128 + tid];
      atomicAdd(const_cast<float*>(P.bias) + m0 + tid, t);
    }
  }

  // ---- epilogue: lane owns C[m][n..n+3] for each of its 16 MFMA tiles ------------------------------
  const int epi = args.epi;
  const unsigned short* __restrict__ aux = static_cast<const unsigned short*>(P.aux);
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + wm + tm * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + wn + tn * 16 + ((lane >> 4) << 2);
      if (n >= N) continue;
      f32x4_t v = acc[tn][tm];
      if (epi & MMF_EPI_BIAS) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(P.bias + n);
        v += b;
      }
      if (epi & MMF_EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (epi & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) {
        const u32x2_t a = *reinterpret_cast<const u32x2_t*>(aux + (size_t)m * P.ldaux + n);
        const float a0 = bf16lo(a[0]), a1 = bf16hi(a[0]), a2 = bf16lo(a[1]), a3 = bf16hi(a[1]);
        if (epi & MMF_EPI_MASK_AUX) {
          v[0] = a0 > 0.f ? v[0] : 0.f; v[1] = a1 > 0.f ? v[1] : 0.f;
          v[2] = a2 > 0.f ? v[2] : 0.f; v[3] = a3 > 0.f ? v[3] : 0.f;
        }
        if (epi & MMF_EPI_ADD_AUX) { v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3; }
      }
      if (OUT_F32) {
        float* c = static_cast<float*>(P.C) + (size_t)m * P.ldc + n;
        if (epi & MMF_EPI_ACCUM) v += *reinterpret_cast<const f32x4_t*>(c);
        *reinterpret_cast<f32x4_t*>(c) = v;
      } else {
        unsigned short* c = static_cast<unsigned short*>(P.C) + (size_t)m * P.ldc + n;
        u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(c) = o;
      }
    }
  }
}

template <bool A_KR, bool B_KR>
int launch(const GemmArgs& a, int total_tiles, int out_f32, hipStream_t s) {
  if (out_f32) hipLaunchKernelGGL((gemm_grouped_kernel<A_KR, B_KR, true>), dim3(total_tiles), dim3(NTHREADS), 0, s, a);
  else         hipLaunchKernelGGL((gemm_grouped_kernel<A_KR, B_KR, false>), dim3(total_tiles), dim3(NTHREADS), 0, s, a);
  return 0;
}

#endif  // MMF_LEGACY_KERNELS
}  // namespace

int mmf_gemm2_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm2.hip: LDS-DMA ring kernel

#ifdef MMF_LEGACY_KERNELS
int mmf_gemm3_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, hipStream_t s);      // gemm3.hip: persistent LDS-DMA ring kernel
#endif
int mmf_gemm4_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm4.hip: 256x256 tile
int mmf_gemm5_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm5.hip: NT, 32-deep k-step, 2 workgroups / CU

// Implementation switch (A/B runs in one process: tools/gemm_bench.py): 0 = automatic (default), 1 =
// register-staged 128x128 kernel of this file, 2 = 256x128 LDS-DMA ring (gemm2.hip), 3 = its persistent
// form (gemm3.hip), 4 = 256x256 LDS-DMA ring (gemm4.hip).  Default from MMF_GEMM_IMPL, else automatic.
static int g_gemm_impl = [] {
  const char* e = getenv("MMF_GEMM_IMPL");
  int v = (e && e[0] >= '0' && e[0] <= '5') ? e[0] - '0' : 0;
#ifndef MMF_LEGACY_KERNELS
  if (v == 1 || v == 3) v = 0;
#endif
  return v;
}();

// Automatic choice between the two ring kernels.  The 256x256 tile does 25 % fewer LDS fragment reads,
// 33 % fewer LDS-DMA issues per MFMA and half the L2 traffic (measured 1.27 vs 1.02 PFLOP/s at 4096^3,
// +22-26 % on the FFN1 / dH groups) but quantises coarser: it is used for NT / NN launches whose
// 256x256 tiling fills at least 80 % of the CU-rounds it occupies.  wgrad (TN) stays on 256x128
// (the transposed-read operands of the bigger wave tile do not fit 256 registers without spills).
static const int g_shortk_mode = [] { const char* e = getenv("MMF_GEMM_SHORTK"); return e ? atoi(e) : 1; }();
static const int g_shortk_nn = [] { const char* e = getenv("MMF_GEMM_SHORTK_NN"); return e ? atoi(e) : 0; }();
// MMF_GEMM_POLICY: 1 = round 1's rule (below), 2 = round 2's rule from the per-group microbenchmarks
// (profiles/r02_gemm_generations.txt: every MulT launch group x {256x128, 256x256, 256x128/32-deep} in isolation).
static const int g_tn5 = [] { const char* e = getenv("MMF_GEMM_TN5"); return e ? atoi(e) : 0; }();   // wgrad on the 32-deep two-workgroups-per-CU kernel
static const int g_policy = [] { const char* e = getenv("MMF_GEMM_POLICY"); return e ? atoi(e) : 2; }();
static int auto_impl(const mmf_gemm_problem* p, int n, int layout) {
  if (layout == MMF_GEMM_TN) return g_tn5 ? 5 : 2;
  long tiles = 0;
  int kmax = 0;
  for (int i = 0; i < n; ++i) {
    tiles += (long)((p[i].M + 255) / 256) * ((p[i].N + 255) / 256);
    kmax = p[i].K > kmax ? p[i].K : kmax;
  }
  static const int cus = [] { int c = mmf_device_cu_count(); return c > 0 ? c : 256; }();
  const long rounds = (tiles + cus - 1) / cus;
  if (g_policy >= 2) {
    // The 256x256 tile (1.2+ PF steady state) wins whenever its tiling either fills its CU rounds (>= 80 %) or is one
    // partial round of at least half the chip: a launch that leaves CUs idle still finishes sooner than two rounds of
    // the 256x128 tile (FFN2 / dX / out-projection groups of three MulT blocks: 177 tiles, +22 ... +32 %).  Otherwise:
    // short reductions go to the two-workgroups-per-CU kernel (NT and NN), long ones to the 256x128 ring (NT) or,
    // for NN, again to the 32-deep kernel (930 vs 897 TF on the dX groups).
    const bool fills = tiles * 100 >= rounds * cus * 80;
    const bool one_round = tiles <= cus && 2 * tiles >= cus;
    if (fills || one_round) return 4;
    if (kmax <= 1024) return 5;
    return layout == MMF_GEMM_NN ? 5 : 2;
  }
  if (tiles >= 2L * cus && tiles * 100 >= rounds * cus * 80) return 4;
  // NT launches with a short reduction that the 256x256 tiling does not fill: the 32-deep, two-workgroups-per-CU
  // form of the 256x128 kernel (one workgroup's pipeline fill / output burst under the other's MFMA loop):
  // +2 ... +9 % on the in-projection and out-projection launches of MulT (MMF_GEMM_SHORTK=0: off)
  if (layout == MMF_GEMM_NT && g_shortk_mode && kmax <= 1024) return 5;
  // NN with a short reduction (round 2, MMF_GEMM_SHORTK_NN=1 to enable): the same two-workgroups-per-CU form
  if (layout == MMF_GEMM_NN && g_shortk_nn && kmax <= 1024) return 5;
  return 2;
}
static int gemm_impl() { return g_gemm_impl; }
static thread_local int t_last_impl = 0;
extern "C" int mmf_gemm_last_impl(void) { return t_last_impl; }
extern "C" int mmf_gemm_select_impl(int impl) {
  if (impl < 0 || impl > 5) MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_select_impl: %d not in 0..5", impl);
#ifndef MMF_LEGACY_KERNELS
  if (impl == 1 || impl == 3)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_select_impl: generation %d (superseded) is only in builds made with `make LEGACY=1`", impl);
#endif
  g_gemm_impl = impl;
  return MMF_OK;
}

extern "C" int mmf_gemm_grouped(const mmf_gemm_problem* problems, int num_problems, int layout,
                                int epilogue, int out_f32, void* stream) {
  return mmf_gemm_grouped_ex(problems, num_problems, layout, epilogue, out_f32, nullptr, stream);
}

extern "C" int mmf_gemm_grouped_ex(const mmf_gemm_problem* problems, int num_problems, int layout,
                                   int epilogue, int out_f32, const mmf_gemm_extra* extra, void* stream) {
  if (!problems || num_problems <= 0 || num_problems > MMF_GEMM_MAX_PROBLEMS)
    MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped: num_problems=%d out of range [1,%d]", num_problems,
             MMF_GEMM_MAX_PROBLEMS);
  if (layout < MMF_GEMM_NT || layout > MMF_GEMM_TN)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: unknown layout %d", layout);
  if ((epilogue & MMF_EPI_ACCUM) && !out_f32)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: MMF_EPI_ACCUM needs f32 output");
  if ((epilogue & MMF_EPI_MASK_AUX) && (epilogue & MMF_EPI_ADD_AUX))
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: MASK_AUX and ADD_AUX are exclusive");
  if ((epilogue & MMF_EPI_COLSUM_A) && (layout != MMF_GEMM_TN || (epilogue & MMF_EPI_BIAS)))
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: COLSUM_A is a TN (wgrad) epilogue and excludes BIAS");
  const bool needs_v2 = extra && (extra->alpha != 1.f || (epilogue & MMF_EPI_DROPOUT));
  if (epilogue & MMF_EPI_DROPOUT) {
    if (!extra || !extra->rng_state || !(extra->dropout_p >= 0.f) || extra->dropout_p >= 1.f)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped_ex: MMF_EPI_DROPOUT needs rng_state and 0 <= p < 1");
  }
  int impl = gemm_impl();
  if (impl == 0) impl = auto_impl(problems, num_problems, layout);
  if (needs_v2 && impl != 4 && impl != 5) impl = 2;   // only gemm2 / gemm4 / gemm5 have the alpha / dropout epilogue
  t_last_impl = impl;
  GemmArgs a;
  a.nprob = num_problems;
  a.epi = epilogue;
  int total = 0;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_gemm_problem& p = problems[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: empty problem M=%d N=%d K=%d", i, p.M, p.N, p.K);
    if (!p.A || !p.B || !p.C) MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: null operand", i);
    if ((epilogue & (MMF_EPI_BIAS | MMF_EPI_COLSUM_A)) && !p.bias) MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: bias is null", i);
    if ((epilogue & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) && !p.aux)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: aux is null", i);
    // the contiguous extent of each operand must be a multiple of 8 elements (16-B vector loads)
    const int a_cols = (layout == MMF_GEMM_TN) ? p.M : p.K;
    const int b_cols = (layout == MMF_GEMM_NT) ? p.K : p.N;
    if ((a_cols & 7) || (b_cols & 7) || (p.lda & 7) || (p.ldb & 7) || (p.N & 3) || (p.ldc & 3) ||
        ((epilogue & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) && (p.ldaux & 3)))
      MMF_FAIL(MMF_E_ALIGN, "mmf_gemm_grouped[%d]: M=%d N=%d K=%d lda=%d ldb=%d ldc=%d ldaux=%d violate "
               "the 8-element (operands) / 4-element (output) granularity", i, p.M, p.N, p.K, p.lda,
               p.ldb, p.ldc, p.ldaux);
    if (p.lda < a_cols || p.ldb < b_cols || p.ldc < p.N)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: leading dimension smaller than the row", i);
    if (!mmf_aligned16(p.A) || !mmf_aligned16(p.B) || !mmf_aligned16(p.C) ||
        (p.bias && !mmf_aligned16(p.bias)) || (p.aux && (reinterpret_cast<uintptr_t>(p.aux) & 7)))
      MMF_FAIL(MMF_E_ALIGN, "mmf_gemm_grouped[%d]: operand pointers must be 16-byte aligned", i);
    a.tile_start[i] = total;
    total += ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    a.p[i] = p;
  }
  a.tile_start[num_problems] = total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (impl == 2) return mmf_gemm2_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  if (impl == 4) return mmf_gemm4_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  if (impl == 5) return mmf_gemm5_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
#ifdef MMF_LEGACY_KERNELS
  if (impl == 3) {
    bool wide_ok = true;              // the persistent kernel only has the 16-byte bf16 epilogue
    for (int i = 0; i < num_problems && !out_f32; ++i)
      wide_ok = wide_ok && !(problems[i].N & 7) && !(problems[i].ldc & 7);
    if (wide_ok) return mmf_gemm3_launch(problems, num_problems, layout, epilogue, out_f32, s);
    return mmf_gemm2_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  }
  switch (layout) {
    case MMF_GEMM_NT: launch<false, false>(a, total, out_f32, s); break;
    case MMF_GEMM_NN: launch<false, true>(a, total, out_f32, s); break;
    default:          launch<true, true>(a, total, out_f32, s); break;
  }
  MMF_CHECK_LAUNCH("mmf_gemm_grouped");
  return MMF_OK;
#else
  (void)a; (void)total;
  MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: kernel generation %d is not in this build (make LEGACY=1)", impl);
#endif
}
