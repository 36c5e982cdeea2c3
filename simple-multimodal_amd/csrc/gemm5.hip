// Grouped bf16 GEMM, NT and (round 2) NN / TN layouts, short-K variant: the 256 x 128 LDS-DMA ring of gemm2.hip with a 32-deep k-step
// and a 72 KiB ring, so that TWO workgroups fit a CU (2 x 72 KiB LDS, <= 128 registers, 16 waves per CU).
//
// Why: at K = 768 a 256 x 128 tile spends ~4 us in its MFMA loop and about as long filling its pipeline and
// writing its output (all CUs burst at tile start and end in lockstep); with one workgroup per CU (gemm2: 144 KiB,
// gemm4: 128 KiB of LDS) nothing overlaps those phases.  With two co-resident workgroups one computes while the
// other loads or stores.
//   tile 256 (m) x 128 (n) x 32 (k), 8 waves as 4 x 2, wave tile 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16 per k-step;
//   LDS image: [rows][32 k] = 64-B rows, 16-B chunk slot = chunk ^ (2 * ((row >> 3) & 1))  (guide st_16x32 swizzle:
//   the four ds_read_b128 lane groups each see 16 distinct bank quads), filled lane-linearly by LDS-DMA with the
//   swizzle on the source address; 3 stages x 24 KiB, 3 DMA pieces per wave per k-step, counted vmcnt(3).
// NN (dgrad: the B operand is W[k][n], its reduction index is the memory row): the B tile uses gemm2.hip's KR image
// ([32 k][128 n] = 256-B rows, chunk XOR swizzle, 8 one-KiB DMA pieces of 4 k-rows) and is read back transposed with
// ds_read_b64_tr_b16 from inline asm (hipcc would drain the LDS-DMA in flight before the builtin).
// Epilogue and launch conventions are gemm2.hip's.
#include "mmf_internal.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int NTHREADS = 512;
constexpr int A_BYTES = BM * BK * 2;            // 16 KiB
constexpr int B_BYTES = BN * BK * 2;            // 8 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // 24 KiB
constexpr int STAGES = 3;
constexpr int LOADS_PER_TILE = STAGE_BYTES / (NTHREADS * 16);   // 3 LDS-DMA instructions per wave per tile
constexpr unsigned OOB = 0x80000000u;

struct GemmArgs {
  int nprob;
  int epi;
  int xcd_granule;                   // mmf_xcd_tile()
  float alpha;                       // multiplies the result after the mask step
  unsigned drop_thresh, site;        // MMF_EPI_DROPOUT
  const unsigned long long* rng_state;
  int tile_start[MMF_GEMM_MAX_PROBLEMS + 1];
  mmf_gemm_problem p[MMF_GEMM_MAX_PROBLEMS];
};

// 64-byte rows (32 k), st_16x32 swizzle: slot = chunk ^ (2 * ((row >> 3) & 1))
__device__ __forceinline__ int kc32_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 3) & 1) << 1)) << 4); }

typedef __attribute__((address_space(3))) void lds_void_t;

// LDS-DMA of the 1-KiB piece c = rows 16c .. 16c+15 (64 B each) of a [rows][32 k] operand tile; lane (r = l >> 2,
// slot s = l & 3) fetches the 16-B chunk that the swizzle puts at slot s of row r (same 64-B segment: coalescing
// is unchanged).  Rows / columns past the matrix: explicit out-of-range offset -> zeros.
__device__ __forceinline__ void dma_piece32(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, int c, int ld,
                                            int idx0, int idx_count, int k0, int K, int lane) {
  const int r = lane >> 2, s = lane & 3;
  const int row = idx0 + 16 * c + r;
  const int col = k0 + ((s ^ (((r >> 3) & 1) << 1)) << 3);
  const bool ok = row < idx_count && col < K;
  const unsigned voff = ok ? (unsigned)(row * ld + col) * 2u : OOB;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, voff, 0, 0, 0);
}

// ---- KR image of the NN layout's B tile (gemm2.hip kr_*: 256-B k-rows, guide T10 image (b)) ----
__device__ __forceinline__ int kr_swz(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }
__device__ __forceinline__ int kr_off(int krow, int chunk) { return krow * 256 + ((chunk ^ kr_swz(krow)) << 4); }
// LDS-DMA of piece c = k-rows 4c .. 4c+3 (256 B each) of a [32 k][128 idx] tile
__device__ __forceinline__ void dma_piece_kr(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, int c, int ld,
                                             int idx0, int idx_count, int k0, int K, int lane) {
  const int krow = 4 * c + (lane >> 4), s = lane & 15;
  const int row = k0 + krow;
  const int col = idx0 + ((s ^ kr_swz(krow)) << 3);
  const bool ok = row < K && col < idx_count;
  const unsigned voff = ok ? (unsigned)(row * ld + col) * 2u : OOB;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ s16x4_t tr_read_asm(const char* p) {
  s16x4_t r;
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
// transposed fragment of the 16 idx [idx0, idx0 + 16) x 32 k of a KR tile: lane l = (idx idx0 + (l & 15), k chunk l >> 4);
// a 256-idx tile (the TN layout's A operand) is two 128-idx halves of 8 KiB each
__device__ __forceinline__ void read_frag_kr32(const char* tile, int idx0, int lane, s16x4_t& lo, s16x4_t& hi) {
  tile += (idx0 >> 7) * 8192;
  idx0 &= 127;
  const int kb = (lane >> 4) << 3;
  const int q = (lane >> 2) & 3, p = lane & 3;
  const int chunk = (idx0 >> 3) + (p >> 1);
  const int sub = (p & 1) << 3;
  lo = tr_read_asm(tile + kr_off(kb + q, chunk) + sub);
  hi = tr_read_asm(tile + kr_off(kb + 4 + q, chunk) + sub);
}
__device__ __forceinline__ bf16x8_t tr_join(const s16x4_t& lo, const s16x4_t& hi) {
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// fragment of 16 rows x 32 k: lane l holds row idx0 + (l & 15), k chunk l >> 4
__device__ __forceinline__ bf16x8_t read_frag32(const char* tile, int idx0, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(tile + kc32_off(idx0 + (lane & 15), lane >> 4));
}

template <bool A_KR, bool B_KR, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 4)      // <= 128 registers: two 8-wave workgroups per CU
void gemm5_grouped_kernel(const GemmArgs args, const int total_tiles) {
  __shared__ __attribute__((aligned(1024))) char smem[STAGES * STAGE_BYTES];

  const int bid = mmf_xcd_tile(blockIdx.x, total_tiles, args.xcd_granule);   // balanced tile -> XCD map (mmf_internal.h)
  int pi = 0;
  while (pi + 1 < args.nprob && bid >= args.tile_start[pi + 1]) ++pi;
  const mmf_gemm_problem& P = args.p[pi];
  const int M = P.M, N = P.N, K = P.K;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int t = bid - args.tile_start[pi];
  // Super-rows of GROUP_M m-tiles, n fastest across a super-row: the ~32 tiles an XCD runs at the same
  // time then form a GROUP_M x 4 block of the output and share A and B panels in that XCD's L2; with a
  // plain m-fastest order they would be 32 different m-tiles of one n-tile (no A reuse: every A slice
  // would stream from the Infinity Cache at ~1/2 the L2 rate).
  constexpr int GROUP_M = 8;
  const int grp = t / (GROUP_M * tiles_n), rem = t % (GROUP_M * tiles_n);
  const int gm = min(GROUP_M, tiles_m - grp * GROUP_M);
  const int m0 = (grp * GROUP_M + rem % gm) * BM;
  const int n0 = (rem / gm) * BN;
  (void)tiles_n;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // buffer descriptors (wave-uniform: built from kernel arguments only)
  const int a_rows = A_KR ? K : M, b_rows = B_KR ? K : N;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(P.A), 0, (int)((size_t)a_rows * P.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(P.B), 0, (int)((size_t)b_rows * P.ldb * 2), 0x00020000);

  auto issue_tile = [&](int kt) {
    char* st = smem + (kt % STAGES) * STAGE_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                   // A: 16 pieces, two per wave
      const int c = wave + 8 * i;
      if (A_KR) dma_piece_kr(rsA, st + c * 1024, c & 7, P.lda, m0 + 128 * (c >> 3), M, k0, K, lane);   // half c >> 3
      else      dma_piece32(rsA, st + c * 1024, c, P.lda, m0, M, k0, K, lane);
    }
    if (B_KR) dma_piece_kr(rsB, st + A_BYTES + wave * 1024, wave, P.ldb, n0, N, k0, K, lane);   // B: 8 pieces, one per wave
    else      dma_piece32(rsB, st + A_BYTES + wave * 1024, wave, P.ldb, n0, N, k0, K, lane);
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fused bias gradient (wgrad): column sums of A via an all-ones n-fragment (gemm2.hip)
  const bool do_colsum = A_KR && (args.epi & MMF_EPI_COLSUM_A) && n0 == 0 && wn == 0;
  f32x4_t csum[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) csum[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const s16x8_t ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  const int nk = (K + BK - 1) / BK;
  issue_tile(0);
  if (nk > 1) issue_tile(1);

  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (this wave's pieces); the younger tile's three loads stay in flight
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS_PER_TILE) : "memory");
    else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // every wave's pieces landed; stage (kt+2)%3 is free
    asm volatile("" ::: "memory");
    const char* sA = smem + (kt % STAGES) * STAGE_BYTES;
    const char* sB = sA + A_BYTES;
    if (kt + 2 < nk) issue_tile(kt + 2);
    bf16x8_t fm[4], fn[4];
    if (A_KR) {                                     // TN: both operands transposed; two batches share the temporaries
      s16x4_t lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) read_frag_kr32(sB, wn + i * 16, lane, lo[i], hi[i]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fn[i] = tr_join(lo[i], hi[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) read_frag_kr32(sA, wm + i * 16, lane, lo[i], hi[i]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fm[i] = tr_join(lo[i], hi[i]);
    } else if (B_KR) {
      s16x4_t nlo[4], nhi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) read_frag_kr32(sB, wn + i * 16, lane, nlo[i], nhi[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fm[i] = read_frag32(sA, wm + i * 16, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fn[i] = tr_join(nlo[i], nhi[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { fn[i] = read_frag32(sB, wn + i * 16, lane); fm[i] = read_frag32(sA, wm + i * 16, lane); }
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[tn], fm[tm], acc[tn][tm], 0, 0, 0);
    if (A_KR && do_colsum) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) csum[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm[tm], csum[tm], 0, 0, 0);
    }
  }
  if (A_KR && do_colsum && lane < 16) {          // every MFMA row holds the same sums: take row 0 (lanes 0..15, reg 0)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      const int m = m0 + wm + tm * 16 + lane;
      if (m < M) atomicAdd(const_cast<float*>(P.bias) + m, csum[tm][0]);
    }
  }

  // ---- epilogue: lane owns C[m][n..n+3] for each of its 16 MFMA tiles -------------------------------
  const int epi = args.epi;
  const unsigned short* __restrict__ aux = static_cast<const unsigned short*>(P.aux);
  const bool do_drop = epi & MMF_EPI_DROPOUT;
  const unsigned drop_key = do_drop ? mmf_rng_key(*args.rng_state, args.site, (unsigned)pi) : 0u;
  const float drop_scale = do_drop ? 1.f / (1.f - (float)args.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  const float alpha = args.alpha;
  auto finish = [&](f32x4_t v, int m, int n) -> f32x4_t {  // bias -> relu -> dropout -> mask -> alpha -> residual
    if (epi & MMF_EPI_BIAS) v += *reinterpret_cast<const f32x4_t*>(P.bias + n);
    if (epi & MMF_EPI_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (do_drop) {
      const unsigned idx = (unsigned)m * (unsigned)N + (unsigned)n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mmf_keep(drop_key, idx + e, args.drop_thresh) ? v[e] * drop_scale : 0.f;
    }
    if (epi & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) {
      const u32x2_t a = *reinterpret_cast<const u32x2_t*>(aux + (size_t)m * P.ldaux + n);
      const float a0 = bf16lo(a[0]), a1 = bf16hi(a[0]), a2 = bf16lo(a[1]), a3 = bf16hi(a[1]);
      if (epi & MMF_EPI_MASK_AUX) {
        v[0] = a0 > 0.f ? v[0] : 0.f; v[1] = a1 > 0.f ? v[1] : 0.f;
        v[2] = a2 > 0.f ? v[2] : 0.f; v[3] = a3 > 0.f ? v[3] : 0.f;
      }
      v *= alpha;
      if (epi & MMF_EPI_ADD_AUX) { v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3; }
    } else {
      v *= alpha;
    }
    return v;
  };
  if (OUT_F32) {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      const int m = m0 + wm + tm * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + wn + tn * 16 + ((lane >> 4) << 2);
        if (n >= N) continue;
        f32x4_t v = finish(acc[tn][tm], m, n);
        float* c = static_cast<float*>(P.C) + (size_t)m * P.ldc + n;
        if (epi & MMF_EPI_ACCUM) v += *reinterpret_cast<const f32x4_t*>(c);
        *reinterpret_cast<f32x4_t*>(c) = v;
      }
    }
  } else if ((N & 7) == 0 && (P.ldc & 7) == 0) {
    // bf16 output, 16-byte stores (guide T21): the tail of a short-K tile is store-ISSUE bound, so the
    // 8-byte pieces of two neighbouring MFMA column tiles are exchanged between the lane groups
    // g = lane>>4 and g^1 with v_permlane16_swap; even groups then hold 8 consecutive columns of tile
    // tn, odd groups of tile tn+1: 8 store instructions per wave instead of 16, same bytes and lines.
    const int g = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      const int m = m0 + wm + tm * 16 + (lane & 15);
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const int nA = n0 + wn + (2 * tp) * 16 + (g << 2), nB = nA + 16;
        const bool okA = m < M && nA < N, okB = m < M && nB < N;
        f32x4_t va = acc[2 * tp][tm], vb = acc[2 * tp + 1][tm];
        if (okA) va = finish(va, m, nA);
        if (okB) vb = finish(vb, m, nB);
        unsigned a0 = pack_bf16x2(va[0], va[1]), a1 = pack_bf16x2(va[2], va[3]);
        unsigned b0 = pack_bf16x2(vb[0], vb[1]), b1 = pack_bf16x2(vb[2], vb[3]);
        const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        // even g: {own tile-A cols 4g..4g+3 | group g+1's tile-A cols}; odd g: {group g-1's tile-B | own tile-B}
        const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
        const int n = (g & 1) ? nB - 4 : nA;
        if (m < M && n < N)
          *reinterpret_cast<u32x4_t*>(static_cast<unsigned short*>(P.C) + (size_t)m * P.ldc + n) = o;
      }
    }
  } else {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      const int m = m0 + wm + tm * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + wn + tn * 16 + ((lane >> 4) << 2);
        if (n >= N) continue;
        const f32x4_t v = finish(acc[tn][tm], m, n);
        const u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(static_cast<unsigned short*>(P.C) + (size_t)m * P.ldc + n) = o;
      }
    }
  }
}

}  // namespace

// called by mmf_gemm_grouped (gemm.hip) after validation
int mmf_gemm5_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s) {
  GemmArgs a;
  a.nprob = num_problems;
  a.epi = epilogue;
  a.xcd_granule = mmf_xcd_granule();
  a.alpha = extra ? extra->alpha : 1.f;
  a.drop_thresh = extra ? mmf_drop_thresh(extra->dropout_p) : 0u;
  a.site = extra ? extra->site : 0u;
  a.rng_state = extra ? reinterpret_cast<const unsigned long long*>(extra->rng_state) : nullptr;
  int total = 0;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_gemm_problem& p = problems[i];
    // 32-bit byte offsets inside the buffer descriptors
    const size_t a_bytes = (size_t)(layout == MMF_GEMM_TN ? p.K : p.M) * p.lda * 2;
    const size_t b_bytes = (size_t)(layout == MMF_GEMM_NT ? p.N : p.K) * p.ldb * 2;
    if (a_bytes >= 0x7fffffffull || b_bytes >= 0x7fffffffull)
      MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped[%d]: operand larger than 2 GiB", i);
    a.tile_start[i] = total;
    total += ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    a.p[i] = p;
  }
  a.tile_start[num_problems] = total;
  if (layout == MMF_GEMM_TN) {
    if (out_f32) hipLaunchKernelGGL((gemm5_grouped_kernel<true, true, true>), dim3(total), dim3(NTHREADS), 0, s, a, total);
    else         hipLaunchKernelGGL((gemm5_grouped_kernel<true, true, false>), dim3(total), dim3(NTHREADS), 0, s, a, total);
  } else if (layout == MMF_GEMM_NN) {
    if (out_f32) hipLaunchKernelGGL((gemm5_grouped_kernel<false, true, true>), dim3(total), dim3(NTHREADS), 0, s, a, total);
    else         hipLaunchKernelGGL((gemm5_grouped_kernel<false, true, false>), dim3(total), dim3(NTHREADS), 0, s, a, total);
  } else {
    if (out_f32) hipLaunchKernelGGL((gemm5_grouped_kernel<false, false, true>), dim3(total), dim3(NTHREADS), 0, s, a, total);
    else         hipLaunchKernelGGL((gemm5_grouped_kernel<false, false, false>), dim3(total), dim3(NTHREADS), 0, s, a, total);
  }
  MMF_CHECK_LAUNCH("mmf_gemm_grouped(v5)");
  return MMF_OK;
}
