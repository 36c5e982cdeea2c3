// Grouped bf16 GEMM, LDS-DMA ring kernel, 256 x 256 TILE variant (see gemm2.hip for the 256 x 128 form and the
// commentary): 8 waves as 2 (m) x 4 (n), each wave 128 x 64 = 8 x 4 MFMA tiles; 2-stage ring of 64 KiB stages.
// Per MFMA it reads 25 % fewer fragment bytes from LDS and issues 33 % fewer LDS-DMA pieces, and it halves the
// L2 -> LDS traffic; it needs >= ~2 tiles per CU to pay for its coarser quantisation (selected per launch).
//
//   tile      256 (m) x 128 (n) x 64 (k) per 512-thread workgroup, 8 waves as 4 (m) x 2 (n), each wave
//             64 x 64 = 4 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16), n on the MFMA row/register axis;
//   staging   HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), no
//             staging registers and no ds_write pass; a 3-stage ring (3 x 48 KiB = 144 KiB of the CU's
//             160 KiB), tile kt+2 is issued while tile kt is multiplied;
//   sync      ONE raw s_barrier per k-step, behind a counted s_waitcnt vmcnt(6) that leaves the next
//             tile's six loads in flight (never vmcnt(0) in the loop; guide section 5 "Pipelining
//             across barriers");
//   images    the same two LDS images as gemm.hip (KC: XOR (row&7) on 128-B rows; KR: guide T10
//             image (b) on 256-B rows).  LDS-DMA writes lane-linearly, so the swizzle is applied to the
//             per-lane SOURCE address and again on the read (guide rule 21): lane (row r, slot s) of a
//             1-KiB piece fetches global chunk s ^ f(r) of the same row, i.e. coalescing is unchanged;
//   bounds    buffer descriptors: rows/columns past the matrix give an out-of-range offset, the
//             hardware range check returns zeros into LDS;
//   order     tiles go to the XCDs in granules of 32 consecutive ids (mmf_xcd_tile, mmf_internal.h);
//   db        wgrad's fused bias gradient is one extra MFMA per (k-substep, m-tile) against an
//             all-ones fragment — no extra LDS or HBM traffic.
// Round 3: the kernel body is split into tile_mainloop (k-steps [kt0, kt1) of a tile) and tile_epilogue.  A stream-K form
// built on them (k-step units dealt evenly to 256 workgroups, partial tiles meeting in a workspace) was parity-green and
// bit-reproducible but not faster — profiles/r03_streamk_negative.txt, git history — and was taken out again.
#include "mmf_internal.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TM = 8, TN = 4;                  // MFMA tiles per wave: 128 (m) x 64 (n)
constexpr int NTHREADS = 512;
constexpr int A_BYTES = BM * BK * 2;            // 32 KiB
constexpr int B_BYTES = BN * BK * 2;            // 32 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // 64 KiB
constexpr int STAGES = 2;
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

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int kr_swz(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }
__device__ __forceinline__ int kr_off(int krow, int chunk) { return krow * 256 + ((chunk ^ kr_swz(krow)) << 4); }

typedef __attribute__((address_space(3))) void lds_void_t;

// Issue the LDS-DMA loads of one 1-KiB piece `c` of an operand tile into `lds_piece` (wave-uniform).
//   KC image: piece c = rows 8c .. 8c+7 (128 B each) of a [rows][64 k] tile.
//   KR image: piece c = k-rows 4(c&15) .. +3 (256 B each) of half (c>>4) of a [64 k][128 idx] x halves tile.
template <bool KR>
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, int c, int ld,
                                          int idx0, int idx_count, int k0, int K, int lane) {
  int row, col;
  bool ok;
  if (!KR) {
    const int r = lane >> 3, s = lane & 7;
    row = idx0 + 8 * c + r;                         // m / n index
    col = k0 + ((s ^ r) << 3);                      // k (8c + r) & 7 == r
    ok = row < idx_count && col < K;
  } else {
    const int krow = 4 * (c & 15) + (lane >> 4), s = lane & 15;
    row = k0 + krow;                                // k
    col = idx0 + 128 * (c >> 4) + ((s ^ kr_swz(krow)) << 3);
    ok = row < K && col < idx_count;
  }
  const unsigned voff = ok ? (unsigned)(row * ld + col) * 2u : OOB;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, voff, 0, 0, 0);
}

// KC fragment: plain 16-byte LDS load (the compiler tracks its lgkmcnt).
__device__ __forceinline__ bf16x8_t read_frag_kc(const char* tile, int idx0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(tile + kc_off(idx0 + (lane & 15), ks * 4 + (lane >> 4)));
}

// KR fragment: two ds_read_b64_tr_b16.  They are issued from inline asm on purpose: hipcc treats the
// ds_read_tr builtin as aliasing the LDS-DMA still in flight and drains it with s_waitcnt vmcnt(0)
// before every such read (one tile of prefetch lost per k-step).  From asm the reads are invisible
// to that bookkeeping; the caller waits with tr_wait() before the first use (guide 5.7 / rule 18).
__device__ __forceinline__ s16x4_t tr_read_asm(const char* p) {
  s16x4_t r;
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ void read_frag_kr(const char* tile, int idx0, int ks, int lane, s16x4_t& lo, s16x4_t& hi) {
  const char* half = tile + (idx0 >> 7) * 16384;
  const int kb = ks * 32 + ((lane >> 4) << 3);
  const int q = (lane >> 2) & 3, p = lane & 3;
  const int chunk = ((idx0 & 127) >> 3) + (p >> 1);
  const int sub = (p & 1) << 3;
  lo = tr_read_asm(half + kr_off(kb + q, chunk) + sub);
  hi = tr_read_asm(half + kr_off(kb + 4 + q, chunk) + sub);
}
__device__ __forceinline__ void tr_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8_t tr_join(const s16x4_t& lo, const s16x4_t& hi) {
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// k-steps [kt0, kt1) of the 256 x 256 tile at (m0, n0): acc += A[m0.., k] B[n0.., k]^T (KC / KR images per layout).
template <bool A_KR, bool B_KR>
__device__ __forceinline__ void tile_mainloop(const mmf_gemm_problem& P, const int m0, const int n0, const int kt0, const int kt1,
                                              char* smem, f32x4_t (&acc)[TN][TM], f32x4_t (&csum)[TM], const bool do_colsum,
                                              const int wave, const int lane, const int wm, const int wn) {
  const int M = P.M, N = P.N, K = P.K;
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
    for (int i = 0; i < 4; ++i) {                   // A: pieces wave, wave+8, wave+16, wave+24
      const int c = wave + 8 * i;
      dma_piece<A_KR>(rsA, st + c * 1024, c, P.lda, m0, M, k0, K, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                   // B: pieces wave, wave+8, wave+16, wave+24
      const int c = wave + 8 * i;
      dma_piece<B_KR>(rsB, st + A_BYTES + c * 1024, c, P.ldb, n0, N, k0, K, lane);
    }
  };

  const s16x8_t ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  issue_tile(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    // 2-stage ring, prefetch distance one tile: only tile kt's loads are outstanding here
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // every wave's pieces landed; the other stage is free
    asm volatile("" ::: "memory");

    const char* sA = smem + (kt % STAGES) * STAGE_BYTES;
    const char* sB = sA + A_BYTES;
    auto compute = [&](int ks) {
      // n-fragments once, m-fragments in two halves of four: 32 fragment registers live instead of 48
      // (with transposed-read operands the full set does not fit 256 registers next to 128 accumulators)
      bf16x8_t fn[TN];
      s16x4_t nlo[TN], nhi[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if (B_KR) read_frag_kr(sB, wn + i * 16, ks, lane, nlo[i], nhi[i]);
        else      fn[i] = read_frag_kc(sB, wn + i * 16, ks, lane);
      }
#pragma unroll
      for (int hm = 0; hm < 2; ++hm) {
        bf16x8_t fm[4];
        s16x4_t mlo[4], mhi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (A_KR) read_frag_kr(sA, wm + (4 * hm + i) * 16, ks, lane, mlo[i], mhi[i]);
          else      fm[i] = read_frag_kc(sA, wm + (4 * hm + i) * 16, ks, lane);
        }
        if (A_KR || B_KR) {
          tr_wait();
          if (hm == 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i) if (B_KR) fn[i] = tr_join(nlo[i], nhi[i]);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) if (A_KR) fm[i] = tr_join(mlo[i], mhi[i]);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[tn][4 * hm + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[tn], fm[i], acc[tn][4 * hm + i], 0, 0, 0);
        if (A_KR && do_colsum) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            csum[4 * hm + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm[i], csum[4 * hm + i], 0, 0, 0);
        }
      }
    };
    // role split between the two waves of a SIMD (waves w and w+4), see gemm2.hip
    const bool want_prefetch = kt + 1 < kt1;
    if (want_prefetch && wave < 4) issue_tile(kt + 1);
    compute(0);
    if (want_prefetch && wave >= 4) issue_tile(kt + 1);
    compute(1);
  }
}

// epilogue: lane owns C[m][n..n+3] for each of its 32 MFMA tiles
template <bool OUT_F32>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& args, const mmf_gemm_problem& P, const int pi, const int m0,
                                              const int n0, f32x4_t (&acc)[TN][TM], const int lane, const int wm, const int wn) {
  const int M = P.M, N = P.N;
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
    for (int tm = 0; tm < TM; ++tm) {
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
    for (int tm = 0; tm < TM; ++tm) {
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
    for (int tm = 0; tm < TM; ++tm) {
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

// tile t of problem P -> (m0, n0).  Super-rows of GROUP_M m-tiles, n fastest across a super-row: the ~32 tiles an XCD runs
// at the same time then form a GROUP_M x 4 block of the output and share A and B panels in that XCD's L2; with a plain
// m-fastest order they would be 32 different m-tiles of one n-tile (no A reuse: every A slice would stream from the
// Infinity Cache at ~1/2 the L2 rate).
__device__ __forceinline__ void tile_origin(const mmf_gemm_problem& P, const int t, int& m0, int& n0) {
  constexpr int GROUP_M = 8;
  const int tiles_m = (P.M + BM - 1) / BM, tiles_n = (P.N + BN - 1) / BN;
  const int grp = t / (GROUP_M * tiles_n), rem = t % (GROUP_M * tiles_n);
  const int gm = min(GROUP_M, tiles_m - grp * GROUP_M);
  m0 = (grp * GROUP_M + rem % gm) * BM;
  n0 = (rem / gm) * BN;
}

template <bool A_KR, bool B_KR, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 2)
void gemm4_grouped_kernel(const GemmArgs args, const int total_tiles) {
  __shared__ __attribute__((aligned(1024))) char smem[STAGES * STAGE_BYTES];

  const int bid = mmf_xcd_tile(blockIdx.x, total_tiles, args.xcd_granule);   // balanced tile -> XCD map (mmf_internal.h)
  int pi = 0;
  while (pi + 1 < args.nprob && bid >= args.tile_start[pi + 1]) ++pi;
  const mmf_gemm_problem& P = args.p[pi];
  int m0, n0;
  tile_origin(P, bid - args.tile_start[pi], m0, n0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // fused bias gradient (wgrad): column sums of A via an all-ones n-fragment
  const bool do_colsum = A_KR && (args.epi & MMF_EPI_COLSUM_A) && n0 == 0 && wn == 0;
  f32x4_t csum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) csum[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  tile_mainloop<A_KR, B_KR>(P, m0, n0, 0, (P.K + BK - 1) / BK, smem, acc, csum, do_colsum, wave, lane, wm, wn);

  if (A_KR && do_colsum) {              // every MFMA row holds the same sums: take row 0 (lanes 0..15, reg 0)
    if (lane < 16) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + wm + tm * 16 + lane;
        if (m < P.M) atomicAdd(const_cast<float*>(P.bias) + m, csum[tm][0]);
      }
    }
  }
  tile_epilogue<OUT_F32>(args, P, pi, m0, n0, acc, lane, wm, wn);
}

template <bool A_KR, bool B_KR>
void launch(const GemmArgs& a, int total, int out_f32, hipStream_t s) {
  if (out_f32) hipLaunchKernelGGL((gemm4_grouped_kernel<A_KR, B_KR, true>), dim3(total), dim3(NTHREADS), 0, s, a, total);
  else         hipLaunchKernelGGL((gemm4_grouped_kernel<A_KR, B_KR, false>), dim3(total), dim3(NTHREADS), 0, s, a, total);
}
}  // namespace

// called by mmf_gemm_grouped (gemm.hip) after validation
int mmf_gemm4_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
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
  switch (layout) {
    case MMF_GEMM_NT: launch<false, false>(a, total, out_f32, s); break;
    case MMF_GEMM_NN: launch<false, true>(a, total, out_f32, s); break;
    default:          launch<true, true>(a, total, out_f32, s); break;
  }
  MMF_CHECK_LAUNCH("mmf_gemm_grouped(v4)");
  return MMF_OK;
}
