// Grouped bf16 GEMM, ONE WAVE PER SIMD form (round 3): 256 x 256 tile, 4 waves as 2 (m) x 2 (n), each wave 128 x 128 =
// 4 x 4 tiles of v_mfma_f32_32x32x16_bf16 with the 256 accumulator registers in the AGPR half of a 512-register wave.
//
// Why another kernel.  The 8-wave ring kernels (gemm2 / gemm4 / gemm5) spend 2,770-2,900 cycles per k-step for 1,024 (256 x 128)
// or 2,048 (256 x 256) cycles of MFMA: after the k-step's barrier every wave issues its LDS-DMA pieces (the CU's fill path moves
// ~60 B/clk, so the burst blocks all eight instruction streams for ~800-1,000 cycles), then every wave reads fragments, then the
// two waves of a SIMD take turns on the matrix pipe — the CU's three resources are used one after the other.  Here
//   * a wave owns its SIMD: nothing it does can be covered by a partner, so its stream is software-pipelined by hand: the
//     fragments of k-substep g+1 are requested one behind each of the first MFMAs of substep g, the LDS-DMA pieces of a later stage
//     go out behind MFMAs that carry no read, and the stage hand-over (counted vmcnt, one s_barrier) sits inside the last substep's
//     MFMAs (measured: free there);
//   * the 128 x 128 wave tile needs 8 fragments per 16 MFMAs (the 128 x 64 tile of gemm4: 12; gemm2's 64 x 64: 16) — a third
//     fewer LDS bytes per MFMA;
//   * every fragment read is inline asm (ds_read_b128 / ds_read_b64_tr_b16) behind counted lgkmcnt waits tied to the destination
//     registers: the compiler's LDS-DMA alias bookkeeping never drains the ring.
// LDS: a ring of NS stages of BK k-columns; a stage holds the m-operand tile and the n-operand tile in the image of lds_image.h —
// [256][BK] for an operand whose reduction index is contiguous in memory (x[m][k], W[n][k]; row reads), [BK][256] for one whose
// reduction index is the memory row (W[k][n], dy[k][m], x[k][n]; transposed reads in the standard MFMA k order, so the two kinds mix).
// n rides the MFMA row / register axis (first operand = n-fragment): a lane holds 4 consecutive n of one output row.
#include "gemm6_parts.h"


// MMF_G6_DBG (build-time ablation bits, timing only, results wrong): 1 no LDS-DMA in the loop, 2 no stage hand-over (vmcnt +
// barrier), 4 no fragment reads in the loop, 16 three s_nop in place of every piece, 32 every refill fetches the same stage,
// 64 no epilogue
#ifndef MMF_G6_DBG
#define MMF_G6_DBG 0
#endif

namespace {

// the whole kernel as a device function (the __global__ wrapper below only owns the LDS): with the inline-asm reads reachable
// directly from a __global__ template hipcc's host pass dropped the kernel's launch stub without a diagnostic
template <bool A_KR, bool B_KR, int BK, int NS, bool OUT_F32>
__device__ __forceinline__ void gemm6_body(const GemmArgs& args, const int total_tiles, char* smem) {
  constexpr int TILE = 256 * BK * 2, STAGE = 2 * TILE, NG = BK / 16, PPO = BK / 8, PPW = 2 * PPO;
  constexpr int WA = A_KR ? 256 : BK, WB = B_KR ? 256 : BK;
  static_assert(NG == 2 || NG == 4, "BK is 32 or 64");

  const int bid = mmf_xcd_tile(blockIdx.x, total_tiles, args.xcd_granule);
  int pi = 0;
  while (pi + 1 < args.nprob && bid >= args.tile_start[pi + 1]) ++pi;
  const mmf_gemm_problem& P = args.p[pi];
  int m0, n0;
  tile_origin(P, bid - args.tile_start[pi], m0, n0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                   // this wave's 128 x 128 quadrant
  const int M = P.M, N = P.N, K = P.K;
  const int KT = (K + BK - 1) / BK;                          // a KR operand's rows past K read as zeros (range check); KC operands: K % BK == 0 (host)

  // ---- LDS-DMA: this wave's PPW pieces of a stage (PPO of the m-operand tile, PPO of the n-operand tile), offsets once --------
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPO; ++i) {
    voff[i] = piece_voff<A_KR, BK>(wave + 4 * i, P.lda, lane);
    voff[PPO + i] = piece_voff<B_KR, BK>(wave + 4 * i, P.ldb, lane);
  }
  const unsigned short* Ab = static_cast<const unsigned short*>(P.A) + (A_KR ? (size_t)m0 : (size_t)m0 * P.lda);
  const unsigned short* Bb = static_cast<const unsigned short*>(P.B) + (B_KR ? (size_t)n0 : (size_t)n0 * P.ldb);
  // bytes from the tile's first element to the matrix' LAST VALID element (range check: rows past the matrix read as zeros).
  // Not "rows x ld": the tile is fetched without a column predicate, and behind the last row's valid columns there may be nothing
  // mapped (a column-slice view of a packed buffer that ends on its allocation's last page).
  const long recA = A_KR ? ((long)(K - 1) * P.lda + (M - m0)) * 2 : ((long)(M - m0 - 1) * P.lda + K) * 2;
  const long recB = B_KR ? ((long)(K - 1) * P.ldb + (N - n0)) * 2 : ((long)(N - n0 - 1) * P.ldb + K) * 2;
  const int kstepA = A_KR ? BK * P.lda : BK, kstepB = B_KR ? BK * P.ldb : BK;     // elements per k-step
  char* const my_pieces = smem + wave * 1024;

  auto issue_piece = [&](int kt, int i, bool empty = false) {   // piece i of this wave for stage kt (wave-uniform control); empty: MMF_G6_DBG & 8
    char* st = my_pieces + (kt % NS) * STAGE;
    if (i < PPO) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned short*>(Ab + (size_t)kt * kstepA), 0, empty ? 0 : (int)(recA - (long)kt * kstepA * 2), 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(st + i * 4096), 16, voff[i], 0, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned short*>(Bb + (size_t)kt * kstepB), 0, empty ? 0 : (int)(recB - (long)kt * kstepB * 2), 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(st + TILE + (i - PPO) * 4096), 16, voff[i], 0, 0, 0);
    }
  };

  // In the loop a piece is TWO instructions (M0, buffer_load ... lds) from inline asm against two descriptors that live in SGPR
  // quads and are advanced by one k-step per stage (three scalar instructions each).  A wave issues at most one instruction per four
  // cycles, so a wave that owns its SIMD has eight issue slots per 32-cycle MFMA; the builtin form rebuilt the descriptor quad for
  // every piece (five to six instructions per piece) and the slots behind the hand-over barrier — MFMA, fragment read, piece — ran
  // out: the matrix pipe waited ~35 cycles per piece (ablations in DESIGN.md section 5, round 3).
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  auto mkdesc = [](const unsigned short* p, long rec) {
    const unsigned long long a = (unsigned long long)(uintptr_t)p;
    return i32x4_t{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)rec, 0x00020000};
  };
  auto advance = [](i32x4_t& d, int bytes) {
    const unsigned long long a = (((unsigned long long)(unsigned)d[1] << 32) | (unsigned)d[0]) + (unsigned long long)bytes;
    d[0] = (int)(unsigned)a; d[1] = (int)(unsigned)(a >> 32); d[2] -= bytes;
  };
  // descriptors of the stage the loop fetches next; stage NS - 1 is the last one the prologue fetched
  i32x4_t dA = mkdesc(Ab + (size_t)(NS - 1) * kstepA, recA - (long)(NS - 1) * kstepA * 2);
  i32x4_t dB = mkdesc(Bb + (size_t)(NS - 1) * kstepB, recB - (long)(NS - 1) * kstepB * 2);
  const unsigned lds_pieces = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)my_pieces;
  auto hot_piece = [&](auto ic, const unsigned ring_base) {   // piece I of this wave into the ring slot at LDS address ring_base (+ wave)
    constexpr int I = decltype(ic)::value, OFF = I < PPO ? I * 4096 : TILE + (I - PPO) * 4096;
    if constexpr (MMF_G6_DBG & 16) {                           // ablation: the piece's three issue slots without the piece
      asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    if (I < PPO) asm volatile("s_add_i32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ring_base), "v"(voff[I]), "s"(dA), "n"(OFF) : "memory");
    else         asm volatile("s_add_i32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ring_base), "v"(voff[I]), "s"(dB), "n"(OFF) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- fragment addressing ------------------------------------------------------------------------------------------------------
  unsigned la0, la1, lb0, lb1;
  Frag4<A_KR>::lane_parts(WA, lane, la0, la1);
  Frag4<B_KR>::lane_parts(WB, lane, lb0, lb1);
  const unsigned smem_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  // the wave's quadrant inside an operand tile: 128 tile rows (KC) or 128 tile columns (KR) = D blocks 4 w .. 4 w + 3
  const unsigned qa = (unsigned)(A_KR ? 512 * 4 * wm : (WA / 32) * 2048 * 4 * wm);
  const unsigned qb = (unsigned)(B_KR ? 512 * 4 * wn : (WB / 32) * 2048 * 4 * wn);
  la0 += smem_base + qa; la1 += smem_base + qa;
  lb0 += smem_base + TILE + qb; lb1 += smem_base + TILE + qb;

  // [tn][tm]; never zeroed: the tile's first sixteen MFMAs take a zero constant as their C operand (256 v_accvgpr_write = 0.5 us per
  // tile for a wave that owns its SIMD)
  f32x16_t acc[4][4];
  // the tile column n0 == 0 carries the bias gradient; its four waves share the extra MFMAs: wave (wm, wn) sums blocks 2 wn, 2 wn + 1 of
  // its 128 m-columns (two extra MFMAs per substep in every wave instead of four in two of them: the tile's barrier waits for the slowest)
  const bool do_colsum = A_KR && (args.epi & MMF_EPI_COLSUM_A) && n0 == 0;
  // wgrad's fused bias gradient (column sums of the m-operand).  The 256 accumulator registers fill the AGPR half exactly: a
  // seventeenth compiler-visible MFMA accumulator made hipcc keep one tile in VGPRs and shuttle it through AGPRs around every
  // use (16 v_accvgpr_write + s_nop 11 + 16 v_accvgpr_read per MFMA: TN ran 40 % slower than NT).  So the sums live in ONE
  // VGPR-resident tile behind inline asm (the VGPR form of the same MFMA): the first operand of block tm's MFMA is a selector
  // fragment — row tm all ones, every other row zero — so row tm of the tile collects block tm's column sums.
  const bool edge_m = m0 + BM > M;
  bool colok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) colok[j] = m0 + 128 * wm + 32 * j + (lane & 31) < M;
  f32x16_t csum;
#pragma unroll
  for (int e = 0; e < 16; ++e) csum[e] = 0.f;
  u32x4_t sel[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned w = ((lane & 31) == j) ? 0x3f803f80u : 0u;
    sel[j] = u32x4_t{w, w, w, w};
  }

  // ---- prologue: fill the ring -----------------------------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < KT) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(s, i);
    }
  if (KT >= NS) vm_wait<PPW * (NS - 1)>(); else vm_wait<0>();
  __builtin_amdgcn_s_barrier();
  Frag4<A_KR> fa[2];
  Frag4<B_KR> fb[2];
  fa[0].template issue<WA, 0, 0>(la0, la1);
  fb[0].template issue<WB, 0, 0>(lb0, lb1);
  frag_wait(fa[0], fb[0]);

  // MFMA i of a substep, (tm, tn) = (i / 4, i % 4); the bias-gradient MFMA rides behind each row's last
  auto mf = [&](const Frag4<A_KR>& a, const Frag4<B_KR>& b, int i, bool first = false) {
    const int tm = i >> 2, tn = i & 3;
    const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.get(tn), a.get(tm), first ? zero : acc[tn][tm], 0, 0, 0);
    if (A_KR && do_colsum && tn == 3 && (tm >> 1) == wn) {
      u32x4_t fm = __builtin_bit_cast(u32x4_t, a.get(tm));
      // columns past M hold whatever lies behind the row in memory; with the selector a NaN there would reach the valid
      // column of the same lane in the other blocks' rows (0 x NaN), so a boundary tile zeroes them first
      if (edge_m && !colok[tm]) fm = u32x4_t{0u, 0u, 0u, 0u};
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(csum) : "v"(sel[tm]), "v"(fm));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // read unit u (0..3: m-operand fragments, 4..7: n-operand fragments) of substep G of the stage at byte offset so
#define MMF_G6_READ(dstA, dstB, G, u, so)                                                              \
  do {                                                                                                 \
    if ((u) == 0) dstA.template issue1<WA, G, 0, 0>(la0 + (so), la1 + (so));                            \
    if ((u) == 1) dstA.template issue1<WA, G, 0, 1>(la0 + (so), la1 + (so));                            \
    if ((u) == 2) dstA.template issue1<WA, G, 0, 2>(la0 + (so), la1 + (so));                            \
    if ((u) == 3) dstA.template issue1<WA, G, 0, 3>(la0 + (so), la1 + (so));                            \
    if ((u) == 4) dstB.template issue1<WB, G, 0, 0>(lb0 + (so), lb1 + (so));                            \
    if ((u) == 5) dstB.template issue1<WB, G, 0, 1>(lb0 + (so), lb1 + (so));                            \
    if ((u) == 6) dstB.template issue1<WB, G, 0, 2>(lb0 + (so), lb1 + (so));                            \
    if ((u) == 7) dstB.template issue1<WB, G, 0, 3>(lb0 + (so), lb1 + (so));                            \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
  // LDS-DMA schedule.  Stage kt's ring slot is free from its hand-over barrier on and is refilled with stage kt + NS: four pieces
  // behind the last four MFMAs of stage kt ("late"), the others behind MFMAs 8, 10, 12, 14 of the not-last substeps of stage
  // kt + 1 ("early") — slots that carry no fragment read.
  static_assert(PPW == 4 + 4 * (NG - 1), "four late pieces + four per not-last substep");
  // a substep that is not the stage's last: MFMA i (i < 8) is followed by one read of the next substep's fragments, the other
  // eight MFMAs cover the reads' latency
#define MMF_G6_SUBSTEP(cur, nxt, G)                                                                    \
  do {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                   \
      mf(fa[cur], fb[cur], i, FIRST && (G) == 1);                                                      \
      if (i < 8 && !(MMF_G6_DBG & 4)) MMF_G6_READ(fa[nxt], fb[nxt], G, i, so);                        \
      if (i == 8 && !(MMF_G6_DBG & 1)) { if (early) hot_piece(std::integral_constant<int, 4 * (G)>{}, ring_prev); }      \
      if (i == 10 && !(MMF_G6_DBG & 1)) { if (early) hot_piece(std::integral_constant<int, 4 * (G) + 1>{}, ring_prev); } \
      if (i == 12 && !(MMF_G6_DBG & 1)) { if (early) hot_piece(std::integral_constant<int, 4 * (G) + 2>{}, ring_prev); } \
      if (i == 14 && !(MMF_G6_DBG & 1)) { if (early) hot_piece(std::integral_constant<int, 4 * (G) + 3>{}, ring_prev); } \
    }                                                                                                  \
    if (!(MMF_G6_DBG & 4)) frag_wait(fa[nxt], fb[nxt]);                                                \
  } while (0)

  using T = std::true_type;
  using F = std::false_type;
  // One stage.  NEXT: stage kt + 1 exists (hand-over + its first fragments).  STEADY: both refills (stage kt - 1 + NS early,
  // stage kt + NS late) exist and no condition is evaluated — a uniform branch costs a wave that owns its SIMD 10-20 idle
  // matrix-pipe cycles; the few stages at the ends of the sweep take the checked form.
  auto stage = [&](auto next_c, auto steady_c, auto first_c, const int kt, const int ahead, const bool e, const bool l) {
    constexpr bool NEXT = decltype(next_c)::value, STEADY = decltype(steady_c)::value, FIRST = decltype(first_c)::value;
    const bool early = STEADY || e, late = STEADY || l;
    const unsigned so = (unsigned)((kt % NS) * STAGE);
    const unsigned ring_cur = lds_pieces + so, ring_prev = lds_pieces + (unsigned)(((kt + NS - 1) % NS) * STAGE);
    if constexpr (NG == 4) {
      MMF_G6_SUBSTEP(0, 1, 1);
      MMF_G6_SUBSTEP(1, 0, 2);
      MMF_G6_SUBSTEP(0, 1, 3);
    } else {
      MMF_G6_SUBSTEP(0, 1, 1);
    }
    // last substep: four MFMAs, the stage hand-over, then the rest: MFMAs 4..11 each followed by one read of stage kt + 1's first
    // fragments, MFMAs 12..15 by the late pieces
    const unsigned sn = (unsigned)(((kt + 1) % NS) * STAGE);
#pragma unroll
    for (int i = 0; i < 4; ++i) mf(fa[1], fb[1], i);
    if constexpr (NEXT) {
      if constexpr (!(MMF_G6_DBG & 2)) {
        if constexpr (STEADY) {
          vm_wait<PPW * (NS - 2)>();                           // the NS - 2 younger stages stay in flight
        } else {                                               // ring running dry: exactly the stages issued after kt + 1
          if (ahead >= 3) vm_wait<PPW * 3>(); else if (ahead == 2) vm_wait<PPW * 2>(); else if (ahead == 1) vm_wait<PPW>(); else vm_wait<0>();
        }
        __builtin_amdgcn_s_barrier();                          // stage kt + 1 landed for everyone; nobody reads stage kt any more
      }
      if constexpr (!(MMF_G6_DBG & 32)) {                      // (ablation 32: the ring is refilled from the same L2-resident stage)
        advance(dA, kstepA * 2);                               // the descriptors now address stage kt + NS
        advance(dB, kstepB * 2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 4; i < 16; ++i) {
      mf(fa[1], fb[1], i);
      if constexpr (NEXT && !(MMF_G6_DBG & 4)) { if (i < 12) MMF_G6_READ(fa[0], fb[0], 0, i - 4, sn); }
      if (i == 12 && !(MMF_G6_DBG & 1)) { if (late) hot_piece(std::integral_constant<int, 0>{}, ring_cur); }
      if (i == 13 && !(MMF_G6_DBG & 1)) { if (late) hot_piece(std::integral_constant<int, 1>{}, ring_cur); }
      if (i == 14 && !(MMF_G6_DBG & 1)) { if (late) hot_piece(std::integral_constant<int, 2>{}, ring_cur); }
      if (i == 15 && !(MMF_G6_DBG & 1)) { if (late) hot_piece(std::integral_constant<int, 3>{}, ring_cur); }
    }
    if constexpr (NEXT && !(MMF_G6_DBG & 4)) frag_wait(fa[0], fb[0]);
  };
  {
    int kt = 0;
    if (KT > 1) {
      stage(T{}, F{}, T{}, 0, min(NS - 2, KT - 2), false, NS < KT);
      for (kt = 1; kt + NS < KT; ++kt) stage(T{}, T{}, F{}, kt, NS - 2, true, true);
      for (; kt + 1 < KT; ++kt) stage(T{}, F{}, F{}, kt, min(NS - 2, KT - 2 - kt), kt - 1 + NS < KT, kt + NS < KT);
      stage(F{}, F{}, F{}, kt, 0, false, false);
    } else {
      stage(F{}, F{}, T{}, 0, 0, false, false);
    }
  }
#undef MMF_G6_SUBSTEP
#undef MMF_G6_READ

  if (A_KR && do_colsum) {              // row tm of the tile (register tm of lanes 0..31) = the sums of block tm's 32 columns
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(csum));   // the asm MFMAs are invisible to the hazard recognizer
    if (lane < 32) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + 128 * wm + 32 * tm + lane;
        if ((tm >> 1) == wn && m < M) atomicAdd(const_cast<float*>(P.bias) + m, csum[tm]);
      }
    }
  }
  if constexpr (MMF_G6_DBG & 64) {                            // ablation: no epilogue (one word per lane keeps the accumulators alive)
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep += acc[i][j][0];
    if (keep == 12345.678f) static_cast<float*>(P.C)[lane] = keep;
    return;
  }
  tile_epilogue<OUT_F32>(args, P, pi, m0 + 128 * wm, n0 + 128 * wn, acc, lane, args.epi);
}

template <bool A_KR, bool B_KR, int BK, int NS, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 1)
void gemm6_grouped_kernel(const GemmArgs args, const int total_tiles) {
  __shared__ __attribute__((aligned(1024))) char smem[NS * 2 * 256 * BK * 2];
  gemm6_body<A_KR, B_KR, BK, NS, OUT_F32>(args, total_tiles, smem);
}

template <bool A_KR, bool B_KR, int BK, int NS>
void launch(const GemmArgs& a, int total, int out_f32, hipStream_t s) {
  if (out_f32) hipLaunchKernelGGL((gemm6_grouped_kernel<A_KR, B_KR, BK, NS, true>), dim3(total), dim3(NTHREADS), 0, s, a, total);
  else         hipLaunchKernelGGL((gemm6_grouped_kernel<A_KR, B_KR, BK, NS, false>), dim3(total), dim3(NTHREADS), 0, s, a, total);
}
}  // namespace

// whether gemm6 can take this launch (gemm.hip asks before selecting it): every K a multiple of the stage depth
bool mmf_gemm6_supports(const mmf_gemm_problem* problems, int num_problems, int layout) {
  if (layout == MMF_GEMM_TN) return true;                    // both operands KR: the K tail is zero-filled by the range check
  const int bk = 32;
  for (int i = 0; i < num_problems; ++i)
    if (problems[i].K % bk) return false;
  return true;
}
bool mmf_gemm6_supports_epi(int epilogue, int out_f32) { return !(out_f32 && (epilogue & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX))); }

// called by mmf_gemm_grouped (gemm.hip) after validation
int mmf_gemm6_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
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
    const size_t a_bytes = (size_t)(layout == MMF_GEMM_TN ? p.K : p.M) * p.lda * 2;
    const size_t b_bytes = (size_t)(layout == MMF_GEMM_NT ? p.N : p.K) * p.ldb * 2;
    if (a_bytes >= 0x7fffffffull || b_bytes >= 0x7fffffffull)
      MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped[%d]: operand larger than 2 GiB", i);
    if (layout != MMF_GEMM_TN && p.K % 32)
      MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped[%d]: the one-wave-per-SIMD kernel needs K %% 32 == 0 for NT / NN (K = %d)", i, p.K);
    if (out_f32 && (epilogue & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)))
      MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: the one-wave-per-SIMD kernel has no aux epilogue with f32 output");
    a.tile_start[i] = total;
    total += ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    a.p[i] = p;
  }
  a.tile_start[num_problems] = total;
  static const int cfg = [] { const char* e = getenv("MMF_GEMM6_CFG"); return e ? atoi(e) : 0; }();   // 0: 32 x 4 stages, 1: 32 x 5, 2: 64 x 2
  bool k64 = true;
  for (int i = 0; i < num_problems; ++i) k64 = k64 && (problems[i].K % 64 == 0 || layout == MMF_GEMM_TN);
  const int c = (cfg == 2 && !k64) ? 0 : cfg;
  switch (layout) {
    case MMF_GEMM_NT:
      if (c == 1) launch<false, false, 32, 5>(a, total, out_f32, s); else if (c == 2) launch<false, false, 64, 2>(a, total, out_f32, s);
      else launch<false, false, 32, 4>(a, total, out_f32, s);
      break;
    case MMF_GEMM_NN:
      if (c == 1) launch<false, true, 32, 5>(a, total, out_f32, s); else if (c == 2) launch<false, true, 64, 2>(a, total, out_f32, s);
      else launch<false, true, 32, 4>(a, total, out_f32, s);
      break;
    default:
      if (c == 1) launch<true, true, 32, 5>(a, total, out_f32, s); else if (c == 2) launch<true, true, 64, 2>(a, total, out_f32, s);
      else launch<true, true, 32, 4>(a, total, out_f32, s);
      break;
  }
  MMF_CHECK_LAUNCH("mmf_gemm_grouped(v6)");
  return MMF_OK;
}
