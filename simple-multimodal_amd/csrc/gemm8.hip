// Grouped bf16 GEMM, persistent one-wave-per-SIMD form on v_mfma_f32_16x16x32_bf16 (round 4, late): gemm7.hip's persistent workgroups, ring
// and drain with the OTHER MFMA shape in the k-loop.
//
// Why.  Under matrix load the part runs at the socket power limit, not at its clock ceiling (profiles/r04_clock_under_load.txt: 1.62 GHz of
// 2.4), and the clock it holds depends on the MFMA shape: bare register-operand loops deliver 2,047 TF with 16x16x32 against 1,790-1,823 with
// 32x32x16 on the same box (tools/mfma_shape_bench.hip; MI355X_MICROARCH.md 'DVFS give-back' (7): 1.12-1.15x), at equal cycles per FLOP and
// equal LDS bytes per FLOP (a 128 x 128 wave tile needs 16 one-KiB fragments per 32 k-columns either way).
// What changes against gemm7.hip:
//   * a wave's 128 x 128 quadrant is 8 x 8 accumulator tiles of 16 x 16 (four registers each, the same 256 AGPRs); a stage (32 k-columns)
//     is ONE k-step of 64 MFMAs; D is n x m as before (a lane holds four consecutive n of one row m);
//   * fragments are 16 rows x 32 k.  The k-contiguous image gets another swizzle — chunk ^ 2 ((row >> 3) & 1) instead of chunk ^ ((row >> 2) & 3):
//     the 16-row read is serviced in the hardware's four 16-lane groups {0-3, 12-15, 20-27}, ... and the old image made it two-way
//     (simulated with tools/lds_image_check.py's bank model: 8 cycles -> 4); the k-major image ([32][256], transposed reads) is unchanged;
//   * all sixteen fragments of stage g + 1 are read during stage g into a second register set, so the stage hand-over stands at the TOP of a
//     stage (after four MFMAs): behind it stage g + 1 is readable and stage g's own slot — read a stage ago — is refilled at once, NS stages
//     ahead instead of NS - 1.  The two register sets alternate, so the k-loop is unrolled by two and K must be a multiple of 64;
//   * flag sets without a bias start a tile's accumulators from a zero C operand in a peeled first stage; with a bias, from 64 MFMAs of the
//     bias' three-way bf16 split against ones (gemm7: 16);
//   * the drain writes each accumulator tile's four packed values as one 8-byte LDS store; whole-row reads and stores as in gemm7.
// NT and NN, bf16 output, K % 64 == 0 and K >= 6 * 32; everything else stays on gemm7 / gemm6 (gemm.hip).
#include <algorithm>
#include "gemm6_parts.h"

namespace {

constexpr int BK = 32, NS = 4;
constexpr int TILE = 256 * BK * 2, STAGE = 2 * TILE, PPO = BK / 8, PPW = 2 * PPO;   // 16 KiB per operand tile, 8 pieces per wave and stage

typedef int i32x4_t __attribute__((ext_vector_type(4)));

// compile-time loop: f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N - 1>{}) — immediates of the asm reads need constants
template <int... I, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

// what the fetch side needs of the NEXT tile (wave-uniform): first element of each operand tile, bytes from there to the operand's
// last valid element (< 2 GiB: host), bytes per stage of the n-operand, stages
struct TileSrc {
  const unsigned short* Ab;
  const unsigned short* Bb;
  int recA, recB, stepB, KT;
};
// what the compute / store side needs
struct TileDst {
  int pi, m0, n0;
};

// A wave-uniform value the vector ALU produced (integer division has no scalar form), handed to the scalar side.  As an inline-asm
// statement on purpose: the builtin is folded away wherever the compiler can prove its operand uniform, and the loop-carried
// descriptor chain that depends on it then lands in VECTOR registers — the "s" operands of the LDS-DMA statements print as VGPR
// quads and the assembler rejects them (seen with this kernel's tile loop; gemm6 has no such loop).
__device__ __forceinline__ int to_sgpr(int x) {
  int r;
  // wait states INSIDE the string (hipcc pads nothing around an asm statement): one between a vector write of x and the
  // readfirstlane (without it m0 / n0 came back stale: a memory fault on the first launch), five before a memory instruction may
  // read the scalar the vector ALU wrote
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(x));
  return r;
}

// per-lane source offset of 1-KiB piece p (the inverse of the image).  k-major operands: gemm6_parts.h's image.  k-contiguous operands
// ([256 rows][32 k]): 8-row subtiles of 512 B, 16-byte chunk ch of row `row` in slot ch ^ 2 ((row >> 3) & 1) of its 64-byte row
template <bool KR>
__device__ __forceinline__ unsigned piece_voff16(int p, int ld, int lane) {
  if constexpr (KR) return piece_voff<true, BK>(p, ld, lane);
  const int st = 2 * p + (lane >> 5), w = lane & 31;
  const int row = 8 * st + (w >> 2), ch = (w & 3) ^ (2 * (st & 1));
  return (unsigned)(row * ld * 2 + ch * 16);
}

template <bool B_KR>
__device__ __forceinline__ void locate_tile(const GemmArgs& args, const int total_tiles, const int orig, TileSrc& s, TileDst& d, int& lda, int& ldb) {
  const int bid = to_sgpr(mmf_xcd_tile(orig, total_tiles, args.xcd_granule));
  int pi = 0;
  while (pi + 1 < args.nprob && bid >= args.tile_start[pi + 1]) ++pi;
  const mmf_gemm_problem& P = args.p[pi];
  int m0, n0;
  tile_origin(P, bid - args.tile_start[pi], m0, n0);
  m0 = to_sgpr(m0);
  n0 = to_sgpr(n0);
  d.pi = pi; d.m0 = m0; d.n0 = n0;
  lda = P.lda; ldb = P.ldb;
  s.Ab = static_cast<const unsigned short*>(P.A) + (size_t)m0 * P.lda;
  s.Bb = static_cast<const unsigned short*>(P.B) + (B_KR ? (size_t)n0 : (size_t)n0 * P.ldb);
  s.recA = (int)(((long)(P.M - m0 - 1) * P.lda + P.K) * 2);
  s.recB = (int)((B_KR ? ((long)(P.K - 1) * P.ldb + (P.N - n0)) : ((long)(P.N - n0 - 1) * P.ldb + P.K)) * 2);
  s.stepB = (B_KR ? BK * P.ldb : BK) * 2;
  s.KT = P.K / BK;
}

// ---- the tile's outputs through LDS -----------------------------------------------------------------------------------------------
// The accumulator layout puts a ROW on each lane: stored from there, one 16-byte store instruction touches 32 rows x 32 B (32 cache
// lines, a quarter of each), and the CU's store path takes ~64 cycles for it: 17.5 B/clk/CU, 7,500 cycles for the 128 KiB of a tile
// (tools/store_path_bench.hip, shape 0); the same bytes as whole rows — 4 rows x 256 B per instruction — leave at 51 B/clk/CU (shape
// 1: 2,600 cycles).  The aux operand of the residual / mask epilogues came in the same row-per-lane shape with 8-byte loads, each behind
// its own predicate branch: ~10 us per tile (out-projection 34 us vs 24 us for the plain epilogue on the same problem).  So each wave
// owns an 8-KiB LDS region [32 rows][256 B] (16-byte chunk c of row r at r * 256 + ((c ^ (r & 15)) << 4): both access shapes are
// conflict-free or two-way) and, per 32-row block tm of its quadrant,
//   aux: eight 16-byte whole-row loads (issued two blocks ahead, range-checked buffer loads: no branches) -> region -> read back in the
//        accumulator layout;
//   out: finished and packed in the accumulator layout -> region (in place) -> read back as whole rows -> eight range-checked 16-byte
//        stores of 4 rows x 256 B.
// LDS operations of one wave execute in order, so the region needs no barrier and no wait between a write and the read behind it.
// CT: the epilogue's flag set (compile time; alpha = 1, no dropout).  The bias is not added here: it is what the tile's accumulators
// START from (bias_init below).
template <int CT>
__device__ __forceinline__ void drain_tile(const GemmArgs& args, const int pi, const mmf_gemm_problem& P, const int mb, const int nb,
                                           f32x4_t (&acc)[8][8], char* region, const int lane) {
  constexpr bool AUX = (CT & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) != 0;
  constexpr bool DROP = (CT & MMF_EPI_DROPOUT) != 0;
  const unsigned drop_key = DROP ? mmf_rng_key(*args.rng_state, args.site, (unsigned)args.orig[pi]) : 0u;
  const float drop_scale = DROP ? 1.f / (1.f - (float)args.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  const float alpha = (CT & MMF_EPI_MASK_AUX) ? args.alpha : 1.f;
  // accumulator tile [tn][tm]: lane (j = l & 15, g4 = l >> 4) holds C[m = 16 tm + j][n = 16 tn + 4 g4 + 0..3]: four consecutive n = 8 bytes of
  // bf16 at row rr = 16 s + j of the 32-row block (s = tm & 1), 16-byte chunk c = 2 tn + (g4 >> 1), half g4 & 1.  Region layout as gemm7:
  // chunk c of row r at r * 256 + ((c ^ (r & 15)) << 4) — a 16-lane group covers 16 rows = 16 different slots, the four groups two chunks
  // x two halves: conflict-free.
  const int j = lane & 15, g4 = lane >> 4, q = lane >> 4, cq = lane & 15;
  auto acc_at = [&](int s, int tn) {
    const unsigned rr = (unsigned)(16 * s + j), c = (unsigned)(2 * tn + (g4 >> 1));
    return region + rr * 256 + ((c ^ (rr & 15)) << 4) + 8 * (g4 & 1);
  };
  const unsigned rd_base = (unsigned)(q * 256 + ((cq ^ q) << 4));
  auto row_at = [&](int it) { return region + it * 1024 + (rd_base ^ (unsigned)((it & 3) << 6)); };
  const bool col_ok = nb + 8 * cq < P.N;
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, (int)((((long)P.M - 1) * P.ldc + P.N) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(AUX ? P.aux : P.C), 0,
                                                                       AUX ? (int)((((long)P.M - 1) * P.ldaux + P.N) * 2) : 0, 0x00020000);
  const unsigned c_off0 = col_ok ? (unsigned)(((long)(mb + q) * P.ldc + nb + 8 * cq) * 2) : 0x80000000u;
  const unsigned a_off0 = col_ok ? (unsigned)(((long)(mb + q) * P.ldaux + nb + 8 * cq) * 2) : 0x80000000u;
  const unsigned c_row4 = (unsigned)(4 * P.ldc * 2), a_row4 = (unsigned)(4 * P.ldaux * 2);

  u32x4_t auxr[2][8];
  auto load_aux = [&](int tb, int slot) {
#pragma unroll
    for (int it = 0; it < 8; ++it)
      auxr[slot][it] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ars, a_off0 + (unsigned)(8 * tb + it) * a_row4, 0, 0));
  };
  if constexpr (AUX) { load_aux(0, 0); load_aux(1, 1); }
#pragma unroll
  for (int tb = 0; tb < 4; ++tb) {                             // 32-row block of the quadrant
    u32x2_t axv[2][8];
    if constexpr (AUX) {
#pragma unroll
      for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4_t*>(row_at(it)) = auxr[tb & 1][it];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) axv[s][tn] = *reinterpret_cast<const u32x2_t*>(acc_at(s, tn));
      __builtin_amdgcn_sched_barrier(0);
      if (tb + 2 < 4) load_aux(tb + 2, tb & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int tm = 2 * tb + s;
#pragma unroll
      for (int tn = 0; tn < 8; ++tn) {
        asm volatile("" : "+a"(acc[tn][tm]));                  // (reads of the accumulators are not hoisted to the head of the drain)
        f32x4_t v = acc[tn][tm];
        if constexpr (CT & MMF_EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (DROP) {
          const unsigned idx = (unsigned)(mb + 16 * tm + j) * (unsigned)P.N + (unsigned)(nb + 16 * tn + 4 * g4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mmf_keep(drop_key, idx + e, args.drop_thresh) ? v[e] * drop_scale : 0.f;
        }
        if constexpr (AUX) {
          const u32x2_t a = axv[s][tn];
          const float a0 = bf16lo(a[0]), a1 = bf16hi(a[0]), a2 = bf16lo(a[1]), a3 = bf16hi(a[1]);
          if constexpr (CT & MMF_EPI_MASK_AUX) {
            v[0] = a0 > 0.f ? v[0] * alpha : 0.f; v[1] = a1 > 0.f ? v[1] * alpha : 0.f;
            v[2] = a2 > 0.f ? v[2] * alpha : 0.f; v[3] = a3 > 0.f ? v[3] * alpha : 0.f;
          } else {
            v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
          }
        }
        *reinterpret_cast<u32x2_t*>(acc_at(s, tn)) = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    u32x4_t w[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) w[it] = *reinterpret_cast<const u32x4_t*>(row_at(it));
#pragma unroll
    for (int it = 0; it < 8; ++it) __builtin_amdgcn_raw_buffer_store_b128(w[it], crs, c_off0 + (unsigned)(8 * tb + it) * c_row4, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// CT: the launch's epilogue flag set, compile time (alpha = 1, no dropout, outputs in 8-column granularity: the host sends
// everything else to gemm6) — one drain form per kernel.
//
// ONE stage body serves every stage of every tile of the workgroup (the first form of this kernel instantiated the stage per role —
// tile's first, steady, descriptor switch, ring running dry, last — as gemm6 does; around the tile loop hipcc then copied the 256
// accumulator registers between the roles' code through vector registers and scratch).  What made the roles differ is removed:
//   * the accumulators never start from a zero C operand: sixteen MFMAs in front of the tile's first stage set them to the bias
//     (or to zero, from zero fragments);
//   * the ring never "runs dry": behind the workgroup's last tile the descriptors get a range of ZERO bytes, so the refills stay in
//     the instruction stream (and in the hand-overs' counts) but touch no memory — the range check answers them with zeros;
//   * the ring fill fetches NS - 1 stages and the late half of the NS-th, so the very first stage already finds its early pieces
//     to issue.
template <bool B_KR, int CT>
__device__ __forceinline__ void gemm8_body(const GemmArgs& args, const int total_tiles, char* smem) {
  constexpr bool A_KR = false;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                   // this wave's 128 x 128 quadrant
  const int nwg = gridDim.x;
  // The walk: round r hands ids r * nwg .. to the workgroups in ascending order when r is even and in DESCENDING order when it is odd.
  // The host sorts the problems by K, longest first, so ids run from the longest tiles to the shortest; the snake gives whoever got the
  // longest tiles of one round the shortest of the next (a static longest-processing-time deal).  With every round ascending, the in-projection
  // dgrad launch of the text group (192 tiles of 24 k-steps, 81 of 48: 273 tiles on 256 CUs) gave its 17 second-round tiles — long ones —
  // to workgroups 0 .. 16, four of which already held long tiles: 96 k-steps where 48 suffice (93 us in the step, 382 TF).
  const int wid = blockIdx.x;
  auto walk = [&](int round) { return round * nwg + ((round & 1) ? nwg - 1 - wid : wid); };
  int round = 0;
  int orig = walk(0);
  TileSrc ns;
  TileDst cd, nd;
  int KT, lda, ldb;
  locate_tile<B_KR>(args, total_tiles, orig, ns, cd, lda, ldb);
  KT = ns.KT;

  // ---- LDS-DMA: this wave's PPW pieces of a stage, per-lane source offsets for the tile being FETCHED -----------------------------
  unsigned voff[PPW], voffn[PPW];
  auto set_voff = [&](unsigned (&v)[PPW], int la, int lb) {
#pragma unroll
    for (int i = 0; i < PPO; ++i) {
      v[i] = piece_voff16<A_KR>(wave + 4 * i, la, lane);
      v[PPO + i] = piece_voff16<B_KR>(wave + 4 * i, lb, lane);
    }
  };
  set_voff(voff, lda, ldb);
  char* const my_pieces = smem + wave * 1024;

  // The descriptors of the stage being fetched, as SCALARS (address low / high word, record bytes): carried across the tile loop as
  // <4 x i32> values they were given vector registers (their SGPR words copied into a VGPR quad at the loop header, which the "s"
  // operands of the LDS-DMA statements then printed: assembler errors); the quads are put together where they are used.
  struct Desc { int lo, hi, rec; };
  auto mkdesc = [](const unsigned short* p, int rec) {
    const unsigned long long a = (unsigned long long)(uintptr_t)p;
    return Desc{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), rec};
  };
  auto advance = [](Desc& d, int bytes) {                    // one stage on; a range that is used up stays empty
    const unsigned long long a = (((unsigned long long)(unsigned)d.hi << 32) | (unsigned)d.lo) + (unsigned long long)bytes;
    d.lo = (int)(unsigned)a; d.hi = (int)(unsigned)(a >> 32);
    // (scalar asm: hipcc turns max(rec - bytes, 0) into a VECTOR saturating subtract, and a vector-computed descriptor word reaches the
    // "s" operand of the LDS-DMA statement as a VGPR — it inserts no readfirstlane there)
    int r;
    asm("s_sub_i32 %0, %1, %2\n\ts_max_i32 %0, %0, 0" : "=s"(r) : "s"(d.rec), "s"(bytes) : "scc");
    d.rec = r;
  };
  Desc dA = mkdesc(ns.Ab, ns.recA);
  Desc dB = mkdesc(ns.Bb, ns.recB);
  constexpr int stepA = BK * 2;
  int stepB = ns.stepB;                                      // bytes per stage of the tile being fetched
  const unsigned lds_pieces = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)my_pieces;
  auto hot_piece = [&](auto ic, const unsigned ring_base) {
    constexpr int I = decltype(ic)::value, OFF = I < PPO ? I * 4096 : TILE + (I - PPO) * 4096;
    const Desc& d = I < PPO ? dA : dB;
    const i32x4_t q = {d.lo, d.hi, d.rec, 0x00020000};
    asm volatile("s_add_i32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ring_base), "v"(voff[I]), "s"(q), "n"(OFF) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the workgroup's one ring fill: NS whole stages (the same two-instruction pieces: no LDS-DMA the compiler knows of, so the LDS
  // accesses of drain_tile are never preceded by a compiler-placed vmcnt(0)) -------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const unsigned rb = lds_pieces + (unsigned)(s * STAGE);
    hot_piece(std::integral_constant<int, 0>{}, rb); hot_piece(std::integral_constant<int, 1>{}, rb);
    hot_piece(std::integral_constant<int, 2>{}, rb); hot_piece(std::integral_constant<int, 3>{}, rb);
    hot_piece(std::integral_constant<int, 4>{}, rb); hot_piece(std::integral_constant<int, 5>{}, rb);
    hot_piece(std::integral_constant<int, 6>{}, rb); hot_piece(std::integral_constant<int, 7>{}, rb);
    advance(dA, stepA); advance(dB, stepB);                  // afterwards: stage NS, the next one to fetch
  }

  bool has_next = walk(1) < total_tiles;
  nd = cd;
  if (has_next) {
    int la, lb;
    locate_tile<B_KR>(args, total_tiles, walk(1), ns, nd, la, lb);
    set_voff(voffn, la, lb);
  } else {
    ns.recA = 0; ns.recB = 0; ns.stepB = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) voffn[i] = voff[i];
  }

  // ---- fragment addressing (tile-independent) ----------------------------------------------------------------------------------------
  // k-contiguous operand: lane (i = l & 15, g = l >> 4) reads chunk g of row 16 blk + i: one lane part + 1024 blk.
  // k-major operand: group g = l >> 4 takes k rows 8 g .. 8 g + 7 of columns 16 jj .. 16 jj + 15 by two transposed reads
  // (lane 4 q + p: row 8 g [+ 4] + q, columns 4 p .. 4 p + 3): lane parts (by the parity of jj) + 512 (jj >> 1).
  const unsigned smem_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const unsigned lrow = (unsigned)(512 * ((lane >> 3) & 1) + 64 * (lane & 7) + 16 * ((lane >> 4) ^ (2 * ((lane >> 3) & 1))));
  const unsigned tq = (unsigned)((lane >> 2) & 3), tp = (unsigned)(lane & 3), tg = (unsigned)(lane >> 4);
  // (the image's swizzle is chunk ^ ((k >> 2) & 3) = chunk ^ (2 (g & 1) + [second read]): its bit 1 meets the column block's parity, so the
  // lane parts come in an even-jj and an odd-jj form)
  auto ltr = [&](unsigned par, unsigned hi) {
    return 4096 * tg + 64 * (tq + 4 * hi) + 32 * (par ^ (tg & 1)) + 16 * ((tp >> 1) ^ hi) + 8 * (tp & 1);
  };
  const unsigned ltlo0 = ltr(0, 0), ltlo1 = ltr(1, 0), lthi0 = ltr(0, 1), lthi1 = ltr(1, 1);
  const unsigned a_base = smem_base + (unsigned)(1024 * 8 * wm) + lrow;
  const unsigned b_base = smem_base + TILE + (B_KR ? (unsigned)(2048 * wn) : (unsigned)(1024 * 8 * wn) + lrow);
  (void)ltlo0; (void)ltlo1; (void)lthi0; (void)lthi1;

  f32x4_t acc[8][8];                  // [tn][tm]: C[m = 16 tm + (l & 15)][n = 16 tn + 4 (l >> 4) + 0..3] of the wave's quadrant

  // Two sets of the sixteen fragments of a stage: while the MFMAs run on one, the next stage's are read into the other.
  struct Frags {
    u32x4_t a[8];
    u32x4_t b[8];                     // k-major B: {lo.x, lo.y, hi.x, hi.y} of the two transposed reads
  };
  Frags F[2];
  auto geta = [&](const Frags& f, int i) { return __builtin_bit_cast(bf16x8_t, f.a[i]); };
  auto getb = [&](const Frags& f, int i) { return __builtin_bit_cast(bf16x8_t, f.b[i]); };
  // read #u of the stage at ring offset `so` into f: u < 8: A fragment u; u >= 8: B fragment u - 8 (k-major: both halves)
  auto rd = [&](Frags& f, auto uc, const unsigned so) {
    constexpr int U = decltype(uc)::value, J = U & 7;
    if constexpr (U < 8) f.a[J] = lds_read_b128<1024 * J>(a_base + so);
    else if constexpr (!B_KR) f.b[J] = lds_read_b128<1024 * J>(b_base + so);
    else {
      const u32x2_t lo_ = lds_read_tr<512 * (J >> 1)>(b_base + ((J & 1) ? ltlo1 : ltlo0) + so);
      const u32x2_t hi_ = lds_read_tr<512 * (J >> 1)>(b_base + ((J & 1) ? lthi1 : lthi0) + so);
      f.b[J] = u32x4_t{lo_[0], lo_[1], hi_[0], hi_[1]};
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto frag_tie = [](Frags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.a[4]), "+v"(f.a[5]), "+v"(f.a[6]), "+v"(f.a[7]),
                 "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]), "+v"(f.b[4]), "+v"(f.b[5]), "+v"(f.b[6]), "+v"(f.b[7]));
  };

  // Bias: with a bias the tile's accumulators START from it — 64 MFMAs of the bias' exact three-way bf16 split (8 + 8 + 8 mantissa bits)
  // against a ones fragment, D[n][m] = sum_k A[n][k] B[k][m] with A[n][0..2] = split, B[0..2][m] = 1 (lanes 0-15 carry k = 0 .. 7).
  // Lane i keeps ONE value per 16-column block tn (bias[nb + 16 tn + (l & 15)], loaded a tile ahead).  Without a bias the tile's
  // first stage takes a zero C operand.
  constexpr bool use_bias = (CT & MMF_EPI_BIAS) != 0;
  float bcur[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto load_bias = [&](float (&b)[8], const TileDst& d) {
    const mmf_gemm_problem& P = args.p[d.pi];
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.bias), 0, P.N * 4, 0x00020000);
    const unsigned off = (unsigned)((d.n0 + 128 * wn + (lane & 15)) * 4);
#pragma unroll
    for (int tn = 0; tn < 8; ++tn)                              // (the builtin returns the 32 bits as an integer)   past N: 0
      b[tn] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(brs, off + (unsigned)(16 * tn * 4), 0, 0));
  };
  auto split3 = [&](float x) {
    const bool lo = lane < 16;
    const unsigned hh = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hh);
    const unsigned mm = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mm);
    const unsigned ll = __float_as_uint(r2) & 0xffff0000u;
    return u32x4_t{lo ? ((hh >> 16) | mm) : 0u, lo ? (ll >> 16) : 0u, 0u, 0u};
  };
  auto bias_init = [&](const float (&b)[8]) {
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    const u32x4_t ow = {lane < 16 ? 0x3f803f80u : 0u, lane < 16 ? 0x00003f80u : 0u, 0u, 0u};
    u32x4_t ones[8] = {ow, ow, ow, ow, ow, ow, ow, ow};       // opaque copies: identical MFMAs would be merged and their results copied
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) asm volatile("" : "+v"(ones[tm]));
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) {
      u32x4_t bw = split3(b[tn]);
      asm volatile("" : "+v"(bw));
#pragma unroll
      for (int tm = 0; tm < 8; ++tm) {
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw), __builtin_bit_cast(bf16x8_t, ones[tm]), zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if constexpr (use_bias) load_bias(bcur, cd);

  vm_wait<PPW * (NS - 1)>();          // stage 0 landed (with the bias loads behind the pieces this waits for more: once per workgroup)
  __builtin_amdgcn_s_barrier();
  static_for<16>([&](auto uc) { rd(F[0], uc, 0u); });
  frag_tie(F[0]);

  // One stage = 64 MFMAs on fragment set CUR while the next stage's sixteen fragments are read into the other set.
  // g: the workgroup's running stage count (ring slot g % NS); kt: the stage's index in its tile.
  //   MFMAs 0..3, then the hand-over: stage g + 1 has landed for everyone, and — its fragments having been read during stage g - 1 —
  //   nobody reads slot g % NS any more: it is refilled with stage g + NS behind MFMAs 28..43; the reads of stage g + 1 follow MFMAs
  //   4..19 (k-major B: two transposed reads per fragment);
  //   behind MFMAs 44 / 45: both descriptors one stage on, or — after the tile's stage KT - NS - 1, whose refill was the tile's last —
  //   switched (with the per-lane offsets) to the next tile's first stage: scalar / vector SELECTS, no branch (gemm7.hip).
  // FIRST: a tile's first stage without a bias (C operand zero).
  auto stage = [&](auto curc, auto firstc, const unsigned g, const int kt, const int ksw) {
    constexpr int CUR = decltype(curc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    Frags& fc = F[CUR];
    Frags& fn = F[CUR ^ 1];
    unsigned long long swmask;
    const Desc nA = mkdesc(ns.Ab, ns.recA), nB = mkdesc(ns.Bb, ns.recB);
    const unsigned ring_cur = lds_pieces + (g % NS) * STAGE;
    const unsigned sn = ((g + 1) % NS) * STAGE;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    static_for<64>([&](auto ic) {
      constexpr int i = decltype(ic)::value, tn = i >> 3, tm = i & 7;
      if constexpr (FIRST) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(getb(fc, tn), geta(fc, tm), zero, 0, 0, 0);
      else                 acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(getb(fc, tn), geta(fc, tm), acc[tn][tm], 0, 0, 0);
      asm volatile("" ::"a"(acc[tn][tm]));                    // (chains the MFMA into the order of the asm statements: gemm7.hip)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i == 3) {
        vm_wait<PPW * (NS - 2)>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i >= 4 && i < 20) rd(fn, std::integral_constant<int, (i >= 4 && i < 20) ? i - 4 : 0>{}, sn);
      if constexpr (i == 28) hot_piece(std::integral_constant<int, 0>{}, ring_cur);
      if constexpr (i == 30) hot_piece(std::integral_constant<int, 1>{}, ring_cur);
      if constexpr (i == 32) hot_piece(std::integral_constant<int, 2>{}, ring_cur);
      if constexpr (i == 34) hot_piece(std::integral_constant<int, 3>{}, ring_cur);
      if constexpr (i == 36) hot_piece(std::integral_constant<int, 4>{}, ring_cur);
      if constexpr (i == 38) hot_piece(std::integral_constant<int, 5>{}, ring_cur);
      if constexpr (i == 40) hot_piece(std::integral_constant<int, 6>{}, ring_cur);
      if constexpr (i == 42) hot_piece(std::integral_constant<int, 7>{}, ring_cur);
      if constexpr (i == 44) {                                           // both descriptors one stage on (eight scalar instructions)
        advance(dA, stepA);
        advance(dB, stepB);
        asm volatile("" ::"s"(dA.lo), "s"(dA.hi), "s"(dA.rec), "s"(dB.lo), "s"(dB.hi), "s"(dB.rec));
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i == 46) {                                           // one compare, seven scalar selects, the lane mask for the vector selects
        asm volatile("s_cmp_eq_u32 %8, %16\n\t"
                     "s_cselect_b32 %0, %9, %0\n\ts_cselect_b32 %1, %10, %1\n\ts_cselect_b32 %2, %11, %2\n\t"
                     "s_cselect_b32 %3, %12, %3\n\ts_cselect_b32 %4, %13, %4\n\ts_cselect_b32 %5, %14, %5\n\t"
                     "s_cselect_b32 %6, %15, %6\n\ts_cselect_b64 %7, -1, 0"
                     : "+s"(dA.lo), "+s"(dA.hi), "+s"(dA.rec), "+s"(dB.lo), "+s"(dB.hi), "+s"(dB.rec), "+s"(stepB), "=s"(swmask)
                     : "s"(kt), "s"(nA.lo), "s"(nA.hi), "s"(nA.rec), "s"(nB.lo), "s"(nB.hi), "s"(nB.rec), "s"(ns.stepB), "s"(ksw)
                     : "scc");
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i == 48) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %4, %8\n\tv_cndmask_b32_e64 %1, %1, %5, %8\n\t"
                     "v_cndmask_b32_e64 %2, %2, %6, %8\n\tv_cndmask_b32_e64 %3, %3, %7, %8"
                     : "+v"(voff[0]), "+v"(voff[1]), "+v"(voff[2]), "+v"(voff[3])
                     : "v"(voffn[0]), "v"(voffn[1]), "v"(voffn[2]), "v"(voffn[3]), "s"(swmask));
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i == 50) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %4, %8\n\tv_cndmask_b32_e64 %1, %1, %5, %8\n\t"
                     "v_cndmask_b32_e64 %2, %2, %6, %8\n\tv_cndmask_b32_e64 %3, %3, %7, %8"
                     : "+v"(voff[4]), "+v"(voff[5]), "+v"(voff[6]), "+v"(voff[7])
                     : "v"(voffn[4]), "v"(voffn[5]), "v"(voffn[6]), "v"(voffn[7]), "s"(swmask));
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    frag_tie(fn);
  };
  using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
  using Yes = std::integral_constant<bool, true>; using No = std::integral_constant<bool, false>;

  char* const region = smem + NS * STAGE + wave * 8192;
  unsigned g = 0;
  for (;;) {
    const int ksw = KT - NS - 1;                               // >= 1 (host: KT even, >= NS + 2)
    if constexpr (use_bias) {
      bias_init(bcur);
      stage(C0{}, No{}, g, 0, ksw);
    } else {
      stage(C0{}, Yes{}, g, 0, ksw);
    }
    stage(C1{}, No{}, g + 1, 1, ksw);
    g += 2;
    for (int kt = 2; kt < KT; kt += 2, g += 2) {
      stage(C0{}, No{}, g, kt, ksw);
      stage(C1{}, No{}, g + 1, kt + 1, ksw);
    }
    // ---- the tile's outputs ----------------------------------------------------------------------------------------------------------
    const mmf_gemm_problem& P = args.p[cd.pi];
    const int mb = cd.m0 + 128 * wm, nb = cd.n0 + 128 * wn;
    if constexpr (use_bias) { if (has_next) load_bias(bcur, nd); }   // the NEXT tile's bias: the drain covers the round trip
    drain_tile<CT & ~MMF_EPI_BIAS>(args, cd.pi, P, mb, nb, acc, region, lane);
    if (!has_next) break;
    ++round;
    orig = walk(round);
    cd = nd;
    KT = ns.KT;
    has_next = walk(round + 1) < total_tiles;
    if (has_next) {
      int la, lb;
      locate_tile<B_KR>(args, total_tiles, walk(round + 1), ns, nd, la, lb);
      set_voff(voffn, la, lb);
    } else {
      ns.recA = 0; ns.recB = 0; ns.stepB = 0;
    }
  }
  vm_wait<0>();                       // the zero-range refills behind the last tile
}

template <bool B_KR, int CT>
__global__ __launch_bounds__(NTHREADS, 1)
void gemm8_persistent_kernel(const GemmArgs args, const int total_tiles) {
  // the ring and, behind it, one 8-KiB output staging region per wave: all 160 KiB of the CU
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE + 4 * 8192];
  gemm8_body<B_KR, CT>(args, total_tiles, smem);
}

// the (layout, flag set) pairs of the fusion step's NT / NN launches: in-projections (NT, bias), FFN1 (NT, bias + ReLU), out-projection
// and FFN2 (NT, bias + residual), plain NT, dgrads (NN, none), dH (NN, ReLU mask), dX (NN, residual gradient)
template <bool B_KR, int CT>
void launch8(const GemmArgs& a, int total, int grid, hipStream_t s) {
  hipLaunchKernelGGL((gemm8_persistent_kernel<B_KR, CT>), dim3(grid), dim3(NTHREADS), 0, s, a, total);
}
bool launch8_select(int layout, int eflags, const GemmArgs* a, int total, int grid, hipStream_t s) {   // a == nullptr: only ask
  if (layout == MMF_GEMM_NT) {
    if (eflags == 0)                                      { if (a) launch8<false, 0>(*a, total, grid, s); return true; }
    if (eflags == MMF_EPI_BIAS)                           { if (a) launch8<false, MMF_EPI_BIAS>(*a, total, grid, s); return true; }
    if (eflags == (MMF_EPI_BIAS | MMF_EPI_RELU))          { if (a) launch8<false, MMF_EPI_BIAS | MMF_EPI_RELU>(*a, total, grid, s); return true; }
    if (eflags == (MMF_EPI_BIAS | MMF_EPI_ADD_AUX))       { if (a) launch8<false, MMF_EPI_BIAS | MMF_EPI_ADD_AUX>(*a, total, grid, s); return true; }
    if (eflags == (MMF_EPI_BIAS | MMF_EPI_RELU | MMF_EPI_DROPOUT)) {       // FFN hidden layer in training mode
      if (a) launch8<false, MMF_EPI_BIAS | MMF_EPI_RELU | MMF_EPI_DROPOUT>(*a, total, grid, s);
      return true;
    }
  } else if (layout == MMF_GEMM_NN) {
    if (eflags == 0)                                      { if (a) launch8<true, 0>(*a, total, grid, s); return true; }
    if (eflags == MMF_EPI_MASK_AUX)                       { if (a) launch8<true, MMF_EPI_MASK_AUX>(*a, total, grid, s); return true; }
    if (eflags == MMF_EPI_ADD_AUX)                        { if (a) launch8<true, MMF_EPI_ADD_AUX>(*a, total, grid, s); return true; }
  }
  return false;
}
}  // namespace

int mmf_gemm7_persistent_wgs();        // gemm7.hip: MMF_GEMM7_WGS / mmf_gemm_set_persistent_workgroups

// whether the persistent kernel can take this launch (gemm.hip asks before selecting it)
bool mmf_gemm8_supports(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue, int out_f32, const mmf_gemm_extra* extra) {
  if (out_f32 || !launch8_select(layout, epilogue, nullptr, 0, 0, nullptr)) return false;
  if (extra && extra->alpha != 1.f && !(layout == MMF_GEMM_NN && epilogue == MMF_EPI_MASK_AUX)) return false;   // alpha rides on the mask only
  if ((epilogue & MMF_EPI_DROPOUT) && !(extra && extra->rng_state)) return false;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_gemm_problem& p = problems[i];
    if (p.K % (2 * BK) || p.K < (NS + 2) * BK || (p.N & 7) || (p.ldc & 7) || (p.aux && (p.ldaux & 7))) return false;
  }
  return true;
}

int mmf_gemm8_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s) {
  if (!mmf_gemm8_supports(problems, num_problems, layout, epilogue, out_f32, extra))
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: the persistent kernel takes NT / NN launches with bf16 output, K %% 32 == 0, K >= %d, N and the "
             "leading dimensions of C / aux multiples of 8, and the flag sets of the fusion step (alpha only with the NN ReLU mask)", (NS + 1) * BK);
  GemmArgs a;
  a.nprob = num_problems;
  a.epi = epilogue;
  a.xcd_granule = mmf_xcd_granule();
  a.alpha = extra ? extra->alpha : 1.f;
  a.drop_thresh = (extra && (epilogue & MMF_EPI_DROPOUT)) ? mmf_drop_thresh(extra->dropout_p) : 0u;
  a.site = extra ? extra->site : 0u;
  a.rng_state = extra ? reinterpret_cast<const unsigned long long*>(extra->rng_state) : nullptr;
  int total = 0;
  int order[MMF_GEMM_MAX_PROBLEMS];                            // longest reduction first (see the walk in gemm8_body)
  for (int i = 0; i < num_problems; ++i) order[i] = i;
  std::stable_sort(order, order + num_problems, [&](int x, int y) { return problems[x].K > problems[y].K; });
  for (int i = 0; i < num_problems; ++i) {
    const mmf_gemm_problem& p = problems[order[i]];
    const size_t a_bytes = (size_t)p.M * p.lda * 2;
    const size_t b_bytes = (size_t)(layout == MMF_GEMM_NT ? p.N : p.K) * p.ldb * 2;
    const size_t c_bytes = (size_t)p.M * p.ldc * 2, x_bytes = p.aux ? (size_t)p.M * p.ldaux * 2 : 0;
    if (a_bytes >= 0x7fffffffull || b_bytes >= 0x7fffffffull || c_bytes >= 0x7fffffffull || x_bytes >= 0x7fffffffull)
      MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped[%d]: operand larger than 2 GiB", i);
    a.tile_start[i] = total;
    total += ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    a.p[i] = p;
    a.orig[i] = (short)order[i];
  }
  a.tile_start[num_problems] = total;
  static const int cus = [] { int c = mmf_device_cu_count(); return c > 0 ? c : 256; }();
  const int want = mmf_gemm7_persistent_wgs() > 0 ? mmf_gemm7_persistent_wgs() : cus;
  const int grid = total < want ? total : want;
  launch8_select(layout, epilogue, &a, total, grid, s);
  MMF_CHECK_LAUNCH("mmf_gemm_grouped(v7)");
  return MMF_OK;
}
