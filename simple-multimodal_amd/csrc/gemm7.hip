// Grouped bf16 GEMM, PERSISTENT one-wave-per-SIMD form (round 4): the 256 x 256 tile / 128 x 128 wave-tile kernel of gemm6.hip with
// one workgroup per CU walking a list of tiles, its LDS ring running on ACROSS tile boundaries.
//
// Why.  A K = 768 tile of gemm6 is 2.7 us of prologue (the ring fill: 128 KiB per CU requested by all 256 CUs at once), 15.4 us of
// k-loop and 4-5 us of epilogue, and in a launch of one to three rounds every CU is in the same phase at the same time: the
// prologue is an HBM burst with the matrix pipe idle, the epilogue a store burst (rounds 2-3; ablation
// -DMMF_G6_DBG=64).  Here
//   * workgroup w (one per CU, grid = min(tiles, CUs)) takes the tiles w, w + grid, ... of the launch's XCD-aware tile order
//     (mmf_xcd_tile: an XCD's 32 workgroups still share A / B panels in its L2);
//   * the LDS-DMA stream never stops: while tile j's last NS stages are multiplied, the refills of the ring fetch the first NS
//     stages of tile j + 1 (descriptors and per-lane source offsets switch at the hand-over of stage KT - NS), so tile j + 1's
//     first fragments are read behind the last MFMAs of tile j — no ring fill except the workgroup's very first;
//   * a tile's outputs are stored between its last stage and the next tile's first; the stores are not waited for (the stage
//     hand-overs keep their piece-only vmcnt counts: with stores in flight they wait for MORE than they need, never for less —
//     vmcnt counts loads and stores together) and drain under the next tile's k-loop.
// NT and NN, bf16 output, every K a multiple of 32 and >= (NS + 1) * 32; everything else stays on gemm6 (gemm.hip).
#include <algorithm>
#include "gemm6_parts.h"

namespace {

constexpr int BK = 32, NS = 4;
constexpr int TILE = 256 * BK * 2, STAGE = 2 * TILE, PPO = BK / 8, PPW = 2 * PPO;   // 16 KiB per operand tile, 8 pieces per wave and stage

typedef int i32x4_t __attribute__((ext_vector_type(4)));

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
                                           f32x16_t (&acc)[4][4], char* region, const int lane) {
  constexpr bool AUX = (CT & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) != 0;
  constexpr bool DROP = (CT & MMF_EPI_DROPOUT) != 0;
  // dropout on the (activated) outputs, the mask of gemm6's epilogue: element m * N + n of the caller's problem `orig[pi]`
  const unsigned drop_key = DROP ? mmf_rng_key(*args.rng_state, args.site, (unsigned)args.orig[pi]) : 0u;
  const float drop_scale = DROP ? 1.f / (1.f - (float)args.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  const float alpha = (CT & MMF_EPI_MASK_AUX) ? args.alpha : 1.f;            // 1 / (1 - p) of a dropout backward rides on the ReLU mask
  const int r = lane & 31, h = lane >> 5, q = lane >> 4, cq = lane & 15;
  // region offsets: accumulator layout (8 bytes at chunk c = 4 tn + g, half h, of row r), whole-row layout (16-byte chunk cq of row 4 it + q)
  const unsigned wr_base = (unsigned)(r * 256 + 8 * h + 16 * (r & 15));
  const unsigned rd_base = (unsigned)(q * 256 + ((cq ^ q) << 4));
  auto acc_at = [&](int c) { return region + (wr_base ^ (unsigned)(c << 4)); };
  auto row_at = [&](int it) { return region + it * 1024 + (rd_base ^ (unsigned)((it & 3) << 6)); };
  // global side: lane's column chunk is fixed (8 bf16 at nb + 8 cq), its row walks 32 tm + 4 it + q.  Range-checked buffer accesses
  // against [base, last valid element]: rows past M fall outside; columns past N get an out-of-range offset.
  const bool col_ok = nb + 8 * cq < P.N;
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, (int)((((long)P.M - 1) * P.ldc + P.N) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(AUX ? P.aux : P.C), 0,
                                                                       AUX ? (int)((((long)P.M - 1) * P.ldaux + P.N) * 2) : 0, 0x00020000);
  const unsigned c_off0 = col_ok ? (unsigned)(((long)(mb + q) * P.ldc + nb + 8 * cq) * 2) : 0x80000000u;
  const unsigned a_off0 = col_ok ? (unsigned)(((long)(mb + q) * P.ldaux + nb + 8 * cq) * 2) : 0x80000000u;
  const unsigned c_row4 = (unsigned)(4 * P.ldc * 2), a_row4 = (unsigned)(4 * P.ldaux * 2);

  u32x4_t auxr[2][8];
  auto load_aux = [&](int tm, int slot) {
#pragma unroll
    for (int it = 0; it < 8; ++it)
      auxr[slot][it] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ars, a_off0 + (unsigned)(8 * tm + it) * a_row4, 0, 0));
  };
  if constexpr (AUX) { load_aux(0, 0); load_aux(1, 1); }
  // (scheduling fences between the groups: without them the compiler hoists all 256 accumulator reads of the unrolled drain to its
  // head and spills what it cannot hold — accumulators included)
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    u32x2_t axv[4][4];
    if constexpr (AUX) {
#pragma unroll
      for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4_t*>(row_at(it)) = auxr[tm & 1][it];
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) axv[tn][g] = *reinterpret_cast<const u32x2_t*>(acc_at(4 * tn + g));
      __builtin_amdgcn_sched_barrier(0);
      if (tm + 2 < 4) load_aux(tm + 2, tm & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      // (the tile passes through an opaque statement HERE: its sixteen accumulator reads cannot be hoisted to the head of the drain,
      // where hipcc otherwise reads 160-250 accumulator registers into vector registers at once and spills the aux pieces in flight)
      asm volatile("" : "+a"(acc[tn][tm]));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t v = {acc[tn][tm][4 * g], acc[tn][tm][4 * g + 1], acc[tn][tm][4 * g + 2], acc[tn][tm][4 * g + 3]};
        if constexpr (CT & MMF_EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (DROP) {
          const unsigned idx = (unsigned)(mb + 32 * tm + r) * (unsigned)P.N + (unsigned)(nb + 32 * tn + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mmf_keep(drop_key, idx + e, args.drop_thresh) ? v[e] * drop_scale : 0.f;
        }
        if constexpr (AUX) {
          const u32x2_t a = axv[tn][g];
          const float a0 = bf16lo(a[0]), a1 = bf16hi(a[0]), a2 = bf16lo(a[1]), a3 = bf16hi(a[1]);
          if constexpr (CT & MMF_EPI_MASK_AUX) {
            v[0] = a0 > 0.f ? v[0] * alpha : 0.f; v[1] = a1 > 0.f ? v[1] * alpha : 0.f;
            v[2] = a2 > 0.f ? v[2] * alpha : 0.f; v[3] = a3 > 0.f ? v[3] * alpha : 0.f;
          } else {
            v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
          }
        }
        *reinterpret_cast<u32x2_t*>(acc_at(4 * tn + g)) = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    u32x4_t w[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) w[it] = *reinterpret_cast<const u32x4_t*>(row_at(it));
#pragma unroll
    for (int it = 0; it < 8; ++it) __builtin_amdgcn_raw_buffer_store_b128(w[it], crs, c_off0 + (unsigned)(8 * tm + it) * c_row4, 0, 0);
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
__device__ __forceinline__ void gemm7_body(const GemmArgs& args, const int total_tiles, char* smem) {
  constexpr bool A_KR = false;
  constexpr int WA = BK, WB = B_KR ? 256 : BK;

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
      v[i] = piece_voff<A_KR, BK>(wave + 4 * i, la, lane);
      v[PPO + i] = piece_voff<B_KR, BK>(wave + 4 * i, lb, lane);
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

  // ---- the workgroup's one ring fill: NS - 1 stages and the late half of the NS-th (the same two-instruction pieces: no LDS-DMA the
  // compiler knows of, so the LDS accesses of drain_tile are never preceded by a compiler-placed vmcnt(0)) ---------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const unsigned rb = lds_pieces + (unsigned)(s * STAGE);
    hot_piece(std::integral_constant<int, 0>{}, rb); hot_piece(std::integral_constant<int, 1>{}, rb);
    hot_piece(std::integral_constant<int, 2>{}, rb); hot_piece(std::integral_constant<int, 3>{}, rb);
    if (s + 1 < NS) {
      hot_piece(std::integral_constant<int, 4>{}, rb); hot_piece(std::integral_constant<int, 5>{}, rb);
      hot_piece(std::integral_constant<int, 6>{}, rb); hot_piece(std::integral_constant<int, 7>{}, rb);
      advance(dA, stepA); advance(dB, stepB);                // afterwards: stage NS - 1, the one being fetched
    }
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
  unsigned la0, la1, lb0, lb1;
  Frag4<A_KR>::lane_parts(WA, lane, la0, la1);
  Frag4<B_KR>::lane_parts(WB, lane, lb0, lb1);
  const unsigned smem_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const unsigned qa = (unsigned)((WA / 32) * 2048 * 4 * wm);
  const unsigned qb = (unsigned)(B_KR ? 512 * 4 * wn : (WB / 32) * 2048 * 4 * wn);
  la0 += smem_base + qa; la1 += smem_base + qa;
  lb0 += smem_base + TILE + qb; lb1 += smem_base + TILE + qb;

  f32x16_t acc[4][4];                 // [tn][tm]; set to the bias (or to zero) by bias_init in front of every tile

  // Bias.  In the accumulator layout a lane would hold 64 bias values per tile (columns nb + 32 tn + 8 g + 4 h + e) through the
  // drain, on top of the aux pieces in flight: with them the drain spills.  Instead the tile's accumulators START from the bias: lane
  // r keeps ONE value per 32-column block tn (bias[nb + 32 tn + r], loaded a tile ahead: at the head of the previous tile's drain) and
  // sixteen MFMAs spread them as outer products with a ones fragment,
  //     D[n][m] = sum_k A[n][k] B[k][m],  A[n][0..2] = the exact three-way bf16 split of bias[n],  B[0..2][m] = 1
  // (exact in f32: 8 + 8 + 8 mantissa bits; attention2.hip does the same with its row statistics).  16 of a K = 768 tile's 784 MFMAs.
  constexpr bool use_bias = (CT & MMF_EPI_BIAS) != 0;
  float bcur[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_bias = [&](float (&b)[4], const TileDst& d) {
    const mmf_gemm_problem& P = args.p[d.pi];
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.bias), 0, P.N * 4, 0x00020000);
    const unsigned off = (unsigned)((d.n0 + 128 * wn + (lane & 31)) * 4);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)                              // (the builtin returns the 32 bits as an integer)   past N: 0
      b[tn] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(brs, off + (unsigned)(32 * tn * 4), 0, 0));
  };
  auto split3 = [&](float x) {
    const bool lo = lane < 32;
    const unsigned hh = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hh);
    const unsigned mm = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mm);
    const unsigned ll = __float_as_uint(r2) & 0xffff0000u;
    const u32x4_t w = {lo ? ((hh >> 16) | mm) : 0u, lo ? (ll >> 16) : 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8_t, w);
  };
  auto bias_init = [&](const float (&b)[4]) {
    const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const u32x4_t ow = {lane < 32 ? 0x3f803f80u : 0u, lane < 32 ? 0x00003f80u : 0u, 0u, 0u};
    // one opaque copy of the ones fragment per row block: sixteen MFMAs with the same operands would be merged into four (or, without
    // a bias, into ONE) and their results copied into the other accumulator tiles register by register
    u32x4_t ones[4] = {ow, ow, ow, ow};
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) asm volatile("" : "+v"(ones[tm]));
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      u32x4_t bw = __builtin_bit_cast(u32x4_t, use_bias ? split3(b[tn]) : __builtin_bit_cast(bf16x8_t, u32x4_t{0u, 0u, 0u, 0u}));
      asm volatile("" : "+v"(bw));
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bw), __builtin_bit_cast(bf16x8_t, ones[tm]), zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if constexpr (use_bias) load_bias(bcur, cd);

  vm_wait<PPW * (NS - 2) + 4>();      // stage 0 landed (with the bias loads behind the pieces this waits for more: once per workgroup)
  __builtin_amdgcn_s_barrier();
  Frag4<A_KR> fa[2];
  Frag4<B_KR> fb[2];
  fa[0].template issue<WA, 0, 0>(la0, la1);
  fb[0].template issue<WB, 0, 0>(lb0, lb1);
  frag_wait(fa[0], fb[0]);

  auto mf = [&](const Frag4<A_KR>& a, const Frag4<B_KR>& b, int i) {
    const int tm = i >> 2, tn = i & 3;
    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.get(tn), a.get(tm), acc[tn][tm], 0, 0, 0);
    // an empty statement that "uses" the tile: the MFMA builtin has no side effect, and in this kernel instruction selection sank the
    // sixteen MFMAs of a substep below the reads and pieces that are written between them (each down to its next use, the same
    // tile's MFMA of the following substep); the volatile statement chains it into the order of the other asm statements
    asm volatile("" ::"a"(acc[tn][tm]));
    __builtin_amdgcn_sched_barrier(0);
  };
#define MMF_G7_READ(dstA, dstB, G, u, so)                                                              \
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

  // One stage (gemm6.hip's schedule, BK = 32: two k-substeps of sixteen MFMAs).  g: the workgroup's running stage count (ring slot
  // g % NS).  kt == ksw: this is stage KT - NS of its tile — from its hand-over on the refills fetch the NEXT tile (or nothing).
  // The hand-over counts PIECES only.  Stores, aux and bias loads of a drain may be younger than the pieces waited for: the wait
  // then covers more than it needs (safe whether or not the hardware retires loads and stores in one order), never less.
  auto stage = [&](const unsigned g, const int kt, const int ksw) {
    unsigned long long swmask;
    const Desc nA = mkdesc(ns.Ab, ns.recA), nB = mkdesc(ns.Bb, ns.recB);
    const unsigned so = (g % NS) * STAGE;
    const unsigned ring_cur = lds_pieces + so, ring_prev = lds_pieces + ((g + NS - 1) % NS) * STAGE;
    // substep 0: MFMA i (i < 8) is followed by one read of substep 1's fragments; the early pieces (second half of the stage whose
    // late half went out at the end of the previous stage) ride behind MFMAs 8, 10, 12, 14
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      mf(fa[0], fb[0], i);
      if (i < 8) MMF_G7_READ(fa[1], fb[1], 1, i, so);
      if (i == 8)  hot_piece(std::integral_constant<int, 4>{}, ring_prev);
      if (i == 10) hot_piece(std::integral_constant<int, 5>{}, ring_prev);
      if (i == 12) hot_piece(std::integral_constant<int, 6>{}, ring_prev);
      if (i == 14) hot_piece(std::integral_constant<int, 7>{}, ring_prev);
    }
    frag_wait(fa[1], fb[1]);
    // substep 1: four MFMAs, the stage hand-over, then MFMAs 4..11 each followed by one read of the next stage's first fragments and
    // MFMAs 12..15 by the late pieces
    const unsigned sn = ((g + 1) % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) mf(fa[1], fb[1], i);
    vm_wait<PPW * (NS - 2)>();
    __builtin_amdgcn_s_barrier();                              // the next stage landed for everyone; nobody reads this one any more
    __builtin_amdgcn_sched_barrier(0);
    // Behind the hand-over: advance the descriptors, or switch them (and the per-lane offsets) to the next tile's first stage — as
    // SELECTS, not as a branch (with a branch here hipcc duplicated the rest of the stage into both arms and joined the two copies'
    // 256 accumulator registers with v_accvgpr_mov chains behind the k-loop), in four portions of at most nine instructions behind
    // MFMAs 4..7 (a wave issues eight instructions per 32-cycle MFMA).  The selects are inline asm: ONE compare feeds all of them
    // (hipcc re-derived the condition per portion: 45 instructions), and left to itself it computes all of it between the vmcnt
    // wait and the barrier, in front of an idle matrix pipe.
#pragma unroll
    for (int i = 4; i < 16; ++i) {
      mf(fa[1], fb[1], i);
      if (i < 12) MMF_G7_READ(fa[0], fb[0], 0, i - 4, sn);
      if (i == 4) {                                            // both descriptors one stage on (eight scalar instructions)
        advance(dA, stepA);
        advance(dB, stepB);
        asm volatile("" ::"s"(dA.lo), "s"(dA.hi), "s"(dA.rec), "s"(dB.lo), "s"(dB.hi), "s"(dB.rec));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 5) {                                            // one compare, seven scalar selects, the lane mask for the vector selects
        asm volatile("s_cmp_eq_u32 %8, %16\n\t"
                     "s_cselect_b32 %0, %9, %0\n\ts_cselect_b32 %1, %10, %1\n\ts_cselect_b32 %2, %11, %2\n\t"
                     "s_cselect_b32 %3, %12, %3\n\ts_cselect_b32 %4, %13, %4\n\ts_cselect_b32 %5, %14, %5\n\t"
                     "s_cselect_b32 %6, %15, %6\n\ts_cselect_b64 %7, -1, 0"
                     : "+s"(dA.lo), "+s"(dA.hi), "+s"(dA.rec), "+s"(dB.lo), "+s"(dB.hi), "+s"(dB.rec), "+s"(stepB), "=s"(swmask)
                     : "s"(kt), "s"(nA.lo), "s"(nA.hi), "s"(nA.rec), "s"(nB.lo), "s"(nB.hi), "s"(nB.rec), "s"(ns.stepB), "s"(ksw)
                     : "scc");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 6) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %4, %8\n\tv_cndmask_b32_e64 %1, %1, %5, %8\n\t"
                     "v_cndmask_b32_e64 %2, %2, %6, %8\n\tv_cndmask_b32_e64 %3, %3, %7, %8"
                     : "+v"(voff[0]), "+v"(voff[1]), "+v"(voff[2]), "+v"(voff[3])
                     : "v"(voffn[0]), "v"(voffn[1]), "v"(voffn[2]), "v"(voffn[3]), "s"(swmask));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 7) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %4, %8\n\tv_cndmask_b32_e64 %1, %1, %5, %8\n\t"
                     "v_cndmask_b32_e64 %2, %2, %6, %8\n\tv_cndmask_b32_e64 %3, %3, %7, %8"
                     : "+v"(voff[4]), "+v"(voff[5]), "+v"(voff[6]), "+v"(voff[7])
                     : "v"(voffn[4]), "v"(voffn[5]), "v"(voffn[6]), "v"(voffn[7]), "s"(swmask));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 12) hot_piece(std::integral_constant<int, 0>{}, ring_cur);
      if (i == 13) hot_piece(std::integral_constant<int, 1>{}, ring_cur);
      if (i == 14) hot_piece(std::integral_constant<int, 2>{}, ring_cur);
      if (i == 15) hot_piece(std::integral_constant<int, 3>{}, ring_cur);
    }
    frag_wait(fa[0], fb[0]);
  };

  char* const region = smem + NS * STAGE + wave * 8192;
  unsigned g = 0;
  for (;;) {
    bias_init(bcur);
    const int ksw = KT - NS;                                   // >= 1 (host)
    for (int kt = 0; kt < KT; ++kt, ++g) stage(g, kt, ksw);
    // ---- the tile's outputs ----------------------------------------------------------------------------------------------------------
    const mmf_gemm_problem& P = args.p[cd.pi];
    const int mb = cd.m0 + 128 * wm, nb = cd.n0 + 128 * wn;
    if constexpr (use_bias) { if (has_next) load_bias(bcur, nd); }   // the NEXT tile's bias: the drain covers the round trip
#ifndef MMF_G7_NODRAIN
    drain_tile<CT & ~MMF_EPI_BIAS>(args, cd.pi, P, mb, nb, acc, region, lane);
#else
    if (lane == 0 && mb == -1) static_cast<volatile unsigned short*>(P.C)[0] = (unsigned short)acc[0][0][0];   // (ablation: the tile is not stored)
#endif
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
#undef MMF_G7_READ
}

template <bool B_KR, int CT>
__global__ __launch_bounds__(NTHREADS, 1)
void gemm7_persistent_kernel(const GemmArgs args, const int total_tiles) {
  // the ring and, behind it, one 8-KiB output staging region per wave: all 160 KiB of the CU
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE + 4 * 8192];
  gemm7_body<B_KR, CT>(args, total_tiles, smem);
}

// the (layout, flag set) pairs of the fusion step's NT / NN launches: in-projections (NT, bias), FFN1 (NT, bias + ReLU), out-projection
// and FFN2 (NT, bias + residual), plain NT, dgrads (NN, none), dH (NN, ReLU mask), dX (NN, residual gradient)
template <bool B_KR, int CT>
void launch7(const GemmArgs& a, int total, int grid, hipStream_t s) {
  hipLaunchKernelGGL((gemm7_persistent_kernel<B_KR, CT>), dim3(grid), dim3(NTHREADS), 0, s, a, total);
}
bool launch7_select(int layout, int eflags, const GemmArgs* a, int total, int grid, hipStream_t s) {   // a == nullptr: only ask
  if (layout == MMF_GEMM_NT) {
    if (eflags == 0)                                      { if (a) launch7<false, 0>(*a, total, grid, s); return true; }
    if (eflags == MMF_EPI_BIAS)                           { if (a) launch7<false, MMF_EPI_BIAS>(*a, total, grid, s); return true; }
    if (eflags == (MMF_EPI_BIAS | MMF_EPI_RELU))          { if (a) launch7<false, MMF_EPI_BIAS | MMF_EPI_RELU>(*a, total, grid, s); return true; }
    if (eflags == (MMF_EPI_BIAS | MMF_EPI_ADD_AUX))       { if (a) launch7<false, MMF_EPI_BIAS | MMF_EPI_ADD_AUX>(*a, total, grid, s); return true; }
    if (eflags == (MMF_EPI_BIAS | MMF_EPI_RELU | MMF_EPI_DROPOUT)) {       // FFN hidden layer in training mode
      if (a) launch7<false, MMF_EPI_BIAS | MMF_EPI_RELU | MMF_EPI_DROPOUT>(*a, total, grid, s);
      return true;
    }
  } else if (layout == MMF_GEMM_NN) {
    if (eflags == 0)                                      { if (a) launch7<true, 0>(*a, total, grid, s); return true; }
    if (eflags == MMF_EPI_MASK_AUX)                       { if (a) launch7<true, MMF_EPI_MASK_AUX>(*a, total, grid, s); return true; }
    if (eflags == MMF_EPI_ADD_AUX)                        { if (a) launch7<true, MMF_EPI_ADD_AUX>(*a, total, grid, s); return true; }
  }
  return false;
}
}  // namespace

static int g_persistent_wgs = [] { const char* e = getenv("MMF_GEMM7_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
extern "C" int mmf_gemm_set_persistent_workgroups(int n) {
  if (n < 0 || n > 65536) MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_set_persistent_workgroups: %d not in 0..65536", n);
  g_persistent_wgs = n;
  return MMF_OK;
}

// whether the persistent kernel can take this launch (gemm.hip asks before selecting it)
bool mmf_gemm7_supports(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue, int out_f32, const mmf_gemm_extra* extra) {
  if (out_f32 || !launch7_select(layout, epilogue, nullptr, 0, 0, nullptr)) return false;
  if (extra && extra->alpha != 1.f && !(layout == MMF_GEMM_NN && epilogue == MMF_EPI_MASK_AUX)) return false;   // alpha rides on the mask only
  if ((epilogue & MMF_EPI_DROPOUT) && !(extra && extra->rng_state)) return false;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_gemm_problem& p = problems[i];
    if (p.K % BK || p.K < (NS + 1) * BK || (p.N & 7) || (p.ldc & 7) || (p.aux && (p.ldaux & 7))) return false;
  }
  return true;
}

int mmf_gemm7_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s) {
  if (!mmf_gemm7_supports(problems, num_problems, layout, epilogue, out_f32, extra))
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
  int order[MMF_GEMM_MAX_PROBLEMS];                            // longest reduction first (see the walk in gemm7_body)
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
  const int want = g_persistent_wgs > 0 ? g_persistent_wgs : cus;
  const int grid = total < want ? total : want;
  launch7_select(layout, epilogue, &a, total, grid, s);
  MMF_CHECK_LAUNCH("mmf_gemm_grouped(v7)");
  return MMF_OK;
}
