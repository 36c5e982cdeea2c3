// Shared pieces of the attention kernels (attention2.hip): the work table, LDS-DMA staging of tile pairs in the dual-use
// LDS image of attn_helpers.h, asm transposed reads with counted waits, and the host-side table fill.
#pragma once
#include "mmf_internal.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "attn_helpers.h"

namespace {

#ifndef DQ_WAVES_PER_SIMD
#define DQ_WAVES_PER_SIMD 3            // dQ kernel at <= 168 registers: three 4-wave workgroups per CU (3 x 52 KiB of LDS)
#endif
#ifndef MMF_DKV_TRDEPTH
#define MMF_DKV_TRDEPTH 3              // transposed fragments in flight ahead of the dV^T / dK^T MFMA chain (DkvStepD)
#endif
#ifndef MMF_DQ_TRDEPTH
#define MMF_DQ_TRDEPTH 1               // ... ahead of the dQ^T chain (PvStepD): the dQ kernel is at its 168-register budget
#endif
constexpr float DEFER = 6.0f;          // log2 domain: P <= 64 before a rescale is forced
// MMF_ATTN_SETPRIO=1 (build-time A/B): raise the wave's issue priority around its MFMA clusters (guide T5), so that of the
// two waves sharing a SIMD the one entering a matrix phase is not held behind the other's softmax VALU stream
#ifndef MMF_ATTN_SETPRIO
#define MMF_ATTN_SETPRIO 0
#endif
__device__ __forceinline__ void mfma_prio(int on) {
#if MMF_ATTN_SETPRIO
  if (on) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#else
  (void)on;
#endif
}

struct AttnArgs2 {
  int nprob;
  float scale;
  unsigned drop_thresh, site;
  float inv_keep;
  const unsigned long long* rng_state;
  int split;      // sweep split of the <= 32-row problems in the backward kernels (MMF_ATTN_SPLIT=0: off)
  int blk_start[MMF_ATTN_MAX_PROBLEMS + 1];   // multiples of 8 (XCD alignment)
  int nwg[MMF_ATTN_MAX_PROBLEMS];             // real workgroups of the problem = B*H*nchunk
  short nchunk[MMF_ATTN_MAX_PROBLEMS];        // query chunks per (b, h)
  short rpc[MMF_ATTN_MAX_PROBLEMS];           // rows per chunk (multiple of 32, <= 256)
  short orig[MMF_ATTN_MAX_PROBLEMS];          // caller's problem index (dropout stream id)
  mmf_attn_problem p[MMF_ATTN_MAX_PROBLEMS];
};

typedef __attribute__((address_space(3))) void lds_void_t;

template <int OFF>
__device__ __forceinline__ s16x4_t tr_read_imm(unsigned addr) {
  s16x4_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// LDS byte addresses of a tile for this lane's transposed reads: tile base + tr_lane_lo / tr_lane_hi (attn_helpers.h)
struct TrBase { unsigned lo, hi; };
__device__ __forceinline__ TrBase tr_base(unsigned tile_lds, unsigned lane_lo, unsigned lane_hi) {
  return TrBase{tile_lds + lane_lo, tile_lds + lane_hi};
}

// One transposed fragment (32 columns [32 D, 32 D + 32) of the 16 tile rows [16 G, 16 G + 16)): two ds_read_b64_tr_b16.
// Issued from inline asm so that hipcc's LDS-DMA alias bookkeeping does not drain the next tile's DMA (s_waitcnt
// vmcnt(0)) before them; tr_wait<N>() ties the destination registers to the counted lgkmcnt wait, so no use can be
// scheduled above it.
template <int DH, int G, int D>
__device__ __forceinline__ void tr_issue(TrBase va, s16x4_t& lo, s16x4_t& hi) {
  lo = tr_read_imm<img_tr_imm<DH, G, D>()>(va.lo);
  hi = tr_read_imm<img_tr_imm<DH, G, D>() + (DH / 32) * 512>(va.hi);
}
template <int PENDING>
__device__ __forceinline__ void tr_wait(s16x4_t& lo, s16x4_t& hi) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(PENDING));
}
__device__ __forceinline__ bf16x8_t join(const s16x4_t& lo, const s16x4_t& hi) {
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// Depth-D read-ahead over a statically known sequence of NF transposed fragments: fragments n+1 .. n+D are in flight
// while MFMA n runs; the counted wait names exactly the 2 * min(D, NF-1-n) younger reads (LDS returns in order; any OTHER
// younger LDS / SMEM operation only makes the wait stricter, never weaker).  Round 3 measured depths 1 / 3 / 4 / 6 of the
// backward chains level (cross x6 160.3 / 163.6 / 162.1 us; 6 spills registers in the dK/dV kernel and loses): with two or
// three waves per SIMD the partner waves cover one wave's LDS latency, what the kernels lacked was LDS bandwidth.
template <int NF, int D, int N>
constexpr int tr_pending() { return 2 * ((N + D < NF - 1 ? N + D : NF - 1) - N); }

// O^T += V^T . P^T (forward) and dQ^T += K^T . dS^T (dQ kernel) for the 32 keys of block KT of the tile at `va`:
// fragment n = (k-substep n / DT, d-tile n % DT)
template <int DH, int KT, int D, int N>
struct PvStepD {
  static constexpr int DT = DH / 32, NF = 2 * DT;
  template <int M>
  static __device__ __forceinline__ void issue(TrBase va, s16x4_t (&lo)[NF], s16x4_t (&hi)[NF]) {
    tr_issue<DH, 2 * KT + M / DT, M % DT>(va, lo[M], hi[M]);
  }
  static __device__ __forceinline__ void prime(TrBase va, s16x4_t (&lo)[NF], s16x4_t (&hi)[NF]) {
    if constexpr (N < D && N < NF) { issue<N>(va, lo, hi); PvStepD<DH, KT, D, N + 1>::prime(va, lo, hi); }
  }
  static __device__ __forceinline__ void run(TrBase va, s16x4_t (&lo)[NF], s16x4_t (&hi)[NF], const f32x16_t& s,
                                             bf16x8_t& pf, f32x16_t (&o)[DT]) {
    if constexpr (N + D < NF) issue<N + D>(va, lo, hi);
    tr_wait<tr_pending<NF, D, N>()>(lo[N], hi[N]);
    if constexpr (N == 0) mfma_prio(1);
    const bf16x8_t vf = join(lo[N], hi[N]);
    if constexpr (N % DT == 0) pf = acc_frag(s, N / DT);
    o[N % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[N % DT], 0, 0, 0);
    if constexpr (N + 1 < NF) PvStepD<DH, KT, D, N + 1>::run(va, lo, hi, s, pf, o);
    else mfma_prio(0);
  }
};

// dV^T += dO^T . P and dK^T += Q^T . dS for the 32 query rows of block QS: fragment n = (k-substep n / (2 DT), d-tile
// (n / 2) % DT, operand n & 1: 0 = dO tile -> dV^T, 1 = Q tile -> dK^T)
template <int DH, int QS, int D, int N>
struct DkvStepD {
  static constexpr int DT = DH / 32, NF = 4 * DT;
  template <int M>
  static __device__ __forceinline__ void issue(TrBase vaQ, TrBase vadO, s16x4_t (&lo)[NF], s16x4_t (&hi)[NF]) {
    tr_issue<DH, 2 * QS + M / (2 * DT), (M / 2) % DT>((M & 1) ? vaQ : vadO, lo[M], hi[M]);
  }
  static __device__ __forceinline__ void prime(TrBase vaQ, TrBase vadO, s16x4_t (&lo)[NF], s16x4_t (&hi)[NF]) {
    if constexpr (N < D && N < NF) { issue<N>(vaQ, vadO, lo, hi); DkvStepD<DH, QS, D, N + 1>::prime(vaQ, vadO, lo, hi); }
  }
  static __device__ __forceinline__ void run(TrBase vaQ, TrBase vadO, s16x4_t (&lo)[NF], s16x4_t (&hi)[NF],
                                             const f32x16_t& pm, const f32x16_t& dsm, bf16x8_t& pf, bf16x8_t& dsf,
                                             f32x16_t (&dv)[DT], f32x16_t (&dk)[DT]) {
    if constexpr (N + D < NF) issue<N + D>(vaQ, vadO, lo, hi);
    tr_wait<tr_pending<NF, D, N>()>(lo[N], hi[N]);
    if constexpr (N == 0) mfma_prio(1);
    const bf16x8_t f = join(lo[N], hi[N]);
    if constexpr (N % (2 * DT) == 0) { pf = acc_frag(pm, N / (2 * DT)); dsf = acc_frag(dsm, N / (2 * DT)); }
    constexpr int dt = (N / 2) % DT;
    if constexpr ((N & 1) == 0) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, pf, dv[dt], 0, 0, 0);
    else                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, dsf, dk[dt], 0, 0, 0);
    if constexpr (N + 1 < NF) DkvStepD<DH, QS, D, N + 1>::run(vaQ, vadO, lo, hi, pm, dsm, pf, dsf, dv, dk);
    else mfma_prio(0);
  }
};

// exact three-way bf16 split of an f32 value / a ones fragment: operands of the outer-product MFMA that spreads per-row statistics
// into the accumulator layout (attention2.hip, "Row statistics of the dK/dV kernel through the matrix pipe")
__device__ __forceinline__ bf16x8_t split3_frag(float x, bool low_half) {
  const unsigned h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  const unsigned m = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);
  const unsigned l = __float_as_uint(r2) & 0xffff0000u;
  const u32x4_t w = {low_half ? ((h >> 16) | m) : 0u, low_half ? (l >> 16) : 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, w);
}
__device__ __forceinline__ bf16x8_t ones3_frag(bool low_half) {
  const u32x4_t w = {low_half ? 0x3f803f80u : 0u, low_half ? 0x00003f80u : 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, w);
}

// One wave's share of the LDS-DMA of a stage = the image (attn_helpers.h) of two 64-row tiles X (at st) and Y (at st +
// tile bytes) of two (T, ld) matrices.  A tile is DH/8 one-KiB pieces; the stage's pieces are dealt to the 4 waves, piece
// p = wave + 4 i, so whether piece i of a wave belongs to X or Y is a compile-time fact (DH/8 is a multiple of 4).  LDS-DMA
// writes a piece lane-linearly (lane L -> bytes [16 L, 16 L + 16) of the piece), so the image's swizzle is applied to the
// per-lane SOURCE offset, which does not depend on the tile index: it is computed once (init) and a tile is addressed
// through a descriptor of its own whose base is the tile's first row and whose range ends after the matrix' row T - 1
// (rows past T read as zeros).  Round 2 recomputed the offsets — a division by DH/8 + 1 per piece — for every tile:
// 850-900 of a wave's ~4,000 cycles per tile in the stamped backward kernels (profiles/r03_attn_bwd_stamps.txt).
template <int DH>
struct TileDma {
  static constexpr int PIECES = DH / 8, NI = 2 * PIECES / 4, NC = DH / 32;
  static_assert(PIECES % 4 == 0, "a tile's pieces must deal evenly to four waves");
  unsigned voff[NI];
  __device__ __forceinline__ void init(int ldx, int ldy, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = (wave + 4 * i) % PIECES;                 // piece inside its tile
      const int st = 2 * p + (lane >> 5), rg = st / NC, cc = st % NC;      // subtile: row group, column block
      const int w = lane & 31, row = 8 * rg + (w >> 2), ch = 4 * cc + ((w & 3) ^ ((row >> 2) & 3));
      voff[i] = (unsigned)(row * (4 * i >= PIECES ? ldy : ldx) * 2 + ch * 16);
    }
  }
  // Xj / Yj: the matrices' (b, h) origins advanced to row 64 j; rows_left = T - 64 j
  __device__ __forceinline__ void issue(const unsigned short* Xj, const unsigned short* Yj, int ldx, int ldy, int rows_left,
                                        char* st, int wave) const {
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Xj), 0, rows_left * ldx * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Yj), 0, rows_left * ldy * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      char* dst = st + (wave + 4 * i) * 1024;
      if (4 * i < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)dst, 16, voff[i], 0, 0, 0);
      else                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void_t*)dst, 16, voff[i], 0, 0, 0);
    }
  }
};

}  // namespace

namespace {
// Work table: problems heaviest first; rows_per_wg rows of the partitioned axis (queries, or keys for the dK/dV
// kernel) per workgroup, balanced over the chunks; workgroup ranges padded to multiples of 8 for the XCD map.
int fill_args2(AttnArgs2& a, const mmf_attn_problem* problems, int n, float scale, float drop_p, const uint64_t* rng_state,
               uint32_t site, int rows_per_wg, bool by_keys, bool balance) {
  a.nprob = n; a.scale = scale;
  a.drop_thresh = (drop_p > 0.f && rng_state) ? mmf_drop_thresh(drop_p) : 0u;
  a.inv_keep = a.drop_thresh ? 1.f / (1.f - (float)a.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  a.site = site;
  a.rng_state = reinterpret_cast<const unsigned long long*>(rng_state);
  static const int split = [] { const char* e = getenv("MMF_ATTN_SPLIT"); return e ? atoi(e) : 1; }();
  a.split = split;
  int order[MMF_ATTN_MAX_PROBLEMS];
  for (int i = 0; i < n; ++i) order[i] = i;
  // Launch order = longest per-workgroup chain first: a workgroup's duration is set by the length of its sweep
  // (keys for the forward / dQ kernels, queries for dK/dV), not by its row count, and the narrow problems (30
  // rows: one active wave walking the whole sweep) are pure latency chains — started first they run beside the
  // wide problems instead of trailing them on an empty chip.
  auto key = [&](const mmf_attn_problem& q) {
    const int part = by_keys ? q.Tk : q.Tq, sweep = by_keys ? q.Tq : q.Tk;
    return (long long)((sweep + 63) / 64) * 4096 - std::min(part, rows_per_wg);
  };
  std::stable_sort(order, order + n, [&](int x, int y) { return key(problems[x]) > key(problems[y]); });
  int total = 0;
  for (int k = 0; k < n; ++k) {
    const mmf_attn_problem& q = problems[order[k]];
    const int part = by_keys ? q.Tk : q.Tq;
    const int nchunk = (part + rows_per_wg - 1) / rows_per_wg;
    const int rpc = balance ? (((part + nchunk - 1) / nchunk) + 31) / 32 * 32 : rows_per_wg;
    a.blk_start[k] = total;
    a.nwg[k] = q.B * q.H * nchunk;
    a.nchunk[k] = (short)nchunk; a.rpc[k] = (short)rpc; a.orig[k] = (short)order[k];
    a.p[k] = q;
    total += (a.nwg[k] + 7) / 8 * 8;
  }
  a.blk_start[n] = total;
  return total;
}
int check_ranges(const char* who, const mmf_attn_problem* p, int n) {
  for (int i = 0; i < n; ++i) {
    const long long lim = 0x7fffffffLL;
    if ((long long)p[i].Tk * p[i].ldk * 2 >= lim || (long long)p[i].Tk * p[i].ldv * 2 >= lim ||
        (long long)p[i].Tq * p[i].ldq * 2 >= lim || (long long)p[i].Tq * p[i].ldo * 2 >= lim)
      MMF_FAIL(MMF_E_SHAPE, "%s: T*ld exceeds the 2 GiB buffer-descriptor range", who);
  }
  return MMF_OK;
}
}  // namespace

