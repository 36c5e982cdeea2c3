// Shared pieces of the second/third-generation attention kernels (attention2.hip, attention3.hip): the work table,
// LDS-DMA staging of [64][DH+8] tile pairs, asm transposed reads with counted waits, and the host-side table fill.
#pragma once
#include "mmf_internal.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "attn_helpers.h"

namespace {

constexpr unsigned OOB = 0x80000000u;
#ifndef DQ_WAVES_PER_SIMD
#define DQ_WAVES_PER_SIMD 3            // dQ kernel at <= 168 registers: three 4-wave workgroups per CU (3 x 52 KiB of LDS)
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
  int debug;      // timing ablations (MMF_ATTN2_DEBUG, results wrong by design): 1 no K/V DMA after tile 0, 2 no compute
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

// One V^T fragment (32 head-dim columns D of the 16 keys [16 G, 16 G + 16) of the tile whose LDS byte address
// plus this lane's tr offset is `va`): two ds_read_b64_tr_b16.  Issued from inline asm so that hipcc's LDS-DMA
// alias bookkeeping does not drain the next tile's DMA (s_waitcnt vmcnt(0)) before them; tr_wait<N>() ties the
// destination registers to the counted lgkmcnt wait, so no use can be scheduled above it.
template <int DH, int G, int D>
__device__ __forceinline__ void tr_issue(unsigned va, s16x4_t& lo, s16x4_t& hi) {
  constexpr int SB = (DH + 8) * 2, OFF = 16 * G * SB + 64 * D;
  lo = tr_read_imm<OFF>(va);
  hi = tr_read_imm<OFF + 8 * SB>(va);
}
template <int PENDING>
__device__ __forceinline__ void tr_wait(s16x4_t& lo, s16x4_t& hi) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(PENDING));
}
__device__ __forceinline__ bf16x8_t join(const s16x4_t& lo, const s16x4_t& hi) {
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// O^T += V^T . P^T for the 32 keys of block KT: 2 DT fragment steps, fragment N + 1 in flight under step N's MFMAs
template <int DH, int NQ, int KT, int N>
struct PvStep {
  static constexpr int DT = DH / 32, NF = 2 * DT;
  static __device__ __forceinline__ void run(unsigned va, s16x4_t lo, s16x4_t hi, const f32x16_t (&s)[NQ],
                                             bf16x8_t (&pf)[NQ], f32x16_t (&o)[NQ][DT]) {
    s16x4_t nlo, nhi;
    if constexpr (N + 1 < NF) tr_issue<DH, 2 * KT + (N + 1) / DT, (N + 1) % DT>(va, nlo, nhi);
    tr_wait<(N + 1 < NF) ? 2 : 0>(lo, hi);
    if constexpr (N == 0) mfma_prio(1);
    const bf16x8_t vf = join(lo, hi);
    if constexpr (N % DT == 0) {
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) pf[qb] = acc_frag(s[qb], N / DT);
    }
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
      o[qb][N % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb], o[qb][N % DT], 0, 0, 0);
    if constexpr (N + 1 < NF) PvStep<DH, NQ, KT, N + 1>::run(va, nlo, nhi, s, pf, o);
    else mfma_prio(0);
  }
};

// The same with the transposed reads TWO fragments ahead (fragments N and N + 1 arrive in flight, N + 2 is issued here):
// an MFMA (32 cycles) is shorter than an LDS round trip, so one fragment of read-ahead leaves most of every read exposed
// whenever the wave has its SIMD to itself (the 30-row problems, partly filled launches).
template <int DH, int NQ, int KT, int N>
struct PvStep2 {
  static constexpr int DT = DH / 32, NF = 2 * DT;
  static __device__ __forceinline__ void run(unsigned va, s16x4_t lo, s16x4_t hi, s16x4_t lo1, s16x4_t hi1,
                                             const f32x16_t (&s)[NQ], bf16x8_t (&pf)[NQ], f32x16_t (&o)[NQ][DT]) {
    s16x4_t lo2, hi2;
    if constexpr (N + 2 < NF) tr_issue<DH, 2 * KT + (N + 2) / DT, (N + 2) % DT>(va, lo2, hi2);
    tr_wait<(N + 2 < NF) ? 4 : (N + 1 < NF) ? 2 : 0>(lo, hi);
    if constexpr (N == 0) mfma_prio(1);
    const bf16x8_t vf = join(lo, hi);
    if constexpr (N % DT == 0) {
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) pf[qb] = acc_frag(s[qb], N / DT);
    }
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
      o[qb][N % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb], o[qb][N % DT], 0, 0, 0);
    if constexpr (N + 1 < NF) PvStep2<DH, NQ, KT, N + 1>::run(va, lo1, hi1, lo2, hi2, s, pf, o);
    else mfma_prio(0);
  }
};

// LDS-DMA of one stage = two [64][DH+8] tiles X (at st) and Y (at st + TILE_B) of rows [64 j, 64 j + 64) of two
// (T, ld) matrices given as buffer descriptors whose range ends after row T-1 (rows past T read as zeros).
// Piece p = wave + 4 i covers image chunks 64 pc .. 64 pc + 63 of X (p < PIECES) or Y; chunk c is row c / CPR,
// 16-B column c % CPR (the last column is the pad: explicit out-of-range offset -> zeros).
template <int DH>
__device__ __forceinline__ void dma_pair(__amdgpu_buffer_rsrc_t rsX, __amdgpu_buffer_rsrc_t rsY, int ldx, int ldy,
                                         char* st, int j, int wave, int lane) {
  constexpr int SB = (DH + 8) * 2, TILE_B = 64 * SB, CPR = DH / 8 + 1, PIECES = TILE_B / 1024, NI = (2 * PIECES + 3) / 4;
  static_assert(TILE_B % 1024 == 0, "a tile must be a whole number of 1-KiB LDS-DMA pieces");
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = wave + 4 * i;
    if (p < 2 * PIECES) {
      const int isy = p >= PIECES, c = (p - isy * PIECES) * 64 + lane;
      const int row = c / CPR, ch = c % CPR;
      const int ld = isy ? ldy : ldx;
      const unsigned off = ch == CPR - 1 ? OOB : (unsigned)((64 * j + row) * ld * 2 + ch * 16);
      if (!isy) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(st + p * 1024), 16, off, 0, 0, 0);
      else      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void_t*)(st + p * 1024), 16, off, 0, 0, 0);
    }
  }
}

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
  const char* dbg = getenv("MMF_ATTN2_DEBUG");
  a.debug = dbg ? atoi(dbg) : 0;
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

