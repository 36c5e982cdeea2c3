// Shared device helpers of the attention kernels (attention.hip: first generation, register-staged;
// attention2.hip: LDS-DMA ring, two query blocks per wave).  MFMA v_mfma_f32_32x32x16_bf16 fragment
// layouts and the padded [rows][DH+8] LDS image are described at each helper.
#pragma once
#include "mmf_internal.h"

namespace {

constexpr int NT = 256;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Combine a value with its partner in the other half-wave (lane i <-> lane i + 32) without the LDS round trip of
// ds_bpermute: v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second, so with
// both operands = x the two results hold {x.lo, x.lo} and {x.hi, x.hi} (guide T12 / T21).
__device__ __forceinline__ float half_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- LDS tile helpers: image [rows][DH + 8] bf16 ------------------------------------------------
// Stage a [ROWS][DH] tile of a (T, ld) matrix whose (b, h) origin is `base`; rows >= T read as zero.
template <int DH, int ROWS>
struct TileStage {
  static constexpr int CPR = DH / 8;                      // 16-B chunks per row
  static constexpr int PER_THREAD = ROWS * CPR / NT;
  static_assert(ROWS * CPR % NT == 0, "tile must split evenly over the workgroup");
  u32x4_t r[PER_THREAD];
  __device__ __forceinline__ void load(const unsigned short* __restrict__ base, int ld, int row0, int T, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int c = tid + NT * i;
      const int row = c / CPR, ch = c % CPR;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (row0 + row < T) v = *reinterpret_cast<const u32x4_t*>(base + (size_t)(row0 + row) * ld + ch * 8);
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int c = tid + NT * i;
      const int row = c / CPR, ch = c % CPR;
      *reinterpret_cast<u32x4_t*>(tile + row * ((DH + 8) * 2) + ch * 16) = r[i];
    }
  }
};

// Row fragment (MFMA 32x32x16 A or B operand whose 32-index is the tile ROW): lane l gets
// tile[row0 + (l & 31)][16 ks + 8 (l >> 5) + 0..7].
template <int DH>
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(tile + (row0 + (lane & 31)) * ((DH + 8) * 2) +
                                            (2 * ks + (lane >> 5)) * 16);
}

// Transposed fragment (A operand X^T[i = column][k = row]) for the 16 tile rows [r0, r0+16) and the
// 32 columns [c0, c0+32): element j of lane (i = l & 31, h = l >> 5) is
// tile[r0 + 8 (j >> 2) + 4 h + (j & 3)][c0 + i] — the k order in which an accumulator tile presents
// its rows when it is used as the other operand.
template <int DH>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int r0, int c0, int lane) {
  constexpr int SB = (DH + 8) * 2;
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const char* a = tile + (r0 + 4 * h + q) * SB + (c0 + 16 * g + 4 * p) * 2;
  const s16x4_t lo = lds_read_tr16(a);
  const s16x4_t hi = lds_read_tr16(a + 8 * SB);
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// registers 8s..8s+7 of a 32x32 accumulator as a bf16 operand fragment
__device__ __forceinline__ bf16x8_t acc_frag(const f32x16_t& x, int s) {
  const u32x4_t w = {pack_bf16x2(x[8 * s + 0], x[8 * s + 1]), pack_bf16x2(x[8 * s + 2], x[8 * s + 3]),
                     pack_bf16x2(x[8 * s + 4], x[8 * s + 5]), pack_bf16x2(x[8 * s + 6], x[8 * s + 7])};
  return __builtin_bit_cast(bf16x8_t, w);
}

// Lane-resident row fragments of 32 rows taken directly from HBM (Q in forward/dQ, K/V in dK/dV):
// lane l holds row (row0 + (l & 31)), columns 16 ks + 8 (l >> 5) + 0..7; rows >= T are zero.
template <int DH>
__device__ __forceinline__ void load_row_frags(bf16x8_t (&f)[DH / 16], const unsigned short* __restrict__ base,
                                               int ld, int row0, int T, int lane) {
  const int row = row0 + (lane & 31);
#pragma unroll
  for (int ks = 0; ks < DH / 16; ++ks) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < T) v = *reinterpret_cast<const u32x4_t*>(base + (size_t)row * ld + 16 * ks + 8 * (lane >> 5));
    f[ks] = __builtin_bit_cast(bf16x8_t, v);
  }
}

// store a [d][q or key] accumulator set as rows of a (T, ld) bf16 matrix: lane owns row (row0 + (l&31)),
// register 4g+i of tile dt is column 32 dt + 8 g + 4 (l>>5) + i.
template <int DH>
__device__ __forceinline__ void store_rows(const f32x16_t (&o)[DH / 32], float mul, unsigned short* __restrict__ base,
                                           int ld, int row0, int T, int lane) {
  const int row = row0 + (lane & 31);
  if (row >= T) return;
  unsigned short* p = base + (size_t)row * ld + 4 * (lane >> 5);
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2_t w = {pack_bf16x2(o[dt][4 * g + 0] * mul, o[dt][4 * g + 1] * mul),
                         pack_bf16x2(o[dt][4 * g + 2] * mul, o[dt][4 * g + 3] * mul)};
      *reinterpret_cast<u32x2_t*>(p + 32 * dt + 8 * g) = w;
    }
}

// ---- whole-row HBM access through a wave-private LDS slice ------------------------------------------
// Fragment-shaped loads / stores (lane = row) touch 32 rows x 32 B (loads) or 32 rows x 8 B (stores) per
// instruction: 32+ cache lines each, and the 8-byte pieces make partial-line writes.  For small
// attention problems that prologue/epilogue is most of the kernel.  Instead a wave moves its 32 rows as
// 16-byte chunks in row-major order (12 lanes per 192-B row) and converts to/from the MFMA fragment
// layout in a [32][DH+8] LDS slice it owns (LDS ops of one wave execute in order: no barrier needed).
template <int DH>
__device__ __forceinline__ void load_row_frags_lds(bf16x8_t (&f)[DH / 16], const unsigned short* __restrict__ base,
                                                   int ld, int row0, int T, int lane, char* slice) {
  constexpr int CPR = DH / 8, SB = (DH + 8) * 2, N = 32 * CPR / 64;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row0 + row < T) v = *reinterpret_cast<const u32x4_t*>(base + (size_t)(row0 + row) * ld + ch * 8);
    *reinterpret_cast<u32x4_t*>(slice + row * SB + ch * 16) = v;
  }
#pragma unroll
  for (int ks = 0; ks < DH / 16; ++ks) f[ks] = row_frag<DH>(slice, 0, ks, lane);
}

template <int DH>
__device__ __forceinline__ void store_rows_lds(const f32x16_t (&o)[DH / 32], float mul, unsigned short* __restrict__ base,
                                               int ld, int row0, int T, int lane, char* slice) {
  constexpr int CPR = DH / 8, SB = (DH + 8) * 2, N = 32 * CPR / 64;
  char* p = slice + (lane & 31) * SB + 8 * (lane >> 5);
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2_t w = {pack_bf16x2(o[dt][4 * g + 0] * mul, o[dt][4 * g + 1] * mul),
                         pack_bf16x2(o[dt][4 * g + 2] * mul, o[dt][4 * g + 3] * mul)};
      *reinterpret_cast<u32x2_t*>(p + (32 * dt + 8 * g) * 2) = w;
    }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(slice + row * SB + ch * 16);
    if (row0 + row < T) *reinterpret_cast<u32x4_t*>(base + (size_t)(row0 + row) * ld + ch * 8) = v;
  }
}

}  // namespace
