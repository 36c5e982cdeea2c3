// Shared device helpers of the attention kernels (attention2.hip): MFMA v_mfma_f32_32x32x16_bf16 fragment layouts and
// the ONE LDS image every tile uses (round 3).
//
// LDS image of a [rows][DH] bf16 tile (rows a multiple of 8): 8-row x 32-column subtiles of 512 B, guide T10 image (a),
//     off(row, ch) = (DH/32)*512*(row >> 3) + 512*(ch >> 2) + 64*(row & 7) + 16*((ch & 3) ^ ((row >> 2) & 3))
// for 16-byte chunk ch of row `row`.  Both kinds of read the kernels issue are conflict-free on it: the row read
// (ds_read_b128, an MFMA operand whose 32-index is the tile row) in 4 LDS cycles and the transposed read
// (ds_read_b64_tr_b16, an operand whose 32-index is the tile COLUMN) in 2.  Rounds 1-2 used 16-byte padded rows
// ([rows][DH+8]): conflict-free for the row reads only — every transposed read was 2-way (4 cycles; simulated lane by
// lane against the MI355X bank rules in tools/lds_image_check.py, and measured: SQ_LDS_BANK_CONFLICT 26-29 % of the
// LDS-active cycles of the round-2 backward kernels), and with about 1 KiB of fragment reads per MFMA the LDS port is as
// busy as the matrix pipe in all three kernels.  The image has no padding: a 64-row tile is 64*DH*2 bytes = DH/8 one-KiB
// LDS-DMA pieces exactly, and a wave-private 32-row slice (prologue loads, epilogue stores) is a quarter of a stage.
#pragma once
#include "mmf_internal.h"
#include "lds_image.h"

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

// ---- the LDS image: img_off<DH>(row, ch) in lds_image.h (shared with gemm6.hip) ------------------------------------
template <int DH> constexpr int img_tile_bytes() { return 64 * DH * 2; }     // one 64-row tile
template <int DH> constexpr int img_slice_bytes() { return 32 * DH * 2; }    // one wave's 32-row slice

// Row fragment (MFMA 32x32x16 A or B operand whose 32-index is the tile ROW): lane l gets
// tile[row0 + (l & 31)][16 ks + 8 (l >> 5) + 0..7].  row0 is a multiple of 32.
template <int DH>
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(tile + img_off<DH>(row0 + (lane & 31), 2 * ks + (lane >> 5)));
}

// This lane's part of a transposed-read address.  The transposed fragment (A operand X^T[i = column][k = row]) of the 16
// tile rows [16 G, 16 G + 16) and the 32 columns [32 D, 32 D + 32) is two ds_read_b64_tr_b16: element j of lane
// (i = l & 31, h = l >> 5) is tile[16 G + 8 (j >> 2) + 4 h + (j & 3)][32 D + i] — the k order in which an accumulator
// tile presents its rows when it is the other operand.  Lane 4q+p of 16-lane group (h, g) supplies the address of row
// 16 G + 4 h + q (+ 8 for the second read), columns 32 D + 16 g + 4 p .. + 3:
//     lo = tr_lane_lo(lane) + img_tr_imm(G, D),   hi = tr_lane_hi(lane) + img_tr_imm(G, D) + (DH/32)*512
// (the swizzle term (row >> 2) & 3 is h for the first read and h ^ 2 for the second, hence two lane parts).
__device__ __forceinline__ unsigned tr_lane_lo(int lane) {
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  return (unsigned)(64 * (4 * h + q) + 16 * ((2 * g + (p >> 1)) ^ h) + 8 * (p & 1));
}
__device__ __forceinline__ unsigned tr_lane_hi(int lane) {
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  return (unsigned)(64 * (4 * h + q) + 16 * ((2 * g + (p >> 1)) ^ (h ^ 2)) + 8 * (p & 1));
}
template <int DH, int G, int D> constexpr int img_tr_imm() { return (DH / 32) * 512 * (2 * G) + 512 * D; }

// registers 8s..8s+7 of a 32x32 accumulator as a bf16 operand fragment
__device__ __forceinline__ bf16x8_t acc_frag(const f32x16_t& x, int s) {
  const u32x4_t w = {pack_bf16x2(x[8 * s + 0], x[8 * s + 1]), pack_bf16x2(x[8 * s + 2], x[8 * s + 3]),
                     pack_bf16x2(x[8 * s + 4], x[8 * s + 5]), pack_bf16x2(x[8 * s + 6], x[8 * s + 7])};
  return __builtin_bit_cast(bf16x8_t, w);
}

// ---- whole-row HBM access through a wave-private LDS slice ------------------------------------------
// Fragment-shaped loads / stores (lane = row) touch 32 rows x 32 B (loads) or 32 rows x 8 B (stores) per
// instruction: 32+ cache lines each, and the 8-byte pieces make partial-line writes.  For small
// attention problems that prologue/epilogue is most of the kernel.  Instead a wave moves its 32 rows as
// 16-byte chunks in row-major order (DH/8 lanes per row) and converts to/from the MFMA fragment
// layout in a 32-row slice of the image it owns (LDS ops of one wave execute in order: no barrier needed).
//
// RowLoad: through a buffer descriptor whose range ends after row T - 1 (rows >= T read as zeros from the range check),
// so there is no lane-dependent branch and hipcc issues all 32 * CPR / 64 loads back to back and waits once.  The
// round-1/2 predicated form put every load in its own basic block with `s_waitcnt vmcnt(0)` in front of its ds_write —
// 6 (forward: Q), 12 (dK/dV: K, V) or 18 (dQ: Q, dO, O) SERIAL HBM round trips per wave before its first MFMA (round 3
// finding in the disassembly; same-box A/B of the two forms: backward cross x6 187.8 -> 160.3 us, self x3 119.2 -> 107.5,
// forward cross x6 55.7 -> 52.7).  issue / commit are split so that several matrices' loads are in flight together.
template <int DH>
struct RowLoad {
  static constexpr int CPR = DH / 8, N = 32 * CPR / 64;
  u32x4_t v[N];
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, int ld, int row0, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
      v[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (row0 + row) * ld * 2 + ch * 16, 0, 0));
    }
  }
  // write the rows into the wave-private slice (image of a 32-row tile)
  __device__ __forceinline__ void store(int lane, char* slice) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
      *reinterpret_cast<u32x4_t*>(slice + img_off<DH>(row, ch)) = v[i];
    }
  }
  // ... and read them back as MFMA row fragments
  __device__ __forceinline__ void commit(bf16x8_t (&f)[DH / 16], int lane, char* slice) const {
    store(lane, slice);
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) f[ks] = row_frag<DH>(slice, 0, ks, lane);
  }
};

// store a [d][q or key] accumulator set as rows of a (T, ld) bf16 matrix: lane owns row (row0 + (l&31)),
// register 4g+i of tile dt is column 32 dt + 8 g + 4 (l>>5) + i.
template <int DH>
__device__ __forceinline__ void store_rows_lds(const f32x16_t (&o)[DH / 32], float mul, unsigned short* __restrict__ base,
                                               int ld, int row0, int T, int lane, char* slice) {
  constexpr int CPR = DH / 8, N = 32 * CPR / 64;
  const int r = lane & 31, half = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2_t w = {pack_bf16x2(o[dt][4 * g + 0] * mul, o[dt][4 * g + 1] * mul),
                         pack_bf16x2(o[dt][4 * g + 2] * mul, o[dt][4 * g + 3] * mul)};
      *reinterpret_cast<u32x2_t*>(slice + img_off<DH>(r, 4 * dt + g) + 8 * half) = w;
    }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(slice + img_off<DH>(row, ch));
    if (row0 + row < T) *reinterpret_cast<u32x4_t*>(base + (size_t)(row0 + row) * ld + ch * 8) = v;
  }
}

}  // namespace
