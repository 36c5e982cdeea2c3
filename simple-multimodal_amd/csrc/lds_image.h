// The one LDS image of a [rows][W] bf16 tile (rows a multiple of 8, W a multiple of 32 columns) used by the attention kernels
// (attention2.hip: W = head dimension) and by the 128 x 128-wave-tile GEMM (gemm6.hip: W = k-depth of a stage for operands whose
// reduction index is contiguous in memory, W = 256 for operands whose reduction index is the memory row): 8-row x 32-column
// subtiles of 512 B, guide T10 image (a),
//     off(row, ch) = (W/32)*512*(row >> 3) + 512*(ch >> 2) + 64*(row & 7) + 16*((ch & 3) ^ ((row >> 2) & 3))
// for 16-byte chunk ch of row `row`.  Row reads (ds_read_b128, MFMA 32x32x16 operand whose 32-index is the tile row) take the
// floor of 4 LDS cycles and transposed reads (ds_read_b64_tr_b16, 32-index = tile column) the floor of 2, for W = 32 / 64 / 96 /
// 256 (tools/lds_image_check.py simulates both lane by lane against the MI355X bank rules).  No padding: consecutive subtiles are
// consecutive 512-B blocks, so a 1-KiB LDS-DMA piece is two subtiles and the swizzle lives in the per-lane SOURCE offset.
#pragma once

namespace {
template <int W>
__device__ __forceinline__ constexpr int img_off(int row, int ch) {
  return (W / 32) * 512 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
}  // namespace
