// LayerNorm forward / backward for the fusion path (nn.LayerNorm at models/fusion_layers.py:205,209).
// HBM-bound: one wavefront owns one row (d <= 2048), 16-byte loads (8 bf16 per lane per chunk),
// f32 statistics by wave butterfly reduction, no LDS in the forward.
// Algorithmic bytes per row: forward 2 x 2d (x in, y out) + 8; backward 3 x 2d (x, dy in; dx out).
#include "mmf_internal.h"
#include <stdlib.h>

namespace {

constexpr int ROWS_PER_BLOCK = 4;      // 4 waves, one row each
constexpr int MAX_CH = 4;              // 4 chunks x 64 lanes x 8 elements = 2048 columns

constexpr int BWD_BLOCK_BUDGET = 2048;  // upper bound of workgroups of a grouped backward launch (workspace rows)
// workgroups actually used (MMF_LN_BWD_BLOCKS, default below): enough rows in flight per CU to cover the HBM latency
static int bwd_blocks(bool lane_form) {
  // the lane form keeps its next row in flight, so two workgroups per CU already cover the HBM latency, and fewer partial
  // rows are left for the finalize pass (round 3, same box: 2048 / 1024 / 512 workgroups 23.5 / 20.5 / 19.9 us per
  // three-problem launch, finalize included; the chunk form 25.1 / 24.1 / 23.8)
  static const int v = [] { const char* e = getenv("MMF_LN_BWD_BLOCKS"); return e ? atoi(e) : 0; }();
  const int b = v > 0 ? v : (lane_form ? 512 : 2048);
  return b < 64 ? 64 : (b > BWD_BLOCK_BUDGET ? BWD_BLOCK_BUDGET : b);
}

struct LnArgs {
  int nprob;
  int d;
  float eps;
  float* ws;                             // backward: [total blocks][2][d] partial column sums
  int blk_start[MMF_LN_MAX_PROBLEMS + 1];
  mmf_ln_problem p[MMF_LN_MAX_PROBLEMS];
};

__device__ __forceinline__ void unpack8(const u32x4_t& w, float (&f)[8]) {
  f[0] = bf16lo(w[0]); f[1] = bf16hi(w[0]); f[2] = bf16lo(w[1]); f[3] = bf16hi(w[1]);
  f[4] = bf16lo(w[2]); f[5] = bf16hi(w[2]); f[6] = bf16lo(w[3]); f[7] = bf16hi(w[3]);
}
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
  return u32x4_t{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                 pack_bf16x2(f[6], f[7])};
}

template <int NCH>
__global__ __launch_bounds__(256)
void ln_fwd_kernel(const LnArgs a) {
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_ln_problem& P = a.p[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = ((int)blockIdx.x - a.blk_start[pi]) * ROWS_PER_BLOCK + wave;
  if (row >= P.rows) return;
  const int d = a.d;
  const unsigned short* x = static_cast<const unsigned short*>(P.x) + (size_t)row * d;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < d) {
      unpack8(*reinterpret_cast<const u32x4_t*>(x + col), v[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[c][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < d) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = v[c][e] - mean; q += t * t; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + a.eps);
  unsigned short* y = static_cast<unsigned short*>(P.y) + (size_t)row * d;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < d) {
      const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(P.gamma + col);
      const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(P.gamma + col + 4);
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(P.beta + col);
      const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(P.beta + col + 4);
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (v[c][e] - mean) * rstd * g0[e] + b0[e];
        o[e + 4] = (v[c][e + 4] - mean) * rstd * g1[e] + b1[e];
      }
      *reinterpret_cast<u32x4_t*>(y + col) = pack8(o);
    }
  }
  if (lane == 0) { P.mean[row] = mean; P.rstd[row] = rstd; }
}

// Round 3: the forward in the lane form of ln_bwd_lane_kernel below (lane l owns columns [8 l, +8) of every 512-column block and
// [512 NV + 4 l, +4) of the last 256; gamma / beta in registers instead of 6 KB of L2 reads per row; a workgroup's waves walk
// strided rows with the next row's load in flight).
template <int NV, int H8>
__global__ __launch_bounds__(256)
void ln_fwd_lane_kernel(const LnArgs a) {
  constexpr int EP = 8 * NV + 4 * H8, d = 512 * NV + 256 * H8;
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_ln_problem& P = a.p[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = a.blk_start[pi + 1] - a.blk_start[pi];
  const float inv_d = 1.f / (float)d;
  float gam[EP], bet[EP];
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4_t g = *reinterpret_cast<const f32x4_t*>(P.gamma + 512 * c + 8 * lane + 4 * h);
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(P.beta + 512 * c + 8 * lane + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gam[8 * c + 4 * h + e] = g[e]; bet[8 * c + 4 * h + e] = b[e]; }
    }
  if (H8) {
    const f32x4_t g = *reinterpret_cast<const f32x4_t*>(P.gamma + 512 * NV + 4 * lane);
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(P.beta + 512 * NV + 4 * lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gam[8 * NV + e] = g[e]; bet[8 * NV + e] = b[e]; }
  }
  struct Row { u32x4_t x[NV > 0 ? NV : 1]; u32x2_t x8; };
  auto fetch = [&](int row, Row& r) {
    const unsigned short* x = static_cast<const unsigned short*>(P.x) + (size_t)row * d;
#pragma unroll
    for (int c = 0; c < NV; ++c) r.x[c] = *reinterpret_cast<const u32x4_t*>(x + 512 * c + 8 * lane);
    if (H8) r.x8 = *reinterpret_cast<const u32x2_t*>(x + 512 * NV + 4 * lane);
  };
  const int step = nblk * ROWS_PER_BLOCK;
  int row = ((int)blockIdx.x - a.blk_start[pi]) * ROWS_PER_BLOCK + wave;
  Row cur, nxt;
  if (row < P.rows) fetch(row, cur);
  for (; row < P.rows; row += step) {
    const bool more = row + step < P.rows;
    if (more) fetch(row + step, nxt);
    float v[EP];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      float t[8];
      unpack8(cur.x[c], t);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[8 * c + e] = t[e];
    }
    if (H8) { v[8 * NV] = bf16lo(cur.x8[0]); v[8 * NV + 1] = bf16hi(cur.x8[0]); v[8 * NV + 2] = bf16lo(cur.x8[1]); v[8 * NV + 3] = bf16hi(cur.x8[1]); }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EP; ++e) s += v[e];
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EP; ++e) { v[e] -= mean; q += v[e] * v[e]; }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + a.eps);
    unsigned short* y = static_cast<unsigned short*>(P.y) + (size_t)row * d;
    float o[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) o[e] = v[e] * rstd * gam[e] + bet[e];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const u32x4_t w = {pack_bf16x2(o[8 * c], o[8 * c + 1]), pack_bf16x2(o[8 * c + 2], o[8 * c + 3]),
                         pack_bf16x2(o[8 * c + 4], o[8 * c + 5]), pack_bf16x2(o[8 * c + 6], o[8 * c + 7])};
      *reinterpret_cast<u32x4_t*>(y + 512 * c + 8 * lane) = w;
    }
    if (H8) {
      const u32x2_t w = {pack_bf16x2(o[8 * NV], o[8 * NV + 1]), pack_bf16x2(o[8 * NV + 2], o[8 * NV + 3])};
      *reinterpret_cast<u32x2_t*>(y + 512 * NV + 4 * lane) = w;
    }
    if (lane == 0) { P.mean[row] = mean; P.rstd[row] = rstd; }
    if (more) cur = nxt;
  }
}

// Backward: a block owns a strided set of rows of ONE problem; each wave walks its rows, keeps the
// dgamma/dbeta partial sums of its columns in registers, the 4 waves combine through LDS and the
// block issues one f32 atomic per column (Guideline 12: reduce on chip, then one atomic per block).
template <int NCH>
__global__ __launch_bounds__(256)
void ln_bwd_kernel(const LnArgs a) {
  __shared__ float red[2][3][NCH * 512];          // [dgamma|dbeta][waves 1..3][column]
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_ln_problem& P = a.p[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = a.blk_start[pi + 1] - a.blk_start[pi];
  const int d = a.d;
  float dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[c][e] = 0.f; db[c][e] = 0.f; }

  for (int row = ((int)blockIdx.x - a.blk_start[pi]) * ROWS_PER_BLOCK + wave; row < P.rows;
       row += nblk * ROWS_PER_BLOCK) {
    const unsigned short* x = static_cast<const unsigned short*>(P.x) + (size_t)row * d;
    const unsigned short* dy = static_cast<const unsigned short*>(P.dy) + (size_t)row * d;
    const float mean = P.mean[row], rstd = P.rstd[row];
    float xh[NCH][8], g[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < d) {
        float xv[8], dv[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(x + col), xv);
        unpack8(*reinterpret_cast<const u32x4_t*>(dy + col), dv);
        const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(P.gamma + col);
        const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(P.gamma + col + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gam = e < 4 ? g0[e] : g1[e - 4];
          xh[c][e] = (xv[e] - mean) * rstd;
          g[c][e] = dv[e] * gam;
          s1 += g[c][e];
          s2 += g[c][e] * xh[c][e];
          dg[c][e] += dv[e] * xh[c][e];
          db[c][e] += dv[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[c][e] = 0.f; g[c][e] = 0.f; }
      }
    }
    const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
    unsigned short* dx = static_cast<unsigned short*>(P.dx) + (size_t)row * d;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < d) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (g[c][e] - c1 - xh[c][e] * c2);
        *reinterpret_cast<u32x4_t*>(dx + col) = pack8(o);
      }
    }
  }
  // combine the 4 waves' column sums, then one atomic per column per block
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[0][wave - 1][(c * 64 + lane) * 8 + e] = dg[c][e];
        red[1][wave - 1][(c * 64 + lane) * 8 + e] = db[c][e];
      }
  }
  __syncthreads();
  if (wave == 0) {                                 // this workgroup's partials -> workspace row
    float* wsg = a.ws + (size_t)blockIdx.x * 2 * d;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < d) {
        f32x4_t g0, g1, b0, b1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float sg = dg[c][e], sb = db[c][e];
#pragma unroll
          for (int w = 0; w < 3; ++w) { sg += red[0][w][col + e]; sb += red[1][w][col + e]; }
          if (e < 4) { g0[e] = sg; b0[e] = sb; } else { g1[e - 4] = sg; b1[e - 4] = sb; }
        }
        *reinterpret_cast<f32x4_t*>(wsg + col) = g0;
        *reinterpret_cast<f32x4_t*>(wsg + col + 4) = g1;
        *reinterpret_cast<f32x4_t*>(wsg + d + col) = b0;
        *reinterpret_cast<f32x4_t*>(wsg + d + col + 4) = b1;
      }
    }
  }
}

// Round 3: the same backward for d = 512 NV + 256 H8 (768 = 512 + 256, 512, 1024, 256 ...) with every lane busy and the
// next row in flight.  ln_bwd_kernel above gives lane l the 16-byte chunks l, l + 64, ...: at d = 768 the second chunk only
// exists for lanes 0-31 (a quarter of the loads and of the arithmetic runs half empty), gamma is re-read from L2 for every
// row (as many bytes as x and dy together), and a wave asks for a row only after it has reduced and stored the one before
// (profiles/r03_layernorm.txt: 2.8 TB/s of algorithmic bytes for the three-problem launches, finalize included).  Here lane l
// owns columns [8 l, 8 l + 8) of each 512-column block (16-byte loads) and [512 NV + 4 l, + 4) of the last 256 (8-byte
// loads), gamma stays in registers, and the loads of the wave's next row are issued before the reductions of the current.
template <int NV, int H8>
__global__ __launch_bounds__(256)
void ln_bwd_lane_kernel(const LnArgs a) {
  constexpr int EP = 8 * NV + 4 * H8;                 // elements per lane
  __shared__ float red[2][3][64 * EP];                // [dgamma|dbeta][waves 1..3][lane-major element]
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_ln_problem& P = a.p[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = a.blk_start[pi + 1] - a.blk_start[pi];
  constexpr int d = 512 * NV + 256 * H8;
  const float inv_d = 1.f / (float)d;
  float gam[EP], dg[EP], db[EP];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(P.gamma + 512 * c + 8 * lane);
    const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(P.gamma + 512 * c + 8 * lane + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gam[8 * c + e] = g0[e]; gam[8 * c + 4 + e] = g1[e]; }
  }
  if (H8) {
    const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(P.gamma + 512 * NV + 4 * lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) gam[8 * NV + e] = g0[e];
  }
#pragma unroll
  for (int e = 0; e < EP; ++e) { dg[e] = 0.f; db[e] = 0.f; }

  struct Row { u32x4_t x[NV > 0 ? NV : 1], y[NV > 0 ? NV : 1]; u32x2_t x8, y8; float mean, rstd; };
  auto fetch = [&](int row, Row& r) {
    const unsigned short* x = static_cast<const unsigned short*>(P.x) + (size_t)row * d;
    const unsigned short* dy = static_cast<const unsigned short*>(P.dy) + (size_t)row * d;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      r.x[c] = *reinterpret_cast<const u32x4_t*>(x + 512 * c + 8 * lane);
      r.y[c] = *reinterpret_cast<const u32x4_t*>(dy + 512 * c + 8 * lane);
    }
    if (H8) {
      r.x8 = *reinterpret_cast<const u32x2_t*>(x + 512 * NV + 4 * lane);
      r.y8 = *reinterpret_cast<const u32x2_t*>(dy + 512 * NV + 4 * lane);
    }
    r.mean = P.mean[row];
    r.rstd = P.rstd[row];
  };
  const int step = nblk * ROWS_PER_BLOCK;
  int row = ((int)blockIdx.x - a.blk_start[pi]) * ROWS_PER_BLOCK + wave;
  Row cur, nxt;
  if (row < P.rows) fetch(row, cur);
  for (; row < P.rows; row += step) {
    const bool more = row + step < P.rows;
    if (more) fetch(row + step, nxt);                  // in flight under this row's reductions and store
    float xv[EP], dv[EP];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      float t[8];
      unpack8(cur.x[c], t);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[8 * c + e] = t[e];
      unpack8(cur.y[c], t);
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[8 * c + e] = t[e];
    }
    if (H8) {
      xv[8 * NV + 0] = bf16lo(cur.x8[0]); xv[8 * NV + 1] = bf16hi(cur.x8[0]); xv[8 * NV + 2] = bf16lo(cur.x8[1]); xv[8 * NV + 3] = bf16hi(cur.x8[1]);
      dv[8 * NV + 0] = bf16lo(cur.y8[0]); dv[8 * NV + 1] = bf16hi(cur.y8[0]); dv[8 * NV + 2] = bf16lo(cur.y8[1]); dv[8 * NV + 3] = bf16hi(cur.y8[1]);
    }
    const float mean = cur.mean, rstd = cur.rstd;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      xv[e] = (xv[e] - mean) * rstd;                   // x hat
      const float g = dv[e] * gam[e];
      s1 += g;
      s2 += g * xv[e];
      dg[e] += dv[e] * xv[e];
      db[e] += dv[e];
      dv[e] = g;
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
    unsigned short* dx = static_cast<unsigned short*>(P.dx) + (size_t)row * d;
    float o[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) o[e] = rstd * (dv[e] - c1 - xv[e] * c2);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const u32x4_t w = {pack_bf16x2(o[8 * c], o[8 * c + 1]), pack_bf16x2(o[8 * c + 2], o[8 * c + 3]),
                         pack_bf16x2(o[8 * c + 4], o[8 * c + 5]), pack_bf16x2(o[8 * c + 6], o[8 * c + 7])};
      *reinterpret_cast<u32x4_t*>(dx + 512 * c + 8 * lane) = w;
    }
    if (H8) {
      const u32x2_t w = {pack_bf16x2(o[8 * NV], o[8 * NV + 1]), pack_bf16x2(o[8 * NV + 2], o[8 * NV + 3])};
      *reinterpret_cast<u32x2_t*>(dx + 512 * NV + 4 * lane) = w;
    }
    if (more) cur = nxt;
  }
  // combine the 4 waves' column sums; this workgroup's partials -> its workspace row (column order)
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < EP; ++e) { red[0][wave - 1][e * 64 + lane] = dg[e]; red[1][wave - 1][e * 64 + lane] = db[e]; }
  }
  __syncthreads();
  if (wave == 0) {
    float* wsg = a.ws + (size_t)blockIdx.x * 2 * d;
#pragma unroll
    for (int e = 0; e < EP; ++e)
#pragma unroll
      for (int w = 0; w < 3; ++w) { dg[e] += red[0][w][e * 64 + lane]; db[e] += red[1][w][e * 64 + lane]; }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      *reinterpret_cast<f32x4_t*>(wsg + 512 * c + 8 * lane) = f32x4_t{dg[8 * c], dg[8 * c + 1], dg[8 * c + 2], dg[8 * c + 3]};
      *reinterpret_cast<f32x4_t*>(wsg + 512 * c + 8 * lane + 4) = f32x4_t{dg[8 * c + 4], dg[8 * c + 5], dg[8 * c + 6], dg[8 * c + 7]};
      *reinterpret_cast<f32x4_t*>(wsg + d + 512 * c + 8 * lane) = f32x4_t{db[8 * c], db[8 * c + 1], db[8 * c + 2], db[8 * c + 3]};
      *reinterpret_cast<f32x4_t*>(wsg + d + 512 * c + 8 * lane + 4) = f32x4_t{db[8 * c + 4], db[8 * c + 5], db[8 * c + 6], db[8 * c + 7]};
    }
    if (H8) {
      *reinterpret_cast<f32x4_t*>(wsg + 512 * NV + 4 * lane) = f32x4_t{dg[8 * NV], dg[8 * NV + 1], dg[8 * NV + 2], dg[8 * NV + 3]};
      *reinterpret_cast<f32x4_t*>(wsg + d + 512 * NV + 4 * lane) = f32x4_t{db[8 * NV], db[8 * NV + 1], db[8 * NV + 2], db[8 * NV + 3]};
    }
  }
}

// second phase: dgamma[j] += sum over the problem's workgroups of their partials.  thread = column,
// blockIdx.z = one of FIN_SLICES slices of the workgroup range; each slice ends in one f32 atomic per
// column (FIN_SLICES adders per address).
constexpr int FIN_SLICES = 64;
__global__ __launch_bounds__(256)
void ln_bwd_finalize_kernel(const LnArgs a) {
  const int pi = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= a.d) return;
  const int b0 = a.blk_start[pi], nb = a.blk_start[pi + 1] - b0;
  const int per = (nb + FIN_SLICES - 1) / FIN_SLICES;
  const int lo = b0 + blockIdx.z * per, hi = min(b0 + nb, lo + per);
  if (lo >= hi) return;
  // four independent row loads in flight per thread: written as one running sum the loop is a chain of exposed HBM round
  // trips (16 us per launch for 12 MB of partials, measured in round 2)
  float sg[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  int b = lo;
  for (; b + 4 <= hi; b += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* w = a.ws + (size_t)(b + u) * 2 * a.d;
      sg[u] += w[col];
      sb[u] += w[a.d + col];
    }
  }
  for (; b < hi; ++b) {
    const float* w = a.ws + (size_t)b * 2 * a.d;
    sg[0] += w[col];
    sb[0] += w[a.d + col];
  }
  atomicAdd(a.p[pi].dgamma + col, (sg[0] + sg[1]) + (sg[2] + sg[3]));
  atomicAdd(a.p[pi].dbeta + col, (sb[0] + sb[1]) + (sb[2] + sb[3]));
}

int check_common(const char* who, const mmf_ln_problem* p, int n, int d) {
  if (!p || n <= 0 || n > MMF_LN_MAX_PROBLEMS) MMF_FAIL(MMF_E_SHAPE, "%s: num_problems=%d out of range", who, n);
  if (d <= 0 || (d & 7) || d > MAX_CH * 512) MMF_FAIL(MMF_E_SHAPE, "%s: d=%d must be a multiple of 8, <= 2048", who, d);
  return MMF_OK;
}

}  // namespace

extern "C" int mmf_layernorm_fwd_grouped(const mmf_ln_problem* problems, int num_problems, int d,
                                         float eps, void* stream) {
  if (int rc = check_common("mmf_layernorm_fwd_grouped", problems, num_problems, d)) return rc;
  LnArgs a; a.nprob = num_problems; a.d = d; a.eps = eps; a.ws = nullptr;
  int total = 0;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_ln_problem& p = problems[i];
    if (p.rows <= 0 || !p.x || !p.y || !p.gamma || !p.beta || !p.mean || !p.rstd)
      MMF_FAIL(MMF_E_SHAPE, "mmf_layernorm_fwd_grouped[%d]: null operand or rows=%d", i, p.rows);
    if (!mmf_aligned16(p.x) || !mmf_aligned16(p.y) || !mmf_aligned16(p.gamma) || !mmf_aligned16(p.beta))
      MMF_FAIL(MMF_E_ALIGN, "mmf_layernorm_fwd_grouped[%d]: pointers must be 16-byte aligned", i);
    a.blk_start[i] = total;
    total += (p.rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    a.p[i] = p;
  }
  a.blk_start[num_problems] = total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const int fwd_lane = [] { const char* e = getenv("MMF_LN_FWD_LANE"); return e ? atoi(e) : 1; }();
  static const int fwd_blocks_env = [] { const char* e = getenv("MMF_LN_FWD_BLOCKS"); return e ? atoi(e) : 0; }();
  long long total_rows = 0;
  for (int i = 0; i < num_problems; ++i) total_rows += problems[i].rows;
  // (round 3, same box, chunk form / lane form: three problems of 15,072 rows 13.0 / 9.6 us with 1024 workgroups; all six,
  // 30,144 rows, 20.7 / 16.4 us with 2048 — profiles/r03_layernorm.txt)
  const int fwd_blocks = fwd_blocks_env > 0 ? fwd_blocks_env : (total_rows > 20000 ? 2048 : 1024);
  if (fwd_lane && (d == 768 || d == 512 || d == 256 || d == 1024) && total > fwd_blocks) {
    // lane form: a bounded number of workgroups shared out over the problems in proportion to their rows
    int t2 = 0;
    for (int i = 0; i < num_problems; ++i) {
      const int rows = problems[i].rows, full = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
      int nb = (int)(((long long)rows * fwd_blocks + total_rows - 1) / total_rows);
      nb = nb > full ? full : (nb < 1 ? 1 : nb);
      a.blk_start[i] = t2;
      t2 += nb;
    }
    a.blk_start[num_problems] = t2;
    if (d == 768)       hipLaunchKernelGGL((ln_fwd_lane_kernel<1, 1>), dim3(t2), dim3(256), 0, s, a);
    else if (d == 512)  hipLaunchKernelGGL((ln_fwd_lane_kernel<1, 0>), dim3(t2), dim3(256), 0, s, a);
    else if (d == 256)  hipLaunchKernelGGL((ln_fwd_lane_kernel<0, 1>), dim3(t2), dim3(256), 0, s, a);
    else                hipLaunchKernelGGL((ln_fwd_lane_kernel<2, 0>), dim3(t2), dim3(256), 0, s, a);
    MMF_CHECK_LAUNCH("mmf_layernorm_fwd_grouped(lane)");
    return MMF_OK;
  }
  const int nch = (d + 511) / 512;
  switch (nch) {
    case 1: hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(total), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(total), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(ln_fwd_kernel<3>, dim3(total), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(total), dim3(256), 0, s, a); break;
  }
  MMF_CHECK_LAUNCH("mmf_layernorm_fwd_grouped");
  return MMF_OK;
}

extern "C" size_t mmf_layernorm_bwd_workspace_bytes(int d) {
  return (size_t)(BWD_BLOCK_BUDGET + MMF_LN_MAX_PROBLEMS) * 2 * (size_t)(d > 0 ? d : 0) * sizeof(float);
}

extern "C" int mmf_layernorm_bwd_grouped(const mmf_ln_problem* problems, int num_problems, int d,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_common("mmf_layernorm_bwd_grouped", problems, num_problems, d)) return rc;
  if (!workspace || workspace_bytes < mmf_layernorm_bwd_workspace_bytes(d) || !mmf_aligned16(workspace))
    MMF_FAIL(MMF_E_SHAPE, "mmf_layernorm_bwd_grouped: workspace of %zu bytes (16-byte aligned) required",
             mmf_layernorm_bwd_workspace_bytes(d));
  LnArgs a; a.nprob = num_problems; a.d = d; a.eps = 0.f; a.ws = static_cast<float*>(workspace);
  int total = 0;
  long long total_rows = 0;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_ln_problem& p = problems[i];
    if (p.rows <= 0 || !p.x || !p.dy || !p.dx || !p.gamma || !p.mean || !p.rstd || !p.dgamma || !p.dbeta)
      MMF_FAIL(MMF_E_SHAPE, "mmf_layernorm_bwd_grouped[%d]: null operand or rows=%d", i, p.rows);
    if (!mmf_aligned16(p.x) || !mmf_aligned16(p.dy) || !mmf_aligned16(p.dx) || !mmf_aligned16(p.gamma))
      MMF_FAIL(MMF_E_ALIGN, "mmf_layernorm_bwd_grouped[%d]: pointers must be 16-byte aligned", i);
    a.p[i] = p;
    total_rows += p.rows;
  }
  // Workgroups are shared out over the problems in proportion to their rows (~2 per CU in total);
  // each leaves one row of column partials in the workspace (same-row float atomics would run ~14x
  // below the streaming rate: MI355X_MICROARCH.md, Global float atomics), summed by the finalize pass.
  static const int lane_form = [] { const char* e = getenv("MMF_LN_BWD_LANE"); return e ? atoi(e) : 1; }();
  const bool use_lane = lane_form && (d == 768 || d == 512 || d == 256 || d == 1024);
  const int budget = bwd_blocks(use_lane);
  for (int i = 0; i < num_problems; ++i) {
    const int rows = problems[i].rows;
    int nb = (int)(((long long)rows * budget + total_rows - 1) / total_rows);
    const int full = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (nb > full) nb = full;
    if (nb < 1) nb = 1;
    a.blk_start[i] = total;
    total += nb;
  }
  a.blk_start[num_problems] = total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nch = (d + 511) / 512;
  if (use_lane) {
    if (d == 768)       hipLaunchKernelGGL((ln_bwd_lane_kernel<1, 1>), dim3(total), dim3(256), 0, s, a);
    else if (d == 512)  hipLaunchKernelGGL((ln_bwd_lane_kernel<1, 0>), dim3(total), dim3(256), 0, s, a);
    else if (d == 256)  hipLaunchKernelGGL((ln_bwd_lane_kernel<0, 1>), dim3(total), dim3(256), 0, s, a);
    else                hipLaunchKernelGGL((ln_bwd_lane_kernel<2, 0>), dim3(total), dim3(256), 0, s, a);
    MMF_CHECK_LAUNCH("mmf_layernorm_bwd_grouped(lane)");
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((d + 255) / 256, num_problems, FIN_SLICES), dim3(256), 0, s, a);
    MMF_CHECK_LAUNCH("mmf_layernorm_bwd_grouped(finalize)");
    return MMF_OK;
  }
  switch (nch) {
    case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(total), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(total), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(ln_bwd_kernel<3>, dim3(total), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(total), dim3(256), 0, s, a); break;
  }
  MMF_CHECK_LAUNCH("mmf_layernorm_bwd_grouped");
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((d + 255) / 256, num_problems, FIN_SLICES), dim3(256), 0, s, a);
  MMF_CHECK_LAUNCH("mmf_layernorm_bwd_grouped(finalize)");
  return MMF_OK;
}
