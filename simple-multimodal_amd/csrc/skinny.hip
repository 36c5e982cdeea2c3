// Skinny-M linear layers (M <= 64 rows: the (B, d) branches of the hierarchical fusion — Early, Contrastive,
// Adaptive, Graph, meta MLPs, reference models/fusion_layers.py:21-28,304-327,395-412,471-476 — and the
// pooled projections of MulT).  These are weight-streaming problems: 2 bytes of W per M*2 FLOP, so the
// 256-row GEMM tile wastes >90 % of its work on them.  Here the WEIGHT tile rides the MFMA row axis
// (v_mfma_f32_16x16x32_bf16, D[i = n or k_in][j = m]) and all of M (1..4 column tiles of 16) is the other
// operand; a workgroup owns one strip of output columns, its 4 waves split the reduction and combine in LDS.
//
//   forward  y[m][n]   = act(sum_k x[m][k] W[n][k] + b[n])        W rows are k-contiguous: fragments straight
//                                                                 from HBM/L2, no LDS staging at all;
//   dgrad    dx[m][k]  = (sum_n dy[m][n] W[n][k]) * (aux>0) * alpha   the reduction index is W's ROW: each wave
//                                                                 stages 32 x 64 blocks of W in a private LDS
//                                                                 slice and reads them back transposed
//                                                                 (ds_read_b64_tr_b16).
// wgrad (dW = dy^T x, output-bound) stays on the grouped TN kernel via the deferred launch.
#include "mmf_internal.h"
#include <stdlib.h>

namespace {

constexpr int SK_THREADS = 512;               // 8 waves split the reduction
constexpr int SK_WAVES = SK_THREADS / 64;
constexpr int MAX_MT = 4;                  // up to 64 rows

struct SkinnyArgs {
  int nprob;
  int flags;                               // MMF_EPI_BIAS | MMF_EPI_RELU | MMF_EPI_MASK_AUX | MMF_EPI_DROPOUT
  float alpha;
  // round 4 (mmf_skinny_linear_*_ex): the casts, dropout and ReLU-gradient launches around a (B, d)-row linear folded into it
  int x_f32;                               // X (forward) / dY (dgrad) is f32: narrowed while the fragment is loaded
  int gate_f32;                            // dgrad: the gate tensor is f32
  float gate_scale;                        // dgrad: dz = dy * (gate > 0) * gate_scale
  unsigned drop_thresh, site;              // MMF_EPI_DROPOUT: forward y = dropout(act(.)); dgrad dz = dy * keep / (1 - p)
  float drop_scale;
  const unsigned long long* rng_state;
  int blk_start[MMF_SKINNY_MAX_PROBLEMS + 1];
  mmf_skinny_problem p[MMF_SKINNY_MAX_PROBLEMS];
  void* y2[MMF_SKINNY_MAX_PROBLEMS];       // forward: second copy of the output in the other dtype, or null
  const void* gate[MMF_SKINNY_MAX_PROBLEMS];   // dgrad: the forward's saved output [M][N], or null
  void* dz[MMF_SKINNY_MAX_PROBLEMS];       // dgrad: the gated gradient as bf16 (the weight gradient's operand), or null;
                                           // forward with an f32 X: the narrowed input as bf16 [M][lddz] (the same operand's other side)
  int ldy2[MMF_SKINNY_MAX_PROBLEMS], ldgate[MMF_SKINNY_MAX_PROBLEMS], lddz[MMF_SKINNY_MAX_PROBLEMS];
};

// eight consecutive elements of row `row` at column `col` of a bf16 or f32 [rows][ld] matrix as f32 (zeros outside)
__device__ __forceinline__ void load8_f32(const void* base, bool f32, int ld, int row, int rows, int col, int cols, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (row >= rows || col >= cols) return;
  if (f32) {
    const float* p = static_cast<const float*>(base) + (size_t)row * ld + col;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else {
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(static_cast<const unsigned short*>(base) + (size_t)row * ld + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf16lo(w[e]); v[2 * e + 1] = bf16hi(w[e]); }
  }
}
__device__ __forceinline__ bf16x8_t pack8(const float (&v)[8]) {
  const u32x4_t w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  return __builtin_bit_cast(bf16x8_t, w);
}

__device__ __forceinline__ bf16x8_t load_frag_rows(const unsigned short* __restrict__ base, int ld, int row, int rows,
                                                   int col, int cols) {
  u32x4_t v = {0u, 0u, 0u, 0u};
  if (row < rows && col < cols) v = *reinterpret_cast<const u32x4_t*>(base + (size_t)row * ld + col);
  return __builtin_bit_cast(bf16x8_t, v);
}

// ---- forward ---------------------------------------------------------------------------------------------
template <bool OUT_F32>
__global__ __launch_bounds__(SK_THREADS)
void skinny_fwd_kernel(const SkinnyArgs a) {
  __shared__ __attribute__((aligned(16))) float red[SK_WAVES - 1][MAX_MT][64][4];
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_skinny_problem& P = a.p[pi];
  const int M = P.M, N = P.N, K = P.K;
  const int n0 = ((int)blockIdx.x - a.blk_start[pi]) * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = (M + 15) >> 4;
  const unsigned short* __restrict__ X = static_cast<const unsigned short*>(P.X);
  const unsigned short* __restrict__ W = static_cast<const unsigned short*>(P.W);
  unsigned short* x16 = static_cast<unsigned short*>(a.dz[pi]);

  // this wave's share of K, in whole 32-element MFMA steps; the next step's fragments are requested
  // before the current step's MFMAs (the chain is otherwise one HBM/L2 latency per step)
  const int steps = (K + 31) >> 5, per = (steps + SK_WAVES - 1) / SK_WAVES;
  const int s0 = wave * per, s1 = min(steps, s0 + per);
  f32x4_t acc[MAX_MT];
#pragma unroll
  for (int t = 0; t < MAX_MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int wrow = n0 + (lane & 15), kc = (lane >> 4) << 3;
  // A wave's steps form a latency chain (few workgroups, one wave per SIMD at most): THREE steps' fragments are in flight, each requested
  // two steps before its MFMAs.  Round 2-3 kept one step ahead: a 3840-deep reduction (the meta MLP: 15 steps per wave) then paid ~1.5 us per step.
  struct Frags { bf16x8_t w, x[MAX_MT]; };
  auto fetch = [&](int s, Frags& f) {
    const int k = (s << 5) + kc;
    f.w = load_frag_rows(W, P.ldw, wrow, N, k, K);
#pragma unroll
    for (int t = 0; t < MAX_MT; ++t)
      if (t < mt) {
        if (a.x_f32) {
          float v[8];
          const int m = t * 16 + (lane & 15);
          load8_f32(P.X, true, P.ldx, m, M, k, K, v);
          f.x[t] = pack8(v);
          if (x16 && n0 == 0 && m < M && k < K)              // the first strip's workgroup leaves the bf16 copy behind (wgrad operand)
            *reinterpret_cast<u32x4_t*>(x16 + (size_t)m * a.lddz[pi] + k) = __builtin_bit_cast(u32x4_t, f.x[t]);
        } else {
          f.x[t] = load_frag_rows(X, P.ldx, t * 16 + (lane & 15), M, k, K);
        }
      }
  };
  auto use = [&](const Frags& f) {
#pragma unroll
    for (int t = 0; t < MAX_MT; ++t)
      if (t < mt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w, f.x[t], acc[t], 0, 0, 0);
  };
  Frags f0, f1, f2;
  if (s0 < s1) fetch(s0, f0);
  if (s0 + 1 < s1) fetch(s0 + 1, f1);
  if (s0 + 2 < s1) fetch(s0 + 2, f2);
  for (int s = s0; s < s1; s += 3) {
    use(f0);
    if (s + 3 < s1) fetch(s + 3, f0);
    if (s + 1 < s1) { use(f1); if (s + 4 < s1) fetch(s + 4, f1); }
    if (s + 2 < s1) { use(f2); if (s + 5 < s1) fetch(s + 5, f2); }
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < MAX_MT; ++t)
      if (t < mt) *reinterpret_cast<f32x4_t*>(&red[wave - 1][t][lane][0]) = acc[t];
  }
  __syncthreads();
  if (wave == 0) {
    const int n = n0 + ((lane >> 4) << 2);             // D[i = n][j = m]: lane holds n..n+3 of row m
#pragma unroll
    for (int t = 0; t < MAX_MT; ++t) {
      const int m = t * 16 + (lane & 15);
      if (t < mt && m < M && n < N) {
        f32x4_t v = acc[t];
#pragma unroll
        for (int w = 0; w < SK_WAVES - 1; ++w) v += *reinterpret_cast<const f32x4_t*>(&red[w][t][lane][0]);
        if (a.flags & MMF_EPI_BIAS) v += *reinterpret_cast<const f32x4_t*>(P.bias + n);
        if (a.flags & MMF_EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.flags & MMF_EPI_DROPOUT) {
          const unsigned key = mmf_rng_key(*a.rng_state, a.site, (unsigned)pi), idx = (unsigned)m * (unsigned)N + (unsigned)n;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mmf_keep(key, idx + e, a.drop_thresh) ? v[e] * a.drop_scale : 0.f;
        }
        const u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        if (OUT_F32) {
          *reinterpret_cast<f32x4_t*>(static_cast<float*>(P.Y) + (size_t)m * P.ldy + n) = v;
          if (a.y2[pi]) *reinterpret_cast<u32x2_t*>(static_cast<unsigned short*>(a.y2[pi]) + (size_t)m * a.ldy2[pi] + n) = o;
        } else {
          *reinterpret_cast<u32x2_t*>(static_cast<unsigned short*>(P.Y) + (size_t)m * P.ldy + n) = o;
          if (a.y2[pi]) *reinterpret_cast<f32x4_t*>(static_cast<float*>(a.y2[pi]) + (size_t)m * a.ldy2[pi] + n) = v;
        }
      }
    }
  }
}

// ---- dgrad -----------------------------------------------------------------------------------------------
// X := dy [M][N_out] (bf16), W [N_out][K_in], Y := dx [M][K_in]; aux (bf16 [M][K_in], optional) is the saved
// activation whose sign gates the gradient (ReLU / dropout mask), alpha the 1/(1-p) of a dropout backward.
// CT = 16-column tiles of k_in per workgroup (strip width 16 CT).  4 is the streaming form; a launch whose 64-column strips would
// leave most CUs idle (K_in = 768: twelve workgroups, 21 us of latency chain for 1.2 MB of weights — 8 % of the hier-seq
// training step's kernel time in round 2) takes 2 or 1: four times the workgroups, each reading 32-byte row pieces that its
// neighbours' strips complete in L2.
template <bool OUT_F32, int CT>
__global__ __launch_bounds__(SK_THREADS)
void skinny_dgrad_kernel(const SkinnyArgs a) {
  constexpr int SB = (16 * CT + 8) * 2;                                  // padded slice row: 16 CT k_in + 8
  constexpr int RED_BYTES = (int)sizeof(float) * (SK_WAVES - 1) * CT * MAX_MT * 64 * 4;    // 112 KiB at CT = 4
  constexpr int SLICE_BYTES = SK_WAVES * 32 * SB;
  __shared__ __attribute__((aligned(16))) char smem[SLICE_BYTES > RED_BYTES ? SLICE_BYTES : RED_BYTES];
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.blk_start[pi + 1]) ++pi;
  const mmf_skinny_problem& P = a.p[pi];
  const int M = P.M, Nout = P.N, Kin = P.K;
  const int c0 = ((int)blockIdx.x - a.blk_start[pi]) * (16 * CT);       // first k_in column of this strip
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = (M + 15) >> 4;
  const unsigned short* __restrict__ dY = static_cast<const unsigned short*>(P.X);
  const unsigned short* __restrict__ W = static_cast<const unsigned short*>(P.W);
  char* slice = smem + wave * 32 * SB;
  const void* gate = a.gate[pi];
  unsigned short* dz = static_cast<unsigned short*>(a.dz[pi]);
  const bool gated = a.x_f32 || gate || dz || (a.flags & MMF_EPI_DROPOUT);
  const unsigned drop_key = (a.flags & MMF_EPI_DROPOUT) ? mmf_rng_key(*a.rng_state, a.site, (unsigned)pi) : 0u;

  const int steps = (Nout + 31) >> 5, per = (steps + SK_WAVES - 1) / SK_WAVES;
  const int s0 = wave * per, s1 = min(steps, s0 + per);
  f32x4_t acc[CT][MAX_MT];                                               // [k_in tile][m tile]
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < MAX_MT; ++t) acc[c][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  // three steps in flight per wave, as in the forward (a 3072-row W strip is 12 steps per wave: 30 -> 10 us for GraphFusion's dgrads)
  struct Block { u32x4_t stage[CT]; bf16x8_t y[MAX_MT]; };
  auto fetch = [&](int s, Block& bk) {                                   // 32 x 16 CT block of W (16-B chunks, coalesced) + dy
    const int r0 = s << 5;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int ch = lane + 64 * i, row = ch / (2 * CT), col = (ch % (2 * CT)) << 3;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (r0 + row < Nout && c0 + col < Kin) v = *reinterpret_cast<const u32x4_t*>(W + (size_t)(r0 + row) * P.ldw + c0 + col);
      bk.stage[i] = v;
    }
#pragma unroll
    for (int t = 0; t < MAX_MT; ++t)
      if (t < mt) {
        const int m = t * 16 + (lane & 15), n = r0 + (g << 3);
        if (!gated) {
          bk.y[t] = load_frag_rows(dY, P.ldx, m, M, n, Nout);
        } else {
          // dz = dy * (gate > 0) * gate_scale [* keep / (1 - p)], narrowed to bf16: the ReLU / dropout gradient and the casts
          // around it happen while the fragment is loaded; the first strip's workgroup also writes dz out (wgrad operand)
          float v[8], gt[8];
          load8_f32(P.X, a.x_f32 != 0, P.ldx, m, M, n, Nout, v);
          if (gate) {
            load8_f32(gate, a.gate_f32 != 0, a.ldgate[pi], m, M, n, Nout, gt);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gt[e] > 0.f ? v[e] * a.gate_scale : 0.f;
          }
          if (a.flags & MMF_EPI_DROPOUT) {
            const unsigned idx = (unsigned)m * (unsigned)Nout + (unsigned)n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = mmf_keep(drop_key, idx + e, a.drop_thresh) ? v[e] * a.drop_scale : 0.f;
          }
          bk.y[t] = pack8(v);
          if (dz && c0 == 0 && m < M && n < Nout)
            *reinterpret_cast<u32x4_t*>(dz + (size_t)m * a.lddz[pi] + n) = __builtin_bit_cast(u32x4_t, bk.y[t]);
        }
      }
  };
  auto use = [&](const Block& bk) {
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int ch = lane + 64 * i, row = ch / (2 * CT), col = (ch % (2 * CT)) << 3;
      *reinterpret_cast<u32x4_t*>(slice + row * SB + col * 2) = bk.stage[i];
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      // W^T fragment: lane (i = k_in column c*16 + (l & 15), k = n_out 8g + e) from rows 8g .. 8g+7 of the slice
      const char* ap = slice + (8 * g + q) * SB + (c * 16 + 4 * pp) * 2;
      const s16x4_t lo = lds_read_tr16(ap);
      const s16x4_t hi = lds_read_tr16(ap + 4 * SB);
      const s16x8_t wv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wv);
#pragma unroll
      for (int t = 0; t < MAX_MT; ++t)
        if (t < mt) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, bk.y[t], acc[c][t], 0, 0, 0);
    }
  };
  Block b0, b1, b2;
  if (s0 < s1) fetch(s0, b0);
  if (s0 + 1 < s1) fetch(s0 + 1, b1);
  if (s0 + 2 < s1) fetch(s0 + 2, b2);
  for (int s = s0; s < s1; s += 3) {
    use(b0);
    if (s + 3 < s1) fetch(s + 3, b0);
    if (s + 1 < s1) { use(b1); if (s + 4 < s1) fetch(s + 4, b1); }
    if (s + 2 < s1) { use(b2); if (s + 5 < s1) fetch(s + 5, b2); }
  }
  __syncthreads();                                                       // slices are dead: reuse LDS for the sums
  float* red = reinterpret_cast<float*>(smem);                           // [7 waves][CT c][MAX_MT][64 lanes][4]
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < MAX_MT; ++t)
        if (t < mt) *reinterpret_cast<f32x4_t*>(red + ((((wave - 1) * CT + c) * MAX_MT + t) * 64 + lane) * 4) = acc[c][t];
  }
  __syncthreads();
  if (wave == 0) {
    const unsigned short* __restrict__ aux = static_cast<const unsigned short*>(P.aux);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int k = c0 + c * 16 + (g << 2);                              // D[i = k_in][j = m]
#pragma unroll
      for (int t = 0; t < MAX_MT; ++t) {
        const int m = t * 16 + (lane & 15);
        if (t < mt && m < M && k < Kin) {
          f32x4_t v = acc[c][t];
#pragma unroll
          for (int w = 0; w < SK_WAVES - 1; ++w) v += *reinterpret_cast<const f32x4_t*>(red + (((w * CT + c) * MAX_MT + t) * 64 + lane) * 4);
          if (a.flags & MMF_EPI_MASK_AUX) {
            const u32x2_t x = *reinterpret_cast<const u32x2_t*>(aux + (size_t)m * P.ldaux + k);
            v[0] = bf16lo(x[0]) > 0.f ? v[0] : 0.f; v[1] = bf16hi(x[0]) > 0.f ? v[1] : 0.f;
            v[2] = bf16lo(x[1]) > 0.f ? v[2] : 0.f; v[3] = bf16hi(x[1]) > 0.f ? v[3] : 0.f;
          }
          v *= a.alpha;
          if (OUT_F32) {
            *reinterpret_cast<f32x4_t*>(static_cast<float*>(P.Y) + (size_t)m * P.ldy + k) = v;
          } else {
            const u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2_t*>(static_cast<unsigned short*>(P.Y) + (size_t)m * P.ldy + k) = o;
          }
        }
      }
    }
  }
}

int check(const char* who, const mmf_skinny_problem* p, int n, int flags, bool dgrad, bool x_f32) {
  if (!p || n <= 0 || n > MMF_SKINNY_MAX_PROBLEMS) MMF_FAIL(MMF_E_SHAPE, "%s: num_problems=%d out of range", who, n);
  for (int i = 0; i < n; ++i) {
    const mmf_skinny_problem& q = p[i];
    if (q.M <= 0 || q.M > 16 * MAX_MT || q.N <= 0 || q.K <= 0)
      MMF_FAIL(MMF_E_SHAPE, "%s[%d]: M=%d (1..%d) N=%d K=%d", who, i, q.M, 16 * MAX_MT, q.N, q.K);
    const int out_cols = dgrad ? q.K : q.N, red = dgrad ? q.N : q.K;
    if ((red & 7) || (q.ldx & (x_f32 ? 3 : 7)) || (q.ldw & 7) || (out_cols & 3) || (q.ldy & 3) || (dgrad && (q.K & 7)))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: reduction extent / leading dimensions must be multiples of 8, outputs of 4", who, i);
    if (!q.X || !q.W || !q.Y || !mmf_aligned16(q.X) || !mmf_aligned16(q.W) || !mmf_aligned16(q.Y))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: null or unaligned operand", who, i);
    if ((flags & MMF_EPI_BIAS) && (!q.bias || !mmf_aligned16(q.bias))) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: bias", who, i);
    if ((flags & MMF_EPI_MASK_AUX) && (!q.aux || (q.ldaux & 3))) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: aux", who, i);
  }
  return MMF_OK;
}

// fills the round-4 fields of the kernel arguments from the extended problem table (or clears them)
int fill_ex(const char* who, SkinnyArgs& a, const mmf_skinny_problem_ex* px, int n, const mmf_skinny_extra* ex, bool dgrad) {
  a.x_f32 = ex ? ex->x_f32 : 0;
  a.gate_f32 = ex ? ex->gate_f32 : 0;
  a.gate_scale = ex ? ex->gate_scale : 1.f;
  a.rng_state = ex ? reinterpret_cast<const unsigned long long*>(ex->rng_state) : nullptr;
  a.site = ex ? ex->site : 0u;
  a.drop_thresh = 0u; a.drop_scale = 1.f;
  if (a.flags & MMF_EPI_DROPOUT) {
    if (!ex || !ex->rng_state || !(ex->dropout_p >= 0.f) || ex->dropout_p >= 1.f)
      MMF_FAIL(MMF_E_SHAPE, "%s: MMF_EPI_DROPOUT needs rng_state and 0 <= p < 1", who);
    a.drop_thresh = mmf_drop_thresh(ex->dropout_p);
    a.drop_scale = 1.f / (1.f - (float)a.drop_thresh * (1.f / 4294967296.f));
  }
  for (int i = 0; i < n; ++i) {
    a.y2[i] = px ? px[i].Y2 : nullptr; a.ldy2[i] = px ? px[i].ldy2 : 0;
    a.gate[i] = px ? px[i].gate : nullptr; a.ldgate[i] = px ? px[i].ldgate : 0;
    a.dz[i] = px ? px[i].dz : nullptr; a.lddz[i] = px ? px[i].lddz : 0;
    if (!px) continue;
    const mmf_skinny_problem& q = px[i].p;
    if (a.x_f32 && (q.ldx & 3)) MMF_FAIL(MMF_E_ALIGN, "%s[%d]: f32 input rows must keep 16-byte alignment (ldx %% 4)", who, i);
    if (!dgrad && a.y2[i] && (!mmf_aligned16(a.y2[i]) || (a.ldy2[i] & 3) || a.ldy2[i] < q.N))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: second output", who, i);
    if (dgrad && a.gate[i] && (!mmf_aligned16(a.gate[i]) || (a.ldgate[i] & (a.gate_f32 ? 3 : 7)) || a.ldgate[i] < q.N))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: gate", who, i);
    if (a.dz[i] && (!mmf_aligned16(a.dz[i]) || (a.lddz[i] & 7) || a.lddz[i] < (dgrad ? q.N : q.K)))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: dz", who, i);
  }
  return MMF_OK;
}

int launch_fwd(SkinnyArgs& a, const mmf_skinny_problem* problems, int num_problems, int out_f32, void* stream) {
  int total = 0;
  for (int i = 0; i < num_problems; ++i) { a.blk_start[i] = total; total += (problems[i].N + 15) / 16; a.p[i] = problems[i]; }
  a.blk_start[num_problems] = total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (out_f32) hipLaunchKernelGGL(skinny_fwd_kernel<true>, dim3(total), dim3(SK_THREADS), 0, s, a);
  else         hipLaunchKernelGGL(skinny_fwd_kernel<false>, dim3(total), dim3(SK_THREADS), 0, s, a);
  MMF_CHECK_LAUNCH("mmf_skinny_linear_fwd");
  return MMF_OK;
}

int launch_dgrad(SkinnyArgs& a, const mmf_skinny_problem* problems, int num_problems, int out_f32, void* stream) {
  int wide = 0;
  for (int i = 0; i < num_problems; ++i) wide += (problems[i].K + 63) / 64;
  // strip width: 64 columns when that already gives the chip a workgroup per two CUs, else 32, else 16 (MMF_SKINNY_CT pins it)
  static const int pin = [] { const char* e = getenv("MMF_SKINNY_CT"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
  const int ct = pin ? pin : wide >= 128 ? 4 : wide >= 48 ? 2 : 1;
  int total = 0;
  for (int i = 0; i < num_problems; ++i) { a.blk_start[i] = total; total += (problems[i].K + 16 * ct - 1) / (16 * ct); a.p[i] = problems[i]; }
  a.blk_start[num_problems] = total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (out_f32) {
    if (ct == 4)      hipLaunchKernelGGL((skinny_dgrad_kernel<true, 4>), dim3(total), dim3(SK_THREADS), 0, s, a);
    else if (ct == 2) hipLaunchKernelGGL((skinny_dgrad_kernel<true, 2>), dim3(total), dim3(SK_THREADS), 0, s, a);
    else              hipLaunchKernelGGL((skinny_dgrad_kernel<true, 1>), dim3(total), dim3(SK_THREADS), 0, s, a);
  } else {
    if (ct == 4)      hipLaunchKernelGGL((skinny_dgrad_kernel<false, 4>), dim3(total), dim3(SK_THREADS), 0, s, a);
    else if (ct == 2) hipLaunchKernelGGL((skinny_dgrad_kernel<false, 2>), dim3(total), dim3(SK_THREADS), 0, s, a);
    else              hipLaunchKernelGGL((skinny_dgrad_kernel<false, 1>), dim3(total), dim3(SK_THREADS), 0, s, a);
  }
  MMF_CHECK_LAUNCH("mmf_skinny_linear_dgrad");
  return MMF_OK;
}

}  // namespace

extern "C" int mmf_skinny_linear_fwd(const mmf_skinny_problem* problems, int num_problems, int flags, int out_f32,
                                     void* stream) {
  if (int rc = check("mmf_skinny_linear_fwd", problems, num_problems, flags & ~MMF_EPI_DROPOUT, false, false)) return rc;
  if (flags & MMF_EPI_DROPOUT) MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_skinny_linear_fwd: dropout needs mmf_skinny_linear_fwd_ex");
  SkinnyArgs a; a.nprob = num_problems; a.flags = flags; a.alpha = 1.f;
  if (int rc = fill_ex("mmf_skinny_linear_fwd", a, nullptr, num_problems, nullptr, false)) return rc;
  return launch_fwd(a, problems, num_problems, out_f32, stream);
}

extern "C" int mmf_skinny_linear_dgrad(const mmf_skinny_problem* problems, int num_problems, int flags, float alpha,
                                       int out_f32, void* stream) {
  if (int rc = check("mmf_skinny_linear_dgrad", problems, num_problems, flags & ~MMF_EPI_DROPOUT, true, false)) return rc;
  if (flags & MMF_EPI_DROPOUT) MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_skinny_linear_dgrad: dropout needs mmf_skinny_linear_dgrad_ex");
  SkinnyArgs a; a.nprob = num_problems; a.flags = flags; a.alpha = alpha;
  if (int rc = fill_ex("mmf_skinny_linear_dgrad", a, nullptr, num_problems, nullptr, true)) return rc;
  return launch_dgrad(a, problems, num_problems, out_f32, stream);
}

extern "C" int mmf_skinny_linear_fwd_ex(const mmf_skinny_problem_ex* problems, int num_problems, int flags, int out_f32,
                                        const mmf_skinny_extra* extra, void* stream) {
  if (!problems || num_problems <= 0 || num_problems > MMF_SKINNY_MAX_PROBLEMS)
    MMF_FAIL(MMF_E_SHAPE, "mmf_skinny_linear_fwd_ex: num_problems=%d out of range", num_problems);
  mmf_skinny_problem plain[MMF_SKINNY_MAX_PROBLEMS];
  for (int i = 0; i < num_problems; ++i) plain[i] = problems[i].p;
  if (int rc = check("mmf_skinny_linear_fwd_ex", plain, num_problems, flags & ~MMF_EPI_DROPOUT, false, extra && extra->x_f32)) return rc;
  SkinnyArgs a; a.nprob = num_problems; a.flags = flags; a.alpha = 1.f;
  if (int rc = fill_ex("mmf_skinny_linear_fwd_ex", a, problems, num_problems, extra, false)) return rc;
  return launch_fwd(a, plain, num_problems, out_f32, stream);
}

extern "C" int mmf_skinny_linear_dgrad_ex(const mmf_skinny_problem_ex* problems, int num_problems, int flags, float alpha,
                                          int out_f32, const mmf_skinny_extra* extra, void* stream) {
  if (!problems || num_problems <= 0 || num_problems > MMF_SKINNY_MAX_PROBLEMS)
    MMF_FAIL(MMF_E_SHAPE, "mmf_skinny_linear_dgrad_ex: num_problems=%d out of range", num_problems);
  mmf_skinny_problem plain[MMF_SKINNY_MAX_PROBLEMS];
  for (int i = 0; i < num_problems; ++i) plain[i] = problems[i].p;
  if (int rc = check("mmf_skinny_linear_dgrad_ex", plain, num_problems, flags & ~MMF_EPI_DROPOUT, true, extra && extra->x_f32)) return rc;
  SkinnyArgs a; a.nprob = num_problems; a.flags = flags; a.alpha = alpha;
  if (int rc = fill_ex("mmf_skinny_linear_dgrad_ex", a, problems, num_problems, extra, true)) return rc;
  return launch_dgrad(a, plain, num_problems, out_f32, stream);
}
