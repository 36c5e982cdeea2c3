// Parts shared by the one-wave-per-SIMD GEMM kernels (gemm6.hip: one tile per workgroup; gemm7.hip: persistent workgroups):
// the kernel-argument table, fragment reads and their LDS addressing, the LDS-DMA source map, tile order and the epilogue.
#pragma once
#include "mmf_internal.h"
#include "lds_image.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256;
constexpr int NTHREADS = 256;

struct GemmArgs {
  int nprob;
  int epi;
  int xcd_granule;                   // mmf_xcd_tile()
  float alpha;                       // multiplies the result after the mask step
  unsigned drop_thresh, site;        // MMF_EPI_DROPOUT
  const unsigned long long* rng_state;
  int tile_start[MMF_GEMM_MAX_PROBLEMS + 1];
  short orig[MMF_GEMM_MAX_PROBLEMS];   // gemm7: the caller's index of problem i (its problems are sorted by K; the dropout stream id is the caller's)
  mmf_gemm_problem p[MMF_GEMM_MAX_PROBLEMS];
};

typedef __attribute__((address_space(3))) void lds_void_t;

template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- fragment reads (inline asm; see the header comment) -----------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ u32x4_t lds_read_b128(unsigned addr) {
  u32x4_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ u32x2_t lds_read_tr(unsigned addr) {
  u32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// One operand's four fragments of a k-substep.  KC: lane (r = l & 31, h = l >> 5) reads chunk 2 G + h of tile row 32 D + r;
// KR: two transposed reads give it rows 16 G + 8 h + 0..7 of tile column 32 D + r (the standard MFMA k order).
template <bool KR> struct Frag4;
template <> struct Frag4<false> {
  u32x4_t v[4];
  // lane parts of the address for even / odd G
  static __device__ __forceinline__ void lane_parts(int W, int lane, unsigned& a0, unsigned& a1) {
    const int r = lane & 31, h = lane >> 5, s = (r >> 2) & 3;
    const unsigned base = (unsigned)((W / 32) * 512 * (r >> 3) + 64 * (r & 7));
    a0 = base + 16u * (unsigned)(h ^ s);
    a1 = base + 16u * (unsigned)((2 + h) ^ s);
  }
  template <int W, int G, int D0>
  __device__ __forceinline__ void issue(unsigned t0, unsigned t1) {          // t0 / t1: tile base + lane part for even / odd G
    const unsigned t = (G & 1) ? t1 : t0;
    constexpr int S = (W / 32) * 2048;                                       // 32 tile rows
    v[0] = lds_read_b128<S * (D0 + 0) + 512 * (G >> 1)>(t);
    v[1] = lds_read_b128<S * (D0 + 1) + 512 * (G >> 1)>(t);
    v[2] = lds_read_b128<S * (D0 + 2) + 512 * (G >> 1)>(t);
    v[3] = lds_read_b128<S * (D0 + 3) + 512 * (G >> 1)>(t);
  }
  template <int W, int G, int D0, int I>
  __device__ __forceinline__ void issue1(unsigned t0, unsigned t1) {
    v[I] = lds_read_b128<(W / 32) * 2048 * (D0 + I) + 512 * (G >> 1)>((G & 1) ? t1 : t0);
  }
  __device__ __forceinline__ bf16x8_t get(int i) const { return __builtin_bit_cast(bf16x8_t, v[i]); }
};
template <> struct Frag4<true> {
  u32x2_t lo[4], hi[4];
  static __device__ __forceinline__ void lane_parts(int, int lane, unsigned& a0, unsigned& a1) {
    const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const unsigned c = (unsigned)(2 * g + (p >> 1));
    a0 = (unsigned)(4096 * h + 64 * q) + 16u * (c ^ (unsigned)(2 * h)) + 8u * (unsigned)(p & 1);
    a1 = (unsigned)(4096 * h + 64 * (4 + q)) + 16u * (c ^ (unsigned)(2 * h + 1)) + 8u * (unsigned)(p & 1);
  }
  template <int W, int G, int D0>
  __device__ __forceinline__ void issue(unsigned t0, unsigned t1) {          // t0 / t1: tile base + lane part of the lo / hi read
    lo[0] = lds_read_tr<8192 * G + 512 * (D0 + 0)>(t0); hi[0] = lds_read_tr<8192 * G + 512 * (D0 + 0)>(t1);
    lo[1] = lds_read_tr<8192 * G + 512 * (D0 + 1)>(t0); hi[1] = lds_read_tr<8192 * G + 512 * (D0 + 1)>(t1);
    lo[2] = lds_read_tr<8192 * G + 512 * (D0 + 2)>(t0); hi[2] = lds_read_tr<8192 * G + 512 * (D0 + 2)>(t1);
    lo[3] = lds_read_tr<8192 * G + 512 * (D0 + 3)>(t0); hi[3] = lds_read_tr<8192 * G + 512 * (D0 + 3)>(t1);
  }
  template <int W, int G, int D0, int I>
  __device__ __forceinline__ void issue1(unsigned t0, unsigned t1) {
    lo[I] = lds_read_tr<8192 * G + 512 * (D0 + I)>(t0); hi[I] = lds_read_tr<8192 * G + 512 * (D0 + I)>(t1);
  }
  __device__ __forceinline__ bf16x8_t get(int i) const {
    const u32x4_t w = {lo[i][0], lo[i][1], hi[i][0], hi[i][1]};
    return __builtin_bit_cast(bf16x8_t, w);
  }
};
// lgkmcnt(0) tied to every destination register of the two operands' fragments: no use can be scheduled above it
__device__ __forceinline__ void frag_wait(Frag4<false>& a, Frag4<false>& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]));
}
__device__ __forceinline__ void frag_wait(Frag4<false>& a, Frag4<true>& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b.lo[0]), "+v"(b.lo[1]), "+v"(b.lo[2]), "+v"(b.lo[3]),
               "+v"(b.hi[0]), "+v"(b.hi[1]), "+v"(b.hi[2]), "+v"(b.hi[3]));
}
__device__ __forceinline__ void frag_wait(Frag4<true>& a, Frag4<true>& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo[0]), "+v"(a.lo[1]), "+v"(a.lo[2]), "+v"(a.lo[3]), "+v"(a.hi[0]), "+v"(a.hi[1]), "+v"(a.hi[2]), "+v"(a.hi[3]),
               "+v"(b.lo[0]), "+v"(b.lo[1]), "+v"(b.lo[2]), "+v"(b.lo[3]), "+v"(b.hi[0]), "+v"(b.hi[1]), "+v"(b.hi[2]), "+v"(b.hi[3]));
}

// per-lane source offset (bytes, relative to the operand tile's first element at k = 0) of 1-KiB piece p: the inverse of the image
template <bool KR, int BK>
__device__ __forceinline__ unsigned piece_voff(int p, int ld, int lane) {
  constexpr int NC = (KR ? 256 : BK) / 32;
  const int st = 2 * p + (lane >> 5), rg = st / NC, cc = st % NC;
  const int w = lane & 31, row = 8 * rg + (w >> 2), ch = 4 * cc + ((w & 3) ^ ((row >> 2) & 3));
  return (unsigned)(row * ld * 2 + ch * 16);
}

// tile t of problem P -> (m0, n0): super-rows of 8 m-tiles, n fastest across a super-row (gemm4.hip)
__device__ __forceinline__ void tile_origin(const mmf_gemm_problem& P, const int t, int& m0, int& n0) {
  const int tiles_m = (P.M + BM - 1) / BM, tiles_n = (P.N + BN - 1) / BN;
  // super-rows of 8 m-tiles.  (Round 4 tried the height that makes an XCD's 32 concurrent tiles touch the fewest operand panels — 32 / tiles_n
  // for narrow outputs, 11 x 3 instead of 8 x 3 + 8 x 1 at N = 768: the step got SLOWER, 2.027 -> 2.052 ms same box.)  MMF_GEMM_GROUPM
  // (build-time) pins another height for A/Bs.
#ifdef MMF_GEMM_GROUPM
  constexpr int GROUP_M = MMF_GEMM_GROUPM;
#else
  constexpr int GROUP_M = 8;
#endif
  const int grp = t / (GROUP_M * tiles_n), rem = t % (GROUP_M * tiles_n);
  const int gm = min(GROUP_M, tiles_m - grp * GROUP_M);
  m0 = (grp * GROUP_M + rem % gm) * BM;
  n0 = (rem / gm) * BN;
}

// epilogue: lane owns C[m][n .. n+3] for m = m_base + 32 tm + (l & 31), n = n_base + 32 tn + 8 g + 4 (l >> 5).
// A wave that owns its SIMD has nobody to cover a load's round trip, and the compiler may not move a load above an earlier store
// to C: the first form of this epilogue (gemm4's: bias / aux / old-C loads next to each use) spent ~35 us per tile in 64 exposed
// round trips.  Everything it reads is therefore requested up front: the bias once, and per 32-row block tm the aux row pieces /
// old C values of block tm + 1 before block tm is finished and stored.
// CT: the epilogue flags as a compile-time mask (alpha = 1, no dropout), or -1: every flag tested at run time.  One wave per SIMD
// walks the 64 register groups alone: with the flags tested per group the epilogue of a 256 x 256 tile took 10.7 us (ablation
// -DMMF_G6_DBG=64), ~4 of which are the CU's store path; the step's flag sets are instantiated (tile_epilogue below).
template <bool OUT_F32, int CT>
__device__ __forceinline__ void tile_epilogue_mode(const GemmArgs& args, const mmf_gemm_problem& P, const int pi, const int mb, const int nb,
                                                   f32x16_t (&acc)[4][4], const int lane, const int epi_all) {
  constexpr int MODE = CT < 0 ? 3 : 0;                     // 3: run-time flags
  const int M = P.M, N = P.N;
  const int epi = CT < 0 ? epi_all : (CT | (epi_all & MMF_EPI_ACCUM));
  const unsigned short* __restrict__ aux = static_cast<const unsigned short*>(P.aux);
  const bool do_drop = MODE == 3 && (epi & MMF_EPI_DROPOUT);
  const unsigned drop_key = do_drop ? mmf_rng_key(*args.rng_state, args.site, (unsigned)pi) : 0u;
  const float drop_scale = do_drop ? 1.f / (1.f - (float)args.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  const float alpha = MODE == 3 ? args.alpha : 1.f;
  const bool use_aux = epi & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX);
  const bool use_old = OUT_F32 && (epi & MMF_EPI_ACCUM);
  const int h = lane >> 5;
  auto ncol = [&](int tn, int g) { return nb + 32 * tn + 8 * g + 4 * h; };

  f32x4_t bv[4][4];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bv[tn][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if ((epi & MMF_EPI_BIAS) && ncol(tn, g) < N) bv[tn][g] = *reinterpret_cast<const f32x4_t*>(P.bias + ncol(tn, g));
    }
  auto finish = [&](f32x4_t v, const f32x4_t& b, const u32x2_t& a, int m, int n) -> f32x4_t {  // bias -> relu -> dropout -> mask -> alpha -> residual
    if constexpr (CT >= 0) {                                // compile-time flag set: bias -> relu -> mask -> residual
      if constexpr (CT & MMF_EPI_BIAS) v += b;
      if constexpr (CT & MMF_EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (CT & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) {
        const float a0 = bf16lo(a[0]), a1 = bf16hi(a[0]), a2 = bf16lo(a[1]), a3 = bf16hi(a[1]);
        if constexpr (CT & MMF_EPI_MASK_AUX) {
          v[0] = a0 > 0.f ? v[0] : 0.f; v[1] = a1 > 0.f ? v[1] : 0.f;
          v[2] = a2 > 0.f ? v[2] : 0.f; v[3] = a3 > 0.f ? v[3] : 0.f;
        } else {
          v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
        }
      }
      return v;
    }
    v += b;
    if (epi & MMF_EPI_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (do_drop) {
      const unsigned idx = (unsigned)m * (unsigned)N + (unsigned)n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mmf_keep(drop_key, idx + e, args.drop_thresh) ? v[e] * drop_scale : 0.f;
    }
    const float a0 = bf16lo(a[0]), a1 = bf16hi(a[0]), a2 = bf16lo(a[1]), a3 = bf16hi(a[1]);
    if (epi & MMF_EPI_MASK_AUX) {
      v[0] = a0 > 0.f ? v[0] : 0.f; v[1] = a1 > 0.f ? v[1] : 0.f;
      v[2] = a2 > 0.f ? v[2] : 0.f; v[3] = a3 > 0.f ? v[3] : 0.f;
    }
    v *= alpha;
    if (epi & MMF_EPI_ADD_AUX) { v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3; }
    return v;
  };
  const bool wide = (N & 7) == 0 && (P.ldc & 7) == 0;
  if constexpr (OUT_F32) {
    // f32 output (wgrad; the f32 heads), one 32 x 32 tile at a time: the old values of tile t + 1 are requested before tile t is
    // stored.  The aux forms are not offered here: mmf_gemm6_launch refuses them with f32 output.
    f32x4_t oldv[2][4];
    auto preload_tile = [&](int t, int slot) {
      const int tm = t >> 2, tn = t & 3, m = mb + 32 * tm + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        oldv[slot][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (m < M && ncol(tn, g) < N)
          oldv[slot][g] = *reinterpret_cast<const f32x4_t*>(static_cast<const float*>(P.C) + (size_t)m * P.ldc + ncol(tn, g));
      }
    };
    if (use_old) preload_tile(0, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int tm = t >> 2, tn = t & 3, m = mb + 32 * tm + (lane & 31);
      f32x4_t v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[g] = f32x4_t{acc[tn][tm][4 * g], acc[tn][tm][4 * g + 1], acc[tn][tm][4 * g + 2], acc[tn][tm][4 * g + 3]};
        v[g] = finish(v[g], bv[tn][g], u32x2_t{0u, 0u}, m, ncol(tn, g));
        if (use_old) v[g] += oldv[t & 1][g];
      }
      if (use_old && t + 1 < 16) preload_tile(t + 1, (t + 1) & 1);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (m < M && ncol(tn, g) < N)
          *reinterpret_cast<f32x4_t*>(static_cast<float*>(P.C) + (size_t)m * P.ldc + ncol(tn, g)) = v[g];
    }
  } else {
    // bf16 output, one 32 x 32 tile at a time, tile t + 1's aux pieces requested before tile t is stored
    u32x2_t axv[2][4];
    auto preload_tile = [&](int t, int slot) {
      const int tm = t >> 2, tn = t & 3, m = mb + 32 * tm + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        axv[slot][g] = u32x2_t{0u, 0u};
        if (m < M && ncol(tn, g) < N) axv[slot][g] = *reinterpret_cast<const u32x2_t*>(aux + (size_t)m * P.ldaux + ncol(tn, g));
      }
    };
    if (use_aux) preload_tile(0, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int tm = t >> 2, tn = t & 3, m = mb + 32 * tm + (lane & 31);
      u32x2_t o[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t v = {acc[tn][tm][4 * g], acc[tn][tm][4 * g + 1], acc[tn][tm][4 * g + 2], acc[tn][tm][4 * g + 3]};
        v = finish(v, bv[tn][g], use_aux ? axv[t & 1][g] : u32x2_t{0u, 0u}, m, ncol(tn, g));
        o[g] = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      if (use_aux && t + 1 < 16) preload_tile(t + 1, (t + 1) & 1);
      if (wide) {
        // 16-byte stores: the 8-byte pieces of register groups g and g + 1 are exchanged between the half-waves
        // (v_permlane32_swap), so lanes 0-31 hold 8 consecutive columns of group 2 gp and lanes 32-63 of group 2 gp + 1
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const auto r0 = __builtin_amdgcn_permlane32_swap(o[2 * gp][0], o[2 * gp + 1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(o[2 * gp][1], o[2 * gp + 1][1], false, false);
          const u32x4_t w = {r0[0], r1[0], r0[1], r1[1]};
          const int n = nb + 32 * tn + 16 * gp + 8 * h;
          if (m < M && n < N)
            *reinterpret_cast<u32x4_t*>(static_cast<unsigned short*>(P.C) + (size_t)m * P.ldc + n) = w;
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (m < M && ncol(tn, g) < N)
            *reinterpret_cast<u32x2_t*>(static_cast<unsigned short*>(P.C) + (size_t)m * P.ldc + ncol(tn, g)) = o[g];
      }
    }
  }
}

// epi_all: the launch's epilogue flags (gemm7.hip passes them without MMF_EPI_BIAS: its accumulators start from the bias)
template <bool OUT_F32>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& args, const mmf_gemm_problem& P, const int pi, const int mb, const int nb,
                                              f32x16_t (&acc)[4][4], const int lane, const int epi_all) {
  const int e = epi_all & ~(MMF_EPI_ACCUM | MMF_EPI_COLSUM_A);
  const bool unit = args.alpha == 1.f;
  // the flag sets of the fusion step: dgrads / wgrad (none), in-projections (bias), FFN1 (bias + ReLU), out-projection and FFN2
  // (bias + residual), dH (ReLU mask), dX (residual gradient)
  if (e == 0 && unit)                                              tile_epilogue_mode<OUT_F32, 0>(args, P, pi, mb, nb, acc, lane, epi_all);
  else if (e == MMF_EPI_BIAS && unit)                              tile_epilogue_mode<OUT_F32, MMF_EPI_BIAS>(args, P, pi, mb, nb, acc, lane, epi_all);
  else if (e == (MMF_EPI_BIAS | MMF_EPI_RELU) && unit)             tile_epilogue_mode<OUT_F32, MMF_EPI_BIAS | MMF_EPI_RELU>(args, P, pi, mb, nb, acc, lane, epi_all);
  else if (!OUT_F32 && e == (MMF_EPI_BIAS | MMF_EPI_ADD_AUX) && unit) tile_epilogue_mode<OUT_F32, MMF_EPI_BIAS | MMF_EPI_ADD_AUX>(args, P, pi, mb, nb, acc, lane, epi_all);
  else if (!OUT_F32 && e == MMF_EPI_MASK_AUX && unit)              tile_epilogue_mode<OUT_F32, MMF_EPI_MASK_AUX>(args, P, pi, mb, nb, acc, lane, epi_all);
  else if (!OUT_F32 && e == MMF_EPI_ADD_AUX && unit)               tile_epilogue_mode<OUT_F32, MMF_EPI_ADD_AUX>(args, P, pi, mb, nb, acc, lane, epi_all);
  else                                                             tile_epilogue_mode<OUT_F32, -1>(args, P, pi, mb, nb, acc, lane, epi_all);
}
}  // namespace
