// Internal helpers shared by the gfx950 kernels of libmmfusion.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include "../../include/mmfusion.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

#define MMF_WAVE 64

// ---- error reporting (host) ---------------------------------------------------------------
void mmf_set_error(const char* fmt, ...);
#define MMF_FAIL(code, ...) do { mmf_set_error(__VA_ARGS__); return (code); } while (0)
#define MMF_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
  if (e_ != hipSuccess) MMF_FAIL(MMF_E_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

static inline bool mmf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- bf16 <-> f32 (device) ------------------------------------------------------------------
// bf16 is carried as its 16-bit pattern.  f32 -> bf16 uses the plain cast so that hipcc emits
// v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN: MI355X_MICROARCH.md correctness table).
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  f32x2_t v = {lo, hi};
  bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf16lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// ---- wave reductions ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- tile id of workgroup `orig` in a grouped launch of `total` tiles -----------------------------------
// The dispatcher deals workgroups to the 8 XCDs round-robin (orig % 8) and in order, so XCD x runs ids x, x+8, ...
// Tiles are handed out in GRANULES of G consecutive tile ids, granule g to XCD g % 8: the tiles an XCD runs at
// one time are still neighbours (they share A/B panels in that XCD's L2), and every XCD gets the same mix of
// problems.  One contiguous range per XCD (G = 0, the first form) is only balanced when all tiles take equally
// long: with the problems ordered by K descending it gave XCD 0 nothing but the longest tiles and the launch
// ended when XCD 0 did.  The last (< 8 G) tiles, and launches smaller than that, keep one range per XCD.
__device__ __forceinline__ int mmf_xcd_tile(int orig, int total, int G) {
  const int xcd = orig & 7;
  const int full = G > 0 ? (total / (8 * G)) * (8 * G) : 0;
  if (orig < full) {
    const int j = orig >> 3;
    return ((j / G) * 8 + xcd) * G + (j % G);
  }
  const int rest = total - full, q = rest >> 3, r = rest & 7;
  return full + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ((orig - full) >> 3);
}
static inline int mmf_xcd_granule() {                       // host: MMF_GEMM_XCD_GRANULE (default 32, 0 = first form)
  static const int g = [] { const char* e = getenv("MMF_GEMM_XCD_GRANULE"); const int v = e ? atoi(e) : 32; return v > 0 ? v : 0; }();
  return g;
}

// ---- counter-based dropout RNG -----------------------------------------------------------------------
// keep(e) for element e of dropout call-site `site`, stream `sub` (problem / (b,h) index) under the
// DEVICE-resident 64-bit state `seed`: two rounds of the lowbias32 integer hash.  Stateless, so the
// backward pass regenerates the forward's mask instead of storing it, and because the state lives in
// device memory (advanced once per training step by the host side) a replayed hipGraph draws fresh
// masks every step.  keep probability = 1 - p with p quantised to 2^-32.
__device__ __forceinline__ unsigned mmf_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned mmf_rng_key(unsigned long long seed, unsigned site, unsigned sub) {
  return mmf_mix32((unsigned)seed ^ (site * 0x9E3779B9u)) ^ (unsigned)(seed >> 32) ^ (sub * 0x85EBCA6Bu);
}
__device__ __forceinline__ bool mmf_keep(unsigned key, unsigned idx, unsigned thresh) {
  return mmf_mix32(key + idx * 0x9E3779B9u) >= thresh;
}
static inline unsigned mmf_drop_thresh(float p) {          // host: p in [0, 1)
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}

// transposed LDS read (ds_read_b64_tr_b16): per 16-lane group a 4-row x 16-column block of
// 16-bit elements; lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives
// column i of the four rows (cdna_hip_programming.md T10).  EXEC must be all ones.
__device__ __forceinline__ s16x4_t lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(lds_addr));
}
