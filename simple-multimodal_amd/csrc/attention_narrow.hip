// Attention problems with one NARROW side (<= 32 rows: MulT's 30-frame video stream against 512 text / 400 audio positions), round 4.
//
// In the wide kernels (attention2.hip) such a problem is one active wave per (b, h) walking the long side tile by tile: a
// chain of DMA -> barrier -> 2 blocks round trips, 2.4 us per 64-row tile, 20 us for 512 positions -- for 0.75 GFLOP.  In the step's
// grouped launches the four x30 problems cost as much as the wide problem beside them (profiles/r03_attention_generations.txt:
// narrow4 20.6 us forward; backward cross group 105.7 us against 73.6 for the 512 x 512 problem alone).  Here the long side is
// split over the four waves of the workgroup instead (wave w takes 32-row blocks w, w + 4, ...), every wave streams its
// blocks on its own -- rows straight from global memory into registers (the next block's loads in flight under the current
// block's MFMAs), through a wave-private LDS slice for the fragment layouts, no LDS-DMA, no barrier inside the sweep -- and the four
// partial results meet once, through LDS, at the end:
//   attn_fwd_narrowq_kernel      Tq <= 32: wave-partial (m, l, O^T) merged flash-decoding style
//   attn_bwd_dq_narrowq_kernel   Tq <= 32: dQ^T partial sums added (and delta = rowsum(dO o O) written for the dK/dV pass)
//   attn_bwd_dkv_narrowk_kernel  Tk <= 32: dK^T, dV^T partial sums added
// Arithmetic, dropout stream indices and output layouts are those of the wide kernels (reference:
// F.multi_head_attention_forward as called at models/fusion_layers.py:161-163,204); the mirrored cases (a narrow side that is
// the partitioned one: 2048 one-block waves) stay in the wide kernels, they are HBM-bound prologue + epilogue.
#include "attn2_common.h"

namespace {

struct NarrowArgs {
  int nprob;
  float scale;
  unsigned drop_thresh, site;
  float inv_keep;
  const unsigned long long* rng_state;
  int blk_start[MMF_ATTN_MAX_PROBLEMS + 1];   // one workgroup per (b, h)
  short orig[MMF_ATTN_MAX_PROBLEMS];          // caller's problem index (dropout stream id)
  mmf_attn_problem p[MMF_ATTN_MAX_PROBLEMS];
};

// per wave: two 32-row slices of the image + 1 KiB of row statistics; the wave's partial result overlays the slices at the end
template <int DH> constexpr int narrow_region() { return 2 * img_slice_bytes<DH>() + 1024; }

// transposed fragment (attn_helpers.h: tr_lane_lo / tr_lane_hi) of a 32-row slice, compiler-visible reads (no LDS-DMA in these
// kernels, so hipcc's own lgkmcnt bookkeeping is exact)
template <int DH, int G, int D>
__device__ __forceinline__ bf16x8_t tr_frag(const char* slice, unsigned lo, unsigned hi) {
  const s16x4_t x = lds_read_tr16(slice + lo + img_tr_imm<DH, G, D>());
  const s16x4_t y = lds_read_tr16(slice + hi + img_tr_imm<DH, G, D>() + (DH / 32) * 512);
  return join(x, y);
}

template <int NR>
__device__ __forceinline__ void put_partial(float* red, const f32x16_t (&x)[NR], int lane) {
#pragma unroll
  for (int t = 0; t < NR; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = x[t][r];
}
template <int NR>
__device__ __forceinline__ void add_partial(const float* red, f32x16_t (&x)[NR], int lane, float f) {
#pragma unroll
  for (int t = 0; t < NR; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = __builtin_fmaf(red[(t * 16 + r) * 64 + lane], f, x[t][r]);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void* base, size_t off, int T, int ld) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(static_cast<const unsigned short*>(base) + off), 0, T * ld * 2, 0x00020000);
}

#define NARROW_HEAD                                                                                      \
  constexpr int KS = DH / 16, DT = DH / 32, SLICE = img_slice_bytes<DH>(), REGION = narrow_region<DH>(); \
  __shared__ __attribute__((aligned(1024))) char smem[4 * REGION];                                       \
  const int bid = blockIdx.x;                                                                            \
  int pi = 0;                                                                                            \
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;                                           \
  const mmf_attn_problem& P = a.p[pi];                                                                   \
  const int bh = bid - a.blk_start[pi], pidx = a.orig[pi];                                               \
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;                                        \
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                             \
  const int Tq = P.Tq, Tk = P.Tk, H = P.H, b = bh / H, h = bh % H;                                       \
  char* s0 = smem + wave * REGION;                                                                       \
  char* s1 = s0 + SLICE;                                                                                 \
  const unsigned tlo = tr_lane_lo(lane), thi = tr_lane_hi(lane);                                         \
  const float c = a.scale * LOG2E;                                                                       \
  const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;     \
  (void)KS; (void)DT; (void)s1; (void)tlo; (void)thi; (void)c; (void)dkey; (void)half

// ---- forward, Tq <= 32 -----------------------------------------------------------------------------------------------
template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 2)
void attn_fwd_narrowq_kernel(const NarrowArgs a) {
  NARROW_HEAD;
  const size_t koff = (size_t)b * Tk * P.ldk + h * DH, voff = (size_t)b * Tk * P.ldv + h * DH;
  const __amdgpu_buffer_rsrc_t rsQ = rows_rsrc(P.Q, (size_t)b * Tq * P.ldq + h * DH, Tq, P.ldq);
  const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(P.K, koff, Tk, P.ldk), rsV = rows_rsrc(P.V, voff, Tk, P.ldv);
  const int nkb = (Tk + 31) >> 5;
  RowLoad<DH> rq, rk, rv;
  rq.issue(rsQ, P.ldq, 0, lane);
  int kb = wave;
  if (kb < nkb) { rk.issue(rsK, P.ldk, 32 * kb, lane); rv.issue(rsV, P.ldv, 32 * kb, lane); }
  bf16x8_t qf[KS];
  rq.commit(qf, lane, s0);

  f32x16_t o[DT];
  float m = NEG_BIG, l = 0.f;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  const unsigned qidx = (unsigned)(lane & 31) * (unsigned)Tk;

  for (; kb < nkb; kb += 4) {
    bf16x8_t kf[KS];
    rk.commit(kf, lane, s0);
    rv.store(lane, s1);
    if (kb + 4 < nkb) { rk.issue(rsK, P.ldk, 32 * (kb + 4), lane); rv.issue(rsV, P.ldv, 32 * (kb + 4), lane); }
    const int k0 = 32 * kb;
    f32x16_t s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);   // S^T = K.Q^T
    if (k0 + 32 > Tk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        s[r] = key < Tk ? s[r] : NEG_BIG;
      }
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = half_max(mx) * c;
    if (!__all(mx <= m + DEFER)) {
      const float mnew = fmaxf(m, mx);
      const float alpha = fast_exp2(m - mnew);
      m = mnew;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float nm = -m;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(__builtin_fmaf(s[r], c, nm));
      s[r] = p;
      rs += p;
    }
    l += rs;
    if (DROP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned key = (unsigned)(k0 + (r & 3) + 8 * (r >> 2) + 4 * half);
        s[r] = mmf_keep(dkey, qidx + key, a.drop_thresh) ? s[r] * a.inv_keep : 0.f;
      }
    }
    const bf16x8_t p0 = acc_frag(s, 0), p1 = acc_frag(s, 1);
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 0>(s1, tlo, thi), p0, o[0], 0, 0, 0);      // O^T += V^T.P^T
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 1>(s1, tlo, thi), p0, o[1], 0, 0, 0);
    if constexpr (DT == 3) o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 2>(s1, tlo, thi), p0, o[2], 0, 0, 0);
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 0>(s1, tlo, thi), p1, o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 1>(s1, tlo, thi), p1, o[1], 0, 0, 0);
    if constexpr (DT == 3) o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 2>(s1, tlo, thi), p1, o[2], 0, 0, 0);
  }

  // merge the four waves' (m, l, O^T): waves 1..3 park theirs over their slices, wave 0 rescales and adds
  float* red = reinterpret_cast<float*>(s0);
  float* ml = reinterpret_cast<float*>(s0 + 2 * SLICE);
  if (wave != 0) {
    put_partial<DT>(red, o, lane);
    ml[lane] = m;
    ml[64 + lane] = l;
  }
  __syncthreads();
  if (wave != 0) return;
  float ms = m;
#pragma unroll
  for (int w = 1; w < 4; ++w) ms = fmaxf(ms, reinterpret_cast<const float*>(smem + w * REGION + 2 * SLICE)[lane]);
  const float f0 = fast_exp2(m - ms);
  l *= f0;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] *= f0;
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float* mlw = reinterpret_cast<const float*>(smem + w * REGION + 2 * SLICE);
    const float fw = fast_exp2(mlw[lane] - ms);
    l = __builtin_fmaf(mlw[64 + lane], fw, l);
    add_partial<DT>(reinterpret_cast<const float*>(smem + w * REGION), o, lane, fw);
  }
  const float lt = half_sum(l);
  unsigned short* Og = static_cast<unsigned short*>(P.O) + (size_t)b * Tq * P.ldo + h * DH;
  store_rows_lds<DH>(o, 1.f / lt, Og, P.ldo, 0, Tq, lane, s0);
  const int qrow = lane & 31;
  if (half == 0 && qrow < Tq) P.LSE[(size_t)bh * Tq + qrow] = ms * LN2 + __logf(lt);
}

// ---- dQ (+ delta), Tq <= 32 ------------------------------------------------------------------------------------------
template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 1)      // head dimension 96 needs ~265 registers; these launches are at most one workgroup per CU anyway
void attn_bwd_dq_narrowq_kernel(const NarrowArgs a) {
  NARROW_HEAD;
  const size_t qoff = (size_t)b * Tq * P.ldq + h * DH, ooff = (size_t)b * Tq * P.ldo + h * DH;
  const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(P.K, (size_t)b * Tk * P.ldk + h * DH, Tk, P.ldk);
  const __amdgpu_buffer_rsrc_t rsV = rows_rsrc(P.V, (size_t)b * Tk * P.ldv + h * DH, Tk, P.ldv);
  const int nkb = (Tk + 31) >> 5;
  RowLoad<DH> rq, rdo, ro, rk, rv;
  rq.issue(rows_rsrc(P.Q, qoff, Tq, P.ldq), P.ldq, 0, lane);
  rdo.issue(rows_rsrc(P.dO, ooff, Tq, P.ldo), P.ldo, 0, lane);
  ro.issue(rows_rsrc(P.O, ooff, Tq, P.ldo), P.ldo, 0, lane);
  int kb = wave;
  if (kb < nkb) { rk.issue(rsK, P.ldk, 32 * kb, lane); rv.issue(rsV, P.ldv, 32 * kb, lane); }
  bf16x8_t qf[KS], dof[KS];
  float delta = 0.f;
  {
    bf16x8_t of[KS];
    rq.commit(qf, lane, s0);
    rdo.commit(dof, lane, s0);
    ro.commit(of, lane, s0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4_t x = __builtin_bit_cast(u32x4_t, of[ks]), y = __builtin_bit_cast(u32x4_t, dof[ks]);
#pragma unroll
      for (int e = 0; e < 4; ++e) delta += bf16lo(x[e]) * bf16lo(y[e]) + bf16hi(x[e]) * bf16hi(y[e]);
    }
    delta = half_sum(delta);
  }
  const int qrow = lane & 31;
  if (wave == 0 && half == 0 && qrow < Tq) P.delta[(size_t)bh * Tq + qrow] = delta;
  const float lse2 = (qrow < Tq ? P.LSE[(size_t)bh * Tq + qrow] : 0.f) * LOG2E;
  const unsigned qidx = (unsigned)qrow * (unsigned)Tk;

  f32x16_t dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  for (; kb < nkb; kb += 4) {
    bf16x8_t kf[KS], vf[KS];
    rk.commit(kf, lane, s0);
    rv.commit(vf, lane, s1);
    if (kb + 4 < nkb) { rk.issue(rsK, P.ldk, 32 * (kb + 4), lane); rv.issue(rsV, P.ldv, 32 * (kb + 4), lane); }
    const int k0 = 32 * kb;
    f32x16_t s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], dof[ks], dp, 0, 0, 0);
    }
    const bool ragged = k0 + 32 > Tk;
    f32x16_t ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = fast_exp2(__builtin_fmaf(s[r], c, -lse2));
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (ragged) p = key < Tk ? p : 0.f;
      float dpv = dp[r];
      if (DROP) dpv = mmf_keep(dkey, qidx + (unsigned)key, a.drop_thresh) ? dpv * a.inv_keep : 0.f;
      ds[r] = p * (dpv - delta);
    }
    const bf16x8_t d0 = acc_frag(ds, 0), d1 = acc_frag(ds, 1);
    dq[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 0>(s0, tlo, thi), d0, dq[0], 0, 0, 0);   // dQ^T += K^T.dS^T
    dq[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 1>(s0, tlo, thi), d0, dq[1], 0, 0, 0);
    if constexpr (DT == 3) dq[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 2>(s0, tlo, thi), d0, dq[2], 0, 0, 0);
    dq[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 0>(s0, tlo, thi), d1, dq[0], 0, 0, 0);
    dq[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 1>(s0, tlo, thi), d1, dq[1], 0, 0, 0);
    if constexpr (DT == 3) dq[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 2>(s0, tlo, thi), d1, dq[2], 0, 0, 0);
  }

  if (wave != 0) put_partial<DT>(reinterpret_cast<float*>(s0), dq, lane);
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w) add_partial<DT>(reinterpret_cast<const float*>(smem + w * REGION), dq, lane, 1.f);
  store_rows_lds<DH>(dq, a.scale, static_cast<unsigned short*>(P.dQ) + qoff, P.ldq, 0, Tq, lane, s0);
}

// ---- dK / dV, Tk <= 32 -----------------------------------------------------------------------------------------------
template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 1)      // ~340 registers at head dimension 96
void attn_bwd_dkv_narrowk_kernel(const NarrowArgs a) {
  NARROW_HEAD;
  const size_t koff = (size_t)b * Tk * P.ldk + h * DH, voff = (size_t)b * Tk * P.ldv + h * DH;
  const __amdgpu_buffer_rsrc_t rsQ = rows_rsrc(P.Q, (size_t)b * Tq * P.ldq + h * DH, Tq, P.ldq);
  const __amdgpu_buffer_rsrc_t rsdO = rows_rsrc(P.dO, (size_t)b * Tq * P.ldo + h * DH, Tq, P.ldo);
  const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(P.LSE + (size_t)bh * Tq, 0, Tq * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(P.delta + (size_t)bh * Tq, 0, Tq * 4, 0x00020000);
  const int nqb = (Tq + 31) >> 5;
  RowLoad<DH> rk, rv, rq, rdo;
  rk.issue(rows_rsrc(P.K, koff, Tk, P.ldk), P.ldk, 0, lane);
  rv.issue(rows_rsrc(P.V, voff, Tk, P.ldv), P.ldv, 0, lane);
  int qb = wave;
  float st_l = 0.f, st_d = 0.f;
  auto fetch = [&](int blk) {
    rq.issue(rsQ, P.ldq, 32 * blk, lane);
    rdo.issue(rsdO, P.ldo, 32 * blk, lane);
    const unsigned off = (unsigned)(32 * blk + (lane & 31)) * 4u;       // rows past Tq read as 0 (harmless: their Q, dO rows are zeros)
    st_l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, off, 0, 0));
    st_d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, off, 0, 0));
  };
  if (qb < nqb) fetch(qb);
  bf16x8_t kf[KS], vf[KS];
  rk.commit(kf, lane, s0);
  rv.commit(vf, lane, s0);

  f32x16_t dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const unsigned kcol = (unsigned)(lane & 31);
  const float nls = -1.f / a.scale;
  const bf16x8_t one3 = ones3_frag(half == 0);

  for (; qb < nqb; qb += 4) {
    bf16x8_t qfr[KS], dofr[KS];
    rq.commit(qfr, lane, s0);
    rdo.commit(dofr, lane, s1);
    const float cl = st_l * nls, cd = -st_d;
    if (qb + 4 < nqb) fetch(qb + 4);
    const int q0 = 32 * qb;
    f32x16_t s, dp, dm, z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(split3_frag(cl, half == 0), one3, z, 0, 0, 0);    // S' starts at -LSE / scale
    dm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(split3_frag(cd, half == 0), one3, z, 0, 0, 0);   // -delta[q] in every key column
    if constexpr (DROP) dp = z; else dp = dm;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr[ks], vf[ks], dp, 0, 0, 0);
    }
    f32x16_t ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(s[r] * c);
      if constexpr (DROP) {
        const unsigned q = (unsigned)(q0 + (r & 3) + 8 * (r >> 2) + 4 * half);
        const bool keep = mmf_keep(dkey, q * (unsigned)Tk + kcol, a.drop_thresh);
        s[r] = keep ? p * a.inv_keep : 0.f;
        ds[r] = p * ((keep ? dp[r] * a.inv_keep : 0.f) + dm[r]);
      } else {
        s[r] = p;
        ds[r] = p * dp[r];
      }
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const bf16x8_t pf = acc_frag(s, sub), dsf = acc_frag(ds, sub);
      if (sub == 0) {
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 0>(s1, tlo, thi), pf, dv[0], 0, 0, 0);      // dV^T += dO^T.P
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 0>(s0, tlo, thi), dsf, dk[0], 0, 0, 0);     // dK^T += Q^T.dS
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 1>(s1, tlo, thi), pf, dv[1], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 1>(s0, tlo, thi), dsf, dk[1], 0, 0, 0);
        if constexpr (DT == 3) {
          dv[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 2>(s1, tlo, thi), pf, dv[2], 0, 0, 0);
          dk[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 0, 2>(s0, tlo, thi), dsf, dk[2], 0, 0, 0);
        }
      } else {
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 0>(s1, tlo, thi), pf, dv[0], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 0>(s0, tlo, thi), dsf, dk[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 1>(s1, tlo, thi), pf, dv[1], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 1>(s0, tlo, thi), dsf, dk[1], 0, 0, 0);
        if constexpr (DT == 3) {
          dv[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 2>(s1, tlo, thi), pf, dv[2], 0, 0, 0);
          dk[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH, 1, 2>(s0, tlo, thi), dsf, dk[2], 0, 0, 0);
        }
      }
    }
  }

  // two merge passes (dK^T, then dV^T) so that a wave's partial fits over its two slices
  float* red = reinterpret_cast<float*>(s0);
  if (wave != 0) put_partial<DT>(red, dk, lane);
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) add_partial<DT>(reinterpret_cast<const float*>(smem + w * REGION), dk, lane, 1.f);
  }
  __syncthreads();
  if (wave != 0) put_partial<DT>(red, dv, lane);
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w) add_partial<DT>(reinterpret_cast<const float*>(smem + w * REGION), dv, lane, 1.f);
  store_rows_lds<DH>(dk, a.scale, static_cast<unsigned short*>(P.dK) + koff, P.ldk, 0, Tk, lane, s0);
  store_rows_lds<DH>(dv, 1.f, static_cast<unsigned short*>(P.dV) + voff, P.ldv, 0, Tk, lane, s0);
}

int fill_narrow(NarrowArgs& a, const mmf_attn_problem* problems, const int* which, int n, float scale, float drop_p,
                const uint64_t* rng_state, uint32_t site) {
  a.nprob = n; a.scale = scale;
  a.drop_thresh = (drop_p > 0.f && rng_state) ? mmf_drop_thresh(drop_p) : 0u;
  a.inv_keep = a.drop_thresh ? 1.f / (1.f - (float)a.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  a.site = site;
  a.rng_state = reinterpret_cast<const unsigned long long*>(rng_state);
  int total = 0;
  for (int k = 0; k < n; ++k) {
    const mmf_attn_problem& q = problems[which[k]];
    a.blk_start[k] = total;
    a.orig[k] = (short)which[k];
    a.p[k] = q;
    total += q.B * q.H;
  }
  a.blk_start[n] = total;
  return total;
}

}  // namespace

#define NARROW_LAUNCH(kern, name)                                                                                  \
  do {                                                                                                             \
    NarrowArgs a;                                                                                                  \
    const int total = fill_narrow(a, problems, which, n, scale, drop_p, rng_state, site);                          \
    const bool dr = a.drop_thresh != 0u;                                                                           \
    if (head_dim == 96) { if (dr) hipLaunchKernelGGL((kern<96, true>), dim3(total), dim3(NT), 0, s, a);            \
                          else    hipLaunchKernelGGL((kern<96, false>), dim3(total), dim3(NT), 0, s, a); }         \
    else                { if (dr) hipLaunchKernelGGL((kern<64, true>), dim3(total), dim3(NT), 0, s, a);            \
                          else    hipLaunchKernelGGL((kern<64, false>), dim3(total), dim3(NT), 0, s, a); }         \
    MMF_CHECK_LAUNCH(name);                                                                                        \
    return MMF_OK;                                                                                                 \
  } while (0)

// problems[which[0..n)]: the caller's problem indices stay the dropout stream ids
int mmf_attn_fwd_narrowq_launch(const mmf_attn_problem* problems, const int* which, int n, int head_dim, float scale, float drop_p,
                                const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  NARROW_LAUNCH(attn_fwd_narrowq_kernel, "mmf_attn_fwd_grouped(narrow q)");
}
int mmf_attn_bwd_dq_narrowq_launch(const mmf_attn_problem* problems, const int* which, int n, int head_dim, float scale, float drop_p,
                                   const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  NARROW_LAUNCH(attn_bwd_dq_narrowq_kernel, "mmf_attn_bwd_grouped(dq, narrow q)");
}
int mmf_attn_bwd_dkv_narrowk_launch(const mmf_attn_problem* problems, const int* which, int n, int head_dim, float scale, float drop_p,
                                    const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  NARROW_LAUNCH(attn_bwd_dkv_narrowk_kernel, "mmf_attn_bwd_grouped(dkv, narrow k)");
}
