// Grouped fused attention for the fusion path (q*scale, QK^T, softmax, P.V of
// F.multi_head_attention_forward as called at models/fusion_layers.py:161-163,204): forward with
// log-sum-exp, and a recompute backward (dQ kernel + dK/dV kernel, no atomics, deterministic).
//
// All nine attentions of a MulT pass (6 cross + 3 self, unequal Tq/Tk) go out as ONE launch.
// MFMA: v_mfma_f32_32x32x16_bf16.  The products are oriented so that every probability tile stays
// in registers between its two uses (guide section 3, "An accumulator tile as the next MFMA's
// operand"):
//   forward / dQ  : S^T[key][q] = K.Q^T  -> the query is the LANE: row max / row sum / LSE / delta
//                   are per-lane scalars (one cross-half shuffle), P^T feeds O^T = V^T.P^T and
//                   dS^T feeds dQ^T = K^T.dS^T straight from the accumulator registers;
//   dK/dV         : S[q][key] = Q.K^T    -> the key is the lane; each wave keeps dK^T, dV^T of its
//                   32 keys in accumulators while the workgroup sweeps the query tiles.
// K/V (forward, dQ) and Q/dO (dK/dV) tiles are staged HBM -> registers -> LDS, double-buffered,
// one barrier per tile; rows are padded by 16 B ([.][DH+8]) so the ds_read_b128 row reads are
// conflict-free, the transposed operands come from the same image via ds_read_b64_tr_b16.
// Softmax runs in the exp2 domain in f32; masked (>= Tk) keys get -1e30.
#include "mmf_internal.h"
#include <stdlib.h>
#include "attn_helpers.h"

namespace {

struct AttnArgs {
  int nprob;
  float scale;
  int debug;      // timing ablations (MMF_ATTN_DEBUG, results wrong by design): 1 no exp, 2 no PV, 4 no QK, 8 no K/V reload
  // attention-probability dropout (0 threshold = off): mask = hash(*rng_state, site, problem/b/h, q*Tk + key)
  unsigned drop_thresh, site;
  float inv_keep;
  const unsigned long long* rng_state;
  int blk_start[MMF_ATTN_MAX_PROBLEMS + 1];
  mmf_attn_problem p[MMF_ATTN_MAX_PROBLEMS];
};

__device__ __forceinline__ int find_problem(const AttnArgs& a, int bid) {
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  return pi;
}

// ================================================================================================
// forward: workgroup = 128 query rows of one (b, h); wave = 32 query rows; KV tiles of 64 keys
// ================================================================================================
#ifdef MMF_LEGACY_KERNELS   // first-generation kernels: built with `make LEGACY=1` for A/B runs only
template <int DH, bool DROP>
__global__ __launch_bounds__(NT)
void attn_fwd_kernel(const AttnArgs a) {
  constexpr int KS = DH / 16, DT = DH / 32;
  constexpr int TILE_B = 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];      // [buf][K | V]

  const int pi = find_problem(a, blockIdx.x);
  const mmf_attn_problem& P = a.p[pi];
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int nqt = (Tq + 127) / 128;
  const int idx = blockIdx.x - a.blk_start[pi];
  const int bh = idx / nqt, qt = idx % nqt;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int q0 = qt * 128 + wave * 32;

  const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;

  // wave-private [32][DH+8] slice inside ring buffer 1: free until the first store into that buffer,
  // which happens behind the prologue barrier below; free again after the k-loop's last barrier
  char* slice = smem + 2 * TILE_B + wave * (32 * (DH + 8) * 2);
  bf16x8_t qf[KS];
  load_row_frags_lds<DH>(qf, Qg, P.ldq, q0, Tq, lane, slice);

  f32x16_t o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = NEG_BIG, l = 0.f;
  const float c = a.scale * LOG2E;
  constexpr bool drop = DROP;
  const unsigned dkey = drop ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pi * 4096 + bh)) : 0u;
  const unsigned qidx = (unsigned)(q0 + (lane & 31)) * (unsigned)Tk;

  TileStage<DH, 64> sk, sv;
  const int ntiles = (Tk + 63) / 64;
  sk.load(Kg, P.ldk, 0, Tk, tid);
  sv.load(Vg, P.ldv, 0, Tk, tid);
  sk.store(smem, tid);
  sv.store(smem + TILE_B, tid);
  __syncthreads();

  int cur = 0;
  for (int j = 0; j < ntiles; ++j) {
    const bool more = j + 1 < ntiles && !(a.debug & 8);
    if (more) {
      sk.load(Kg, P.ldk, (j + 1) * 64, Tk, tid);
      sv.load(Vg, P.ldv, (j + 1) * 64, Tk, tid);
    }
    const char* sK = smem + cur * 2 * TILE_B;
    const char* sV = sK + TILE_B;
    const int kb = j * 64;

    f32x16_t s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      if (!(a.debug & 4)) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sK, 32 * kt, ks, lane), qf[ks], s[kt], 0, 0, 0);
      }
    }
    // scale into the exp2 domain, mask the ragged tail, running max
    const bool ragged = kb + 64 > Tk;
    float mx = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = s[kt][r] * c;
        if (ragged) {
          const int key = kb + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
          v = key < Tk ? v : NEG_BIG;
        }
        s[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(m, mx);
    const float alpha = fast_exp2(m - mnew);
    m = mnew;
    float rs = 0.f;
    if (!(a.debug & 1)) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[kt][r] - mnew);
        s[kt][r] = p;
        rs += p;
      }
    }
    l = l * alpha + rs;
    if (drop) {              // nn.MultiheadAttention(dropout=p): drop/rescale the probabilities fed to P.V only
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned key = (unsigned)(kb + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half);
          s[kt][r] = mmf_keep(dkey, qidx + key, a.drop_thresh) ? s[kt][r] * a.inv_keep : 0.f;
        }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    // O^T += V^T . P^T
    if (!(a.debug & 2))
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8_t pf = acc_frag(s[kt], ss);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(sV, 32 * kt + 16 * ss, 32 * dt, lane), pf, o[dt], 0, 0, 0);
      }
    if (more) {
      char* d = smem + (cur ^ 1) * 2 * TILE_B;
      sk.store(d, tid);
      sv.store(d + TILE_B, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  unsigned short* Og = static_cast<unsigned short*>(P.O) + (size_t)b * Tq * P.ldo + h * DH;
  store_rows_lds<DH>(o, inv, Og, P.ldo, q0, Tq, lane, slice);
  const int qrow = q0 + (lane & 31);
  if (half == 0 && qrow < Tq) P.LSE[(size_t)bh * Tq + qrow] = m * LN2 + __logf(l);
}

// ================================================================================================
// backward, dQ (+ delta): same sweep as the forward; per KV tile
//   S^T = K.Q^T, P^T = exp(S^T*scale - LSE), dP^T = V.dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T.dS^T
// ================================================================================================
template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 2)      // 2 workgroups per CU: <= 256 VGPR+AGPR per lane
void attn_bwd_dq_kernel(const AttnArgs a) {
  constexpr int KS = DH / 16, DT = DH / 32;
  constexpr int TILE_B = 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];

  const int pi = find_problem(a, blockIdx.x);
  const mmf_attn_problem& P = a.p[pi];
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int nqt = (Tq + 127) / 128;
  const int idx = blockIdx.x - a.blk_start[pi];
  const int bh = idx / nqt, qt = idx % nqt;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int q0 = qt * 128 + wave * 32;
  const int qrow = q0 + (lane & 31);

  const size_t qoff = (size_t)b * Tq * P.ldq + h * DH, ooff = (size_t)b * Tq * P.ldo + h * DH;
  const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + qoff;
  const unsigned short* Og = static_cast<const unsigned short*>(P.O) + ooff;
  const unsigned short* dOg = static_cast<const unsigned short*>(P.dO) + ooff;
  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;

  char* slice = smem + 2 * TILE_B + wave * (32 * (DH + 8) * 2);     // see attn_fwd_kernel
  bf16x8_t qf[KS], dof[KS];
  load_row_frags_lds<DH>(qf, Qg, P.ldq, q0, Tq, lane, slice);
  load_row_frags_lds<DH>(dof, dOg, P.ldo, q0, Tq, lane, slice);
  // delta[q] = sum_d dO[q][d] * O[q][d]  (each half-lane pair covers the row once)
  float delta = 0.f;
  {
    bf16x8_t of[KS];
    load_row_frags_lds<DH>(of, Og, P.ldo, q0, Tq, lane, slice);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4_t x = __builtin_bit_cast(u32x4_t, of[ks]), y = __builtin_bit_cast(u32x4_t, dof[ks]);
#pragma unroll
      for (int e = 0; e < 4; ++e) delta += bf16lo(x[e]) * bf16lo(y[e]) + bf16hi(x[e]) * bf16hi(y[e]);
    }
    delta += __shfl_xor(delta, 32, 64);
    if (half == 0 && qrow < Tq) P.delta[(size_t)bh * Tq + qrow] = delta;
  }
  const float c = a.scale * LOG2E;
  const float lse2 = (qrow < Tq ? P.LSE[(size_t)bh * Tq + qrow] : 0.f) * LOG2E;
  constexpr bool drop = DROP;
  const unsigned dkey = drop ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pi * 4096 + bh)) : 0u;
  const unsigned qidx = (unsigned)qrow * (unsigned)Tk;

  f32x16_t dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  TileStage<DH, 64> sk, sv;
  const int ntiles = (Tk + 63) / 64;
  sk.load(Kg, P.ldk, 0, Tk, tid);
  sv.load(Vg, P.ldv, 0, Tk, tid);
  sk.store(smem, tid);
  sv.store(smem + TILE_B, tid);
  __syncthreads();

  int cur = 0;
  for (int j = 0; j < ntiles; ++j) {
    const bool more = j + 1 < ntiles;
    if (more) {
      sk.load(Kg, P.ldk, (j + 1) * 64, Tk, tid);
      sv.load(Vg, P.ldv, (j + 1) * 64, Tk, tid);
    }
    const char* sK = smem + cur * 2 * TILE_B;
    const char* sV = sK + TILE_B;
    const int kb = j * 64;
    const bool ragged = kb + 64 > Tk;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sK, 32 * kt, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sV, 32 * kt, ks, lane), dof[ks], dp, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = fast_exp2(s[r] * c - lse2);
        const int key = kb + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ragged) p = key < Tk ? p : 0.f;
        float dpv = dp[r];                                // d(P_dropped) -> dP through the same mask
        if (drop) dpv = mmf_keep(dkey, qidx + (unsigned)key, a.drop_thresh) ? dpv * a.inv_keep : 0.f;
        s[r] = p * (dpv - delta);                         // dS^T (scale applied at the store)
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8_t dsf = acc_frag(s, ss);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(sK, 32 * kt + 16 * ss, 32 * dt, lane), dsf, dq[dt], 0, 0, 0);
      }
    }
    if (more) {
      char* d = smem + (cur ^ 1) * 2 * TILE_B;
      sk.store(d, tid);
      sv.store(d + TILE_B, tid);
    }
    __syncthreads();
    cur ^= 1;
  }
  unsigned short* dQg = static_cast<unsigned short*>(P.dQ) + qoff;
  store_rows_lds<DH>(dq, a.scale, dQg, P.ldq, q0, Tq, lane, slice);
}

// ================================================================================================
// backward, dK and dV: workgroup = 128 keys of one (b, h); wave = 32 keys kept in registers with
// their dK^T / dV^T accumulators; sweep over query tiles of 64 rows (Q and dO staged in LDS).
//   S = Q.K^T, P = exp(S*scale - LSE[q]), dV^T += dO^T.P, dP = dO.V^T, dS = P (dP - delta[q]),
//   dK^T += Q^T.dS
// ================================================================================================
template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 2)      // 2 workgroups per CU: <= 256 VGPR+AGPR per lane
void attn_bwd_dkv_kernel(const AttnArgs a) {
  constexpr int KS = DH / 16, DT = DH / 32;
  constexpr int TILE_B = 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B + 2 * 2 * 64 * 4];   // [buf][Q | dO], [buf][lse2 | delta]
  float* sstat = reinterpret_cast<float*>(smem + 4 * TILE_B);

  const int pi = find_problem(a, blockIdx.x);
  const mmf_attn_problem& P = a.p[pi];
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int nkt = (Tk + 127) / 128;
  const int idx = blockIdx.x - a.blk_start[pi];
  const int bh = idx / nkt, ktile = idx % nkt;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int k0 = ktile * 128 + wave * 32;
  const bool wave_active = k0 < Tk;                                   // wave-uniform

  const size_t ooff = (size_t)b * Tq * P.ldo + h * DH;
  const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
  const unsigned short* dOg = static_cast<const unsigned short*>(P.dO) + ooff;
  const size_t koff = (size_t)b * Tk * P.ldk + h * DH, voff = (size_t)b * Tk * P.ldv + h * DH;
  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + koff;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + voff;
  const float* LSEg = P.LSE + (size_t)bh * Tq;
  const float* DELg = P.delta + (size_t)bh * Tq;

  char* slice = smem + 2 * TILE_B + wave * (32 * (DH + 8) * 2);     // see attn_fwd_kernel
  bf16x8_t kf[KS], vf[KS];
  load_row_frags_lds<DH>(kf, Kg, P.ldk, k0, Tk, lane, slice);
  load_row_frags_lds<DH>(vf, Vg, P.ldv, k0, Tk, lane, slice);

  f32x16_t dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const float c = a.scale * LOG2E;
  constexpr bool drop = DROP;
  const unsigned dkey = drop ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pi * 4096 + bh)) : 0u;
  const unsigned kcol = (unsigned)(k0 + (lane & 31));

  TileStage<DH, 64> sq, sdo;
  float stat = 0.f;                                                   // threads 0..63: lse2, 64..127: delta
  auto load_stat = [&](int qb) {
    if (tid < 128) {
      const int q = qb + (tid & 63);
      if (tid < 64) stat = q < Tq ? LSEg[q] * LOG2E : 1.0e30f;        // rows >= Tq: P = exp2(-inf) = 0
      else          stat = q < Tq ? DELg[q] : 0.f;
    }
  };
  const int ntiles = (Tq + 63) / 64;
  sq.load(Qg, P.ldq, 0, Tq, tid);
  sdo.load(dOg, P.ldo, 0, Tq, tid);
  load_stat(0);
  sq.store(smem, tid);
  sdo.store(smem + TILE_B, tid);
  if (tid < 128) sstat[tid] = stat;
  __syncthreads();

  int cur = 0;
  for (int j = 0; j < ntiles; ++j) {
    const bool more = j + 1 < ntiles;
    const char* sQ = smem + cur * 2 * TILE_B;
    const char* sdO = sQ + TILE_B;
    const float* sl = sstat + cur * 128;
    if (wave_active) {
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sQ, 32 * qs, ks, lane), kf[ks], s, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sdO, 32 * qs, ks, lane), vf[ks], dp, 0, 0, 0);
        f32x16_t ds;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(sl + 32 * qs + 8 * g + 4 * half);
          const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(sl + 64 + 32 * qs + 8 * g + 4 * half);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float p = fast_exp2(s[4 * g + i] * c - l4[i]);
            float pd = p, dpv = dp[4 * g + i];
            if (drop) {
              const unsigned q = (unsigned)(j * 64 + 32 * qs + 8 * g + 4 * half + i);
              const bool keep = mmf_keep(dkey, q * (unsigned)Tk + kcol, a.drop_thresh);
              pd = keep ? p * a.inv_keep : 0.f;
              dpv = keep ? dpv * a.inv_keep : 0.f;
            }
            s[4 * g + i] = pd;                            // dV^T += dO^T . P_dropped
            ds[4 * g + i] = p * (dpv - d4[i]);
          }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8_t pf = acc_frag(s, ss), dsf = acc_frag(ds, ss);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(sdO, 32 * qs + 16 * ss, 32 * dt, lane), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(sQ, 32 * qs + 16 * ss, 32 * dt, lane), dsf, dk[dt], 0, 0, 0);
          }
        }
      }
    }
    if (more) {       // the next Q/dO tile is fetched here, not ahead of the MFMAs: holding it in registers
      sq.load(Qg, P.ldq, (j + 1) * 64, Tq, tid);       // across the compute phase costs 25 VGPRs and
      sdo.load(dOg, P.ldo, (j + 1) * 64, Tq, tid);     // the second workgroup on the CU (<= 256
      load_stat((j + 1) * 64);                         // registers) hides this latency better
      char* d = smem + (cur ^ 1) * 2 * TILE_B;
      sq.store(d, tid);
      sdo.store(d + TILE_B, tid);
      if (tid < 128) sstat[(cur ^ 1) * 128 + tid] = stat;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (wave_active) {
    store_rows_lds<DH>(dk, a.scale, static_cast<unsigned short*>(P.dK) + koff, P.ldk, k0, Tk, lane, slice);
    store_rows_lds<DH>(dv, 1.f, static_cast<unsigned short*>(P.dV) + voff, P.ldv, k0, Tk, lane, slice);
  }
}

#endif  // MMF_LEGACY_KERNELS

int validate(const char* who, const mmf_attn_problem* p, int n, int head_dim, bool bwd) {
  if (!p || n <= 0 || n > MMF_ATTN_MAX_PROBLEMS) MMF_FAIL(MMF_E_SHAPE, "%s: num_problems=%d out of range", who, n);
  if (head_dim != 64 && head_dim != 96) MMF_FAIL(MMF_E_UNSUPPORTED, "%s: head_dim=%d (supported: 64, 96)", who, head_dim);
  for (int i = 0; i < n; ++i) {
    const mmf_attn_problem& q = p[i];
    if (q.B <= 0 || q.H <= 0 || q.Tq <= 0 || q.Tk <= 0) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: B=%d H=%d Tq=%d Tk=%d", who, i, q.B, q.H, q.Tq, q.Tk);
    const int w = q.H * head_dim;
    if (q.ldq < w || q.ldk < w || q.ldv < w || q.ldo < w || (q.ldq & 7) || (q.ldk & 7) || (q.ldv & 7) || (q.ldo & 7))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: row strides must be >= H*head_dim and multiples of 8", who, i);
    if (!q.Q || !q.K || !q.V || !q.O || !q.LSE) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: null operand", who, i);
    if (!mmf_aligned16(q.Q) || !mmf_aligned16(q.K) || !mmf_aligned16(q.V) || !mmf_aligned16(q.O))
      MMF_FAIL(MMF_E_ALIGN, "%s[%d]: Q/K/V/O must be 16-byte aligned", who, i);
    if (bwd) {
      if (!q.dO || !q.delta || !q.dQ || !q.dK || !q.dV) MMF_FAIL(MMF_E_SHAPE, "%s[%d]: null gradient operand", who, i);
      if (!mmf_aligned16(q.dO) || !mmf_aligned16(q.dQ) || !mmf_aligned16(q.dK) || !mmf_aligned16(q.dV))
        MMF_FAIL(MMF_E_ALIGN, "%s[%d]: gradient pointers must be 16-byte aligned", who, i);
    }
  }
  return MMF_OK;
}

#ifdef MMF_LEGACY_KERNELS
int fill_args(AttnArgs& a, const mmf_attn_problem* p, int n, float scale, bool by_keys, float drop_p,
              const uint64_t* rng_state, uint32_t site) {
  a.nprob = n; a.scale = scale;
  a.drop_thresh = (drop_p > 0.f && rng_state) ? mmf_drop_thresh(drop_p) : 0u;
  a.inv_keep = a.drop_thresh ? 1.f / (1.f - (float)a.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  a.site = site;
  a.rng_state = reinterpret_cast<const unsigned long long*>(rng_state);
  const char* dbg = getenv("MMF_ATTN_DEBUG");
  a.debug = dbg ? atoi(dbg) : 0;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    a.blk_start[i] = total;
    const int T = by_keys ? p[i].Tk : p[i].Tq;
    total += p[i].B * p[i].H * ((T + 127) / 128);
    a.p[i] = p[i];
  }
  a.blk_start[n] = total;
  return total;
}
#endif  // MMF_LEGACY_KERNELS

}  // namespace

// second generation (attention2.hip)
int mmf_attn_fwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s);
int mmf_attn_bwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s);
// fourth-generation forward (attention4.hip: 8-wave ping-pong for the wide problems, second generation for the rest)
int mmf_attn_fwd4_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s);
#ifdef MMF_LEGACY_KERNELS
int mmf_attn_fwd3_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s);
#endif
// 0 automatic, 1 first generation, 2 second generation, 3 third-generation forward, 4 fourth-generation forward
static int g_attn_impl = 0;
static const int g_attn_fwd_gen = [] { const char* e = getenv("MMF_ATTN_FWD_GEN"); return e ? atoi(e) : 2; }();   // automatic choice of the forward
extern "C" int mmf_attn_select_impl(int impl) {
  if (impl < 0 || impl > 4) MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_attn_select_impl: impl=%d (0 auto, 1, 2, 3, 4)", impl);
#ifndef MMF_LEGACY_KERNELS
  if (impl == 1 || impl == 3)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_attn_select_impl: generation %d (superseded) is only in builds made with `make LEGACY=1`", impl);
#endif
  g_attn_impl = impl;
  return MMF_OK;
}

extern "C" int mmf_attn_fwd_grouped_ex(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                       float scale, float dropout_p, const uint64_t* rng_state, uint32_t site,
                                       void* stream) {
  if (int rc = validate("mmf_attn_fwd_grouped", problems, num_problems, head_dim, false)) return rc;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f || (dropout_p > 0.f && !rng_state))
    MMF_FAIL(MMF_E_SHAPE, "mmf_attn_fwd_grouped_ex: dropout needs 0 <= p < 1 and an rng_state");
  if (g_attn_impl == 4 || (g_attn_impl == 0 && g_attn_fwd_gen == 4))
    return mmf_attn_fwd4_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site,
                                static_cast<hipStream_t>(stream));
#ifndef MMF_LEGACY_KERNELS
  return mmf_attn_fwd2_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site,
                              static_cast<hipStream_t>(stream));
#else
  if (g_attn_impl == 3)
    return mmf_attn_fwd3_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site,
                                static_cast<hipStream_t>(stream));
  if (g_attn_impl != 1 && !getenv("MMF_ATTN_DEBUG"))
    return mmf_attn_fwd2_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site,
                                static_cast<hipStream_t>(stream));
  AttnArgs a;
  const int total = fill_args(a, problems, num_problems, scale, false, dropout_p, rng_state, site);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool dr = a.drop_thresh != 0u;
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_fwd_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_fwd_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
  MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped");
  return MMF_OK;
#endif
}

extern "C" int mmf_attn_bwd_grouped_ex(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                       float scale, float dropout_p, const uint64_t* rng_state, uint32_t site,
                                       void* stream) {
  if (int rc = validate("mmf_attn_bwd_grouped", problems, num_problems, head_dim, true)) return rc;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f || (dropout_p > 0.f && !rng_state))
    MMF_FAIL(MMF_E_SHAPE, "mmf_attn_bwd_grouped_ex: dropout needs 0 <= p < 1 and an rng_state");
  hipStream_t s = static_cast<hipStream_t>(stream);
#ifndef MMF_LEGACY_KERNELS
  return mmf_attn_bwd2_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site, s);
#else
  if (g_attn_impl != 1 && !getenv("MMF_ATTN_DEBUG"))
    return mmf_attn_bwd2_launch(problems, num_problems, head_dim, scale, dropout_p, rng_state, site, s);
  AttnArgs a;
  int total = fill_args(a, problems, num_problems, scale, false, dropout_p, rng_state, site);   // dQ (+ delta) first
  const bool dr = a.drop_thresh != 0u;
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_bwd_dq_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dq_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL((attn_bwd_dq_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dq_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
  MMF_CHECK_LAUNCH("mmf_attn_bwd_grouped(dq)");
  total = fill_args(a, problems, num_problems, scale, true, dropout_p, rng_state, site);        // then dK/dV (reads delta)
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dkv_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
  MMF_CHECK_LAUNCH("mmf_attn_bwd_grouped(dkv)");
  return MMF_OK;
#endif
}

extern "C" int mmf_attn_fwd_grouped(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                    float scale, void* stream) {
  return mmf_attn_fwd_grouped_ex(problems, num_problems, head_dim, scale, 0.f, nullptr, 0u, stream);
}

extern "C" int mmf_attn_bwd_grouped(const mmf_attn_problem* problems, int num_problems, int head_dim,
                                    float scale, void* stream) {
  return mmf_attn_bwd_grouped_ex(problems, num_problems, head_dim, scale, 0.f, nullptr, 0u, stream);
}
