// Attention forward, third generation: one wave per SIMD with the whole 512-register file, software-pipelined
// across 64-key tiles so that the MFMA pipe and the VALU work on different tiles at the same time.
//
// Why: the second generation (attention2.hip) is bound by its per-wave instruction stream, not by HBM or the DMA
// (MMF_ATTN2_DEBUG ablation: 58.9 of 63.1 us remain with the K/V DMA disabled; SQ_WAIT_ANY = 49 % of wave cycles at
// one wave per SIMD).  Within one wave QK^T -> softmax -> P.V of a block are a dependent chain, so the matrix pipe
// idles during the ~600-cycle softmax and the VALU idles during the MFMAs; two waves per SIMD only half hide that.
// Here a wave keeps TWO score sets: while the VALU turns tile j's scores into probabilities, the MFMA pipe computes
// tile j+1's QK^T into the other set (guide "4-wave, one-wave-per-SIMD" structure, compiler-scheduled):
//
//     per tile j :  [barrier: tile j+1 landed]  DMA tile j+3
//        phase A :  Sn_a = K_a(j+1).Q^T   (12 MFMA)   ||   P_a = exp2(S_a(j) c - m), row sums     (VALU)
//        phase B :  Sn_b = K_b(j+1).Q^T   (12 MFMA)   ||   P_b                                     (VALU)
//        phase C :  O^T += V^T(j).P^T     (24 MFMA)   ||   row maxima of Sn                         (VALU)
//        decide  :  raise the running maximum for tile j+1 if needed (rare branch; nothing is pending here:
//                   every P of tile j has been multiplied into O), swap S <-> Sn
//
// Workgroup = 4 waves = 256 query rows of one (b, h), two 32-row query blocks per wave; K/V tiles by LDS-DMA into a
// 4-stage ring (106 KiB): tile j (V), j+1 (K) live, j+2 in flight, j+3 being issued.  Layouts, the deferred
// maximum and the work order are those of attention2.hip.  Used for the problems with Tq > 64 and Tk > 64; the
// narrow (x30) problems are HBM streaming problems and stay on the second generation at two workgroups per CU.
#include "attn2_common.h"

namespace {

constexpr int NST = 4;                 // LDS ring stages

// O^T += V^T . P^T for the 64 keys of one tile: 4 DT fragment steps (16-key group G = N / DT, head-dim block
// N % DT), fragment N + 1 in flight under step N's MFMAs.  p[G / 2][qb][G % 2] are the packed probabilities.
template <int DH, int NQ, int N>
struct Pv3Step {
  static constexpr int DT = DH / 32, NF = 4 * DT;
  static __device__ __forceinline__ void run(unsigned va, s16x4_t lo, s16x4_t hi, const bf16x8_t (&p)[2][NQ][2],
                                             f32x16_t (&o)[NQ][DT]) {
    s16x4_t nlo, nhi;
    if constexpr (N + 1 < NF) tr_issue<DH, (N + 1) / DT, (N + 1) % DT>(va, nlo, nhi);
    tr_wait<(N + 1 < NF) ? 2 : 0>(lo, hi);
    const bf16x8_t vf = join(lo, hi);
    constexpr int G = N / DT;
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
      o[qb][N % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, p[G / 2][qb][G % 2], o[qb][N % DT], 0, 0, 0);
    if constexpr (N + 1 < NF) Pv3Step<DH, NQ, N + 1>::run(va, nlo, nhi, p, o);
  }
};

template <int DH, bool DROP, int NQ>
__device__ __forceinline__ void fwd3_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
                                          const int qs, char* smem) {
  constexpr int KS = DH / 16, DT = DH / 32, SB = (DH + 8) * 2, TILE_B = 64 * SB, STAGE_B = 2 * TILE_B;
  constexpr int NQA = NQ > 0 ? NQ : 1;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;
  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Kg), 0, Tk * P.ldk * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Vg), 0, Tk * P.ldv * 2, 0x00020000);
  auto issue = [&](int j) { dma_pair<DH>(rsK, rsV, P.ldk, P.ldv, smem + (j & (NST - 1)) * STAGE_B, j, wave, lane); };
  const int ntiles = (Tk + 63) / 64;
  issue(0);
  if (ntiles > 1) issue(1);
  if (ntiles > 2) issue(2);

  // Q fragments through the wave's slice of stage 3 (free until the DMA of tile 3, issued behind the first loop barrier)
  char* slice = smem + (NST - 1) * STAGE_B + wave * (32 * SB);
  bf16x8_t qf[NQA][KS];
  if constexpr (NQ > 0) {
    const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) load_row_frags_lds<DH>(qf[qb], Qg, P.ldq, qs + 128 * qb, Tq, lane, slice);
  }
  f32x16_t o[NQA][DT];
  float m[NQA], l[NQA];
#pragma unroll
  for (int qb = 0; qb < NQA; ++qb) {
    m[qb] = NEG_BIG; l[qb] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][dt][r] = 0.f;
  }
  const float c = a.scale * LOG2E;
  const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;
  const unsigned troff = (unsigned)((4 * half + ((lane >> 2) & 3)) * SB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // S^T[key][q] = K.Q^T for the 32-key block KT of the tile in `sK` whose first key is k0: raw scores.  Keys past
  // Tk start from NEG_BIG instead of 0 (their K rows are zeros from the buffer range check), so ragged tiles need
  // no separate masking pass and no branch: the select replaces the accumulator's zero-initialisation move.
  auto qk = [&](const char* sK, auto KTc, f32x16_t (&s)[NQA], int k0) {
    constexpr int KT = decltype(KTc)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float init = (k0 + (r & 3) + 8 * (r >> 2) + 4 * half) < Tk ? 0.f : NEG_BIG;
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) s[qb][r] = init;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t kf = row_frag<DH>(sK, 32 * KT, ks, lane);
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb)
        s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);
    }
  };
  // running-maximum decision for a whole tile (both blocks), before any of its probabilities is formed
  auto decide = [&](const f32x16_t (&sa)[NQA], const f32x16_t (&sb)[NQA]) {
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      float mx = sa[qb][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sa[qb][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sb[qb][r]);
      mx = half_max(mx) * c;
      if (!__all(mx <= m[qb] + DEFER)) {                   // wave-uniform, rare after the first tile
        const float mnew = fmaxf(m[qb], mx);
        const float alpha = fast_exp2(m[qb] - mnew);
        m[qb] = mnew;
        l[qb] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
      }
    }
  };
  // probabilities of one block: p = exp2(s c - m), row sums into l, dropout, packed as the P.V operand
  auto softmax = [&](const f32x16_t (&s)[NQA], bf16x8_t (&p)[NQA][2], int k0) {
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      const float nm = -m[qb];
      f32x16_t e;
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        e[r] = fast_exp2(__builtin_fmaf(s[qb][r], c, nm));
        rs += e[r];
      }
      l[qb] += rs;
      if (DROP) {
        const unsigned qidx = (unsigned)(qs + 128 * qb + (lane & 31)) * (unsigned)Tk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned key = (unsigned)(k0 + (r & 3) + 8 * (r >> 2) + 4 * half);
          e[r] = mmf_keep(dkey, qidx + key, a.drop_thresh) ? e[r] * a.inv_keep : 0.f;
        }
      }
      p[qb][0] = acc_frag(e, 0);
      p[qb][1] = acc_frag(e, 1);
    }
  };

  f32x16_t sa[NQA], sb[NQA], na[NQA], nb[NQA];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // prologue: tiles 0..2 landed (the Q loads' own waits drain them anyway)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (NQ > 0) {
    qk(smem, std::integral_constant<int, 0>{}, sa, 0);
    qk(smem, std::integral_constant<int, 1>{}, sb, 32);
    decide(sa, sb);
  }

  // One tile: S_a / S_b hold tile j's scores (maximum already decided), N_a / N_b receive tile j+1's.
  auto iteration = [&](int j, f32x16_t (&S_a)[NQA], f32x16_t (&S_b)[NQA], f32x16_t (&N_a)[NQA], f32x16_t (&N_b)[NQA],
                       auto has_next_c) {
    constexpr bool HAS_NEXT = decltype(has_next_c)::value;
    if constexpr (HAS_NEXT) {
      // tile j+1 landed: at most tile j+2's pieces may still be in flight.  A wave issues 2 PIECES / 4 (rounded up
      // or down) pieces per tile; waiting down to the smaller count is right for every wave (the others merely
      // wait for one piece of tile j+2 as well)
      constexpr int PER_TILE = (2 * (TILE_B / 1024)) / 4;
      if (j + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_TILE) : "memory");
      else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // tile j+1 visible; every wave is done with tile j-1
      asm volatile("" ::: "memory");
      if (j + 3 < ntiles) issue(j + 3);
    }
    if constexpr (NQ > 0) {
      const char* sKn = smem + ((j + 1) & (NST - 1)) * STAGE_B;
      const unsigned va = smem_lds + (j & (NST - 1)) * STAGE_B + TILE_B + troff;
      const int kb = j * 64;
      bf16x8_t p[2][NQA][2];
      s16x4_t lo, hi;
      if constexpr (HAS_NEXT) qk(sKn, std::integral_constant<int, 0>{}, N_a, kb + 64);   // phase A
      softmax(S_a, p[0], kb);
      if constexpr (HAS_NEXT) qk(sKn, std::integral_constant<int, 1>{}, N_b, kb + 96);   // phase B
      tr_issue<DH, 0, 0>(va, lo, hi);
      softmax(S_b, p[1], kb + 32);
      Pv3Step<DH, NQ, 0>::run(va, lo, hi, p, o);                                          // phase C
      if constexpr (HAS_NEXT) decide(N_a, N_b);
    }
  };
  // two tiles per trip so that the score sets swap roles without register copies
  int j = 0;
  for (; j + 2 < ntiles; j += 2) {
    iteration(j, sa, sb, na, nb, std::true_type{});
    iteration(j + 1, na, nb, sa, sb, std::true_type{});
  }
  if (j + 1 < ntiles) {
    iteration(j, sa, sb, na, nb, std::true_type{});
    iteration(j + 1, na, nb, sa, sb, std::false_type{});
  } else {
    iteration(j, sa, sb, na, nb, std::false_type{});
  }

  if constexpr (NQ > 0) {
    // stage of tile (ntiles): never written (no DMA beyond the last tile) and not read by anyone now
    char* oslice = smem + (ntiles & (NST - 1)) * STAGE_B + wave * (32 * SB);
    unsigned short* Og = static_cast<unsigned short*>(P.O) + (size_t)b * Tq * P.ldo + h * DH;
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      const float lt = half_sum(l[qb]);
      store_rows_lds<DH>(o[qb], 1.f / lt, Og, P.ldo, qs + 128 * qb, Tq, lane, oslice);
      const int qrow = qs + 128 * qb + (lane & 31);
      if (half == 0 && qrow < Tq) P.LSE[(size_t)bh * Tq + qrow] = m[qb] * LN2 + __logf(lt);
    }
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 1)
void attn_fwd3_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE_B];
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi], rpc = a.rpc[pi];
  const int bh = item / nchunk, q0 = (item % nchunk) * rpc;
  const int nb = (min(P.Tq, q0 + rpc) - q0 + 31) >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nq = (wave < nb) + (wave + 4 < nb);
  const int qs = q0 + 32 * wave, pidx = a.orig[pi];
  if (nq == 2)      fwd3_wave<DH, DROP, 2>(a, P, pidx, bh, qs, smem);
  else if (nq == 1) fwd3_wave<DH, DROP, 1>(a, P, pidx, bh, qs, smem);
  else              fwd3_wave<DH, DROP, 0>(a, P, pidx, bh, qs, smem);
}

}  // namespace

int mmf_attn_fwd2_launch_indexed(const mmf_attn_problem* problems, const int* idx, int n, int head_dim, float scale,
                                 float drop_p, const uint64_t* rng_state, uint32_t site, hipStream_t s);

// Wide problems (Tq > 64 and Tk > 64) on the third generation, the rest on the second; `orig` keeps the caller's
// problem index so that the dropout streams match the backward kernels.
int mmf_attn_fwd3_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  if (int rc = check_ranges("mmf_attn_fwd_grouped", problems, n)) return rc;
  mmf_attn_problem wide[MMF_ATTN_MAX_PROBLEMS], narrow[MMF_ATTN_MAX_PROBLEMS];
  int wide_idx[MMF_ATTN_MAX_PROBLEMS], narrow_idx[MMF_ATTN_MAX_PROBLEMS], nw = 0, nn = 0;
  for (int i = 0; i < n; ++i) {
    if (problems[i].Tq > 64 && problems[i].Tk > 64) { wide[nw] = problems[i]; wide_idx[nw++] = i; }
    else                                            { narrow[nn] = problems[i]; narrow_idx[nn++] = i; }
  }
  if (nw) {
    AttnArgs2 a;
    const int total = fill_args2(a, wide, nw, scale, drop_p, rng_state, site, 256, false, true);
    for (int k = 0; k < nw; ++k) a.orig[k] = (short)wide_idx[a.orig[k]];
    const bool dr = a.drop_thresh != 0u;
    if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd3_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd3_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
    else                { if (dr) hipLaunchKernelGGL((attn_fwd3_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd3_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
    MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped(v3)");
  }
  if (nn) {
    return mmf_attn_fwd2_launch_indexed(narrow, narrow_idx, nn, head_dim, scale, drop_p, rng_state, site, s);
  }
  return MMF_OK;
}
