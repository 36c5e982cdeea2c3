// Grouped fused attention, second generation (forward): fat workgroups + LDS-DMA.
//
// What changed against attention.hip (kept for A/B behind mmf_attn_select_impl(1)) and why — the first
// generation's launch fetched 3.1x the algorithmic bytes (profiles/r01_pmc_traffic.json: 324 MB per launch)
// because the four 128-row query tiles of one (b, h) landed on four different XCDs and each re-read K/V
// through its own L2, and it was bound by per-workgroup fixed cost (profiles/r01_attention_ablation.txt):
//   * a workgroup (4 waves) covers up to 256 query rows of one (b, h); a wave owns TWO 32-row query blocks,
//     so every K fragment (ds_read_b128) and V^T fragment (ds_read_b64_tr_b16) feeds two MFMAs and K/V are
//     fetched once per 256 query rows instead of once per 128;
//   * workgroup ids are remapped so that the chunks of one (b, h) (and neighbouring heads of one batch
//     row) run on the SAME XCD (blockIdx % 8): the second chunk's K/V come from that XCD's L2;
//   * K/V tiles go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 2-stage ring, the next tile's
//     13 + 13 one-KiB pieces in flight under the current tile's MFMAs: no staging registers (the two query
//     blocks' state is 208 of the 256 registers a 2-waves-per-SIMD kernel may use) and no ds_write pass;
//     rows past Tk and the 16-B pad chunk of each row come back as zeros from the buffer range check;
//   * softmax: scores stay raw in the accumulators, p = exp2(fma(s, c, -m)) is one FMA + one v_exp per
//     score; the running maximum is only raised when some row's tile maximum exceeds it by more than
//     2^DEFER (guide T13: P <= 2^DEFER instead of <= 1, exact after normalisation because m, l and O are
//     rescaled together and the tile's P is exponentiated after the decision);
//   * problems are launched heaviest first so the 30-row problems fill the tail.
// Fragment layouts, the padded [64][DH+8] image and the P^T-as-operand orientation are those of
// attention.hip (attn_helpers.h).
#include "attn2_common.h"

namespace {

// One wave of the forward: NQ (0, 1 or 2) query blocks of 32 rows at rows qs and qs + 128.  Waves with
// NQ == 0 only take part in the K/V staging and the barriers.
template <int DH, bool DROP, int NQ, int ST>
__device__ __forceinline__ void fwd2_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
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

  // ST-stage ring (ST = 2: tile j+1 in flight under tile j; ST = 3: tiles j+1 and j+2 — a single workgroup's tile of
  // compute (~1.6 us) is shorter than a K/V DMA round trip under load, so with one tile of prefetch every tile's
  // barrier waits for memory: a 256-CU launch of ONE problem ran 3.4 us per 64-key tile against ~1.6 us of work)
  auto issue = [&](int j) { dma_pair<DH>(rsK, rsV, P.ldk, P.ldv, smem + (j % ST) * STAGE_B, j, wave, lane); };
  const int ntiles = (Tk + 63) / 64;
  issue(0);
  if (ST > 2 && ntiles > 1) issue(1);

  // Q fragments through the wave's slice of the last stage (free until the loop issues tile ST-1 behind its first barrier)
  char* slice = smem + (ST - 1) * STAGE_B + wave * (32 * SB);
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
  // this lane's part of a transposed-read address (attn_helpers.h tr_frag)
  const unsigned troff = (unsigned)((4 * half + ((lane >> 2) & 3)) * SB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // pieces of one tile this wave issues (dma_pair: piece p = wave + 4 i < 2 PIECES): the count a counted vmcnt leaves in flight
  constexpr int PIECES2 = 2 * (TILE_B / 1024);
  for (int j = 0; j < ntiles; ++j) {
    if (ST > 2 && j + 1 < ntiles) {                        // leave tile j+1's pieces in flight
      if (wave < PIECES2 % 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES2 / 4 + 1) : "memory");
      else                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES2 / 4) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile j have landed
    }
    __builtin_amdgcn_s_barrier();                          // ... everyone's; the stage of tile j-1 is free
    asm volatile("" ::: "memory");
    if (j + ST - 1 < ntiles && !(a.debug & 1)) issue(j + ST - 1);
    if constexpr (NQ > 0) {
      const char* sK = smem + (j % ST) * STAGE_B;
      const unsigned va = smem_lds + (j % ST) * STAGE_B + TILE_B + troff;
      const int kb = j * 64;
      // one 32-key block: S^T = K.Q^T (raw scores), online softmax (lane = query), O^T += V^T.P^T
      auto block = [&](auto KTc) {
        constexpr int KT = decltype(KTc)::value;
        const int k0 = kb + 32 * KT;
        if (k0 >= Tk) return;                              // wave-uniform: nothing but masked keys
        f32x16_t s[NQ];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[qb][r] = 0.f;
        mfma_prio(1);
        if constexpr (NQ == 1) {
          // all KS fragment reads in flight before the first MFMA: one exposed LDS latency per block instead of KS / 2
          bf16x8_t kf[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) kf[ks] = row_frag<DH>(sK, 32 * KT, ks, lane);
          __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise re-pairs them with the MFMAs)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[0][ks], s[0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);               // the asm V^T reads below are invisible to the compiler's lgkmcnt count
        } else {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t kf = row_frag<DH>(sK, 32 * KT, ks, lane);
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb)
              s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);
          }
        }
        mfma_prio(0);
        s16x4_t lo, hi, lo1, hi1;
        tr_issue<DH, 2 * KT, 0>(va, lo, hi);               // the first two V^T fragments land under the softmax
        tr_issue<DH, 2 * KT + 1 / DT, 1 % DT>(va, lo1, hi1);
        const bool ragged = k0 + 32 > Tk;
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
          if (ragged) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
              s[qb][r] = key < Tk ? s[qb][r] : NEG_BIG;
            }
          }
          float mx = s[qb][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][r]);
          mx = half_max(mx) * c;
          if (!__all(mx <= m[qb] + DEFER)) {               // wave-uniform: raise the running maximum
            const float mnew = fmaxf(m[qb], mx);
            const float alpha = fast_exp2(m[qb] - mnew);
            m[qb] = mnew;
            l[qb] *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
          }
          const float nm = -m[qb];
          float rs = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = fast_exp2(__builtin_fmaf(s[qb][r], c, nm));
            s[qb][r] = p;
            rs += p;
          }
          l[qb] += rs;
          if (DROP) {          // nn.MultiheadAttention(dropout=p): drop/rescale the probabilities fed to P.V only
            const unsigned qidx = (unsigned)(qs + 128 * qb + (lane & 31)) * (unsigned)Tk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned key = (unsigned)(k0 + (r & 3) + 8 * (r >> 2) + 4 * half);
              s[qb][r] = mmf_keep(dkey, qidx + key, a.drop_thresh) ? s[qb][r] * a.inv_keep : 0.f;
            }
          }
        }
        bf16x8_t pf[NQ];
        PvStep2<DH, NQ, KT, 0>::run(va, lo, hi, lo1, hi1, s, pf, o);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    }
  }

  if constexpr (NQ > 0) {
    // the stage the last tile did not use is free (every wave passed the last barrier): reuse the wave's slice there
    char* oslice = smem + (ntiles % ST) * STAGE_B + wave * (32 * SB);
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

template <int DH, bool DROP, int ST>
__global__ __launch_bounds__(NT, 2)
void attn_fwd2_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(1024))) char smem[ST * STAGE_B];         // ST = 3, DH = 96: 78 KiB, two workgroups per CU
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  // XCD x (= blockIdx % 8; blk_start is a multiple of 8) walks a contiguous range of the problem's work
  // items, so the chunks of one (b, h) and the heads of one batch row share that XCD's L2
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi], rpc = a.rpc[pi];
  const int bh = item / nchunk, q0 = (item % nchunk) * rpc;
  const int nb = (min(P.Tq, q0 + rpc) - q0 + 31) >> 5;      // 32-row query blocks in this chunk (1..8)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nq = (a.debug & 2) ? 0 : (wave < nb) + (wave + 4 < nb);   // blocks wave and wave + 4
  const int qs = q0 + 32 * wave, pidx = a.orig[pi];
  if (nq == 2)      fwd2_wave<DH, DROP, 2, ST>(a, P, pidx, bh, qs, smem);
  else if (nq == 1) fwd2_wave<DH, DROP, 1, ST>(a, P, pidx, bh, qs, smem);
  else              fwd2_wave<DH, DROP, 0, ST>(a, P, pidx, bh, qs, smem);
}


// Default since round 2 (MMF_ATTN_FWD_ROWS=256 selects the two-blocks-per-wave kernel above): one 32-row query block per wave (128 rows per workgroup), <= 168 registers, so
// THREE workgroups share a CU (3 x 52 KiB of LDS) and a wave's LDS / MFMA dependency waits have two other waves on
// its SIMD to hide under; the price is twice the K/V fragment reads and DMA per query row.
template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 3)
void attn_fwd2n_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B];
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi], rpc = a.rpc[pi];         // rpc <= 128 here
  const int bh = item / nchunk, q0 = (item % nchunk) * rpc;
  const int nb = (min(P.Tq, q0 + rpc) - q0 + 31) >> 5;      // 32-row query blocks in this chunk (1..4)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qs = q0 + 32 * wave, pidx = a.orig[pi];
  if (wave < nb && !(a.debug & 2)) fwd2_wave<DH, DROP, 1, 2>(a, P, pidx, bh, qs, smem);
  else                             fwd2_wave<DH, DROP, 0, 2>(a, P, pidx, bh, qs, smem);
}

// ================================================================================================
// backward, second generation.  Same arithmetic as attention.hip's two kernels (dQ + delta, then dK/dV; recompute
// from LSE; no atomics, deterministic); what changes is how the swept tiles reach the MFMAs:
//   * K/V (dQ kernel) and Q/dO (+ LSE, delta; dK/dV kernel) tiles arrive by LDS-DMA into a 2-stage ring, the next
//     tile in flight under the current tile's MFMAs.  The first generation fetched through registers and, in the
//     dK/dV kernel, only AFTER the MFMAs: one exposed HBM/L2 latency per tile, which made the x30 problems
//     (one active wave walking 8 tiles) cost 37-41 us each;
//   * XCD-aware work order, heaviest problem first (see the forward);
//   * 32-row blocks that hold only padding are skipped;
//   * transposed operands by asm ds_read_b64_tr_b16 with counted waits, the next fragment in flight under the
//     current MFMAs.
// ================================================================================================

// dQ^T += K^T . dS^T for block KT: PvStep with the K tile as the transposed operand.
// ACTIVE false: the wave only stages tiles and joins the barriers.  (The sweep split of the dK/dV kernel below was tried here
// too — waves 0 and 1 sharing the 32 query rows of a x30 problem — and taken out again: at this kernel's 168-register budget
// (three waves per SIMD) a third instantiation of the body spilled 31 registers instead of 6, a runtime selector 66, and the
// spills land in the main path's loop.)
template <int DH, bool DROP, bool ACTIVE>
__device__ __forceinline__ void dq2_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
                                         const int qs, char* smem) {
  constexpr int KS = DH / 16, DT = DH / 32, SB = (DH + 8) * 2, TILE_B = 64 * SB, STAGE_B = 2 * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;
  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Kg), 0, Tk * P.ldk * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Vg), 0, Tk * P.ldv * 2, 0x00020000);
  auto issue = [&](int j) { dma_pair<DH>(rsK, rsV, P.ldk, P.ldv, smem + (j & 1) * STAGE_B, j, wave, lane); };
  const int ntiles = (Tk + 63) / 64;
  issue(0);

  const int qrow = qs + (lane & 31);
  const size_t qoff = (size_t)b * Tq * P.ldq + h * DH, ooff = (size_t)b * Tq * P.ldo + h * DH;
  char* slice = smem + STAGE_B + wave * (32 * SB);
  bf16x8_t qf[KS], dof[KS];
  float delta = 0.f, lse2 = 0.f;
  if constexpr (ACTIVE) {
    load_row_frags_lds<DH>(qf, static_cast<const unsigned short*>(P.Q) + qoff, P.ldq, qs, Tq, lane, slice);
    load_row_frags_lds<DH>(dof, static_cast<const unsigned short*>(P.dO) + ooff, P.ldo, qs, Tq, lane, slice);
    bf16x8_t of[KS];
    load_row_frags_lds<DH>(of, static_cast<const unsigned short*>(P.O) + ooff, P.ldo, qs, Tq, lane, slice);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4_t x = __builtin_bit_cast(u32x4_t, of[ks]), y = __builtin_bit_cast(u32x4_t, dof[ks]);
#pragma unroll
      for (int e = 0; e < 4; ++e) delta += bf16lo(x[e]) * bf16lo(y[e]) + bf16hi(x[e]) * bf16hi(y[e]);
    }
    delta = half_sum(delta);
    if (half == 0 && qrow < Tq) P.delta[(size_t)bh * Tq + qrow] = delta;
    lse2 = (qrow < Tq ? P.LSE[(size_t)bh * Tq + qrow] : 0.f) * LOG2E;
  }
  const float c = a.scale * LOG2E;
  const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;
  const unsigned qidx = (unsigned)qrow * (unsigned)Tk;
  f32x16_t dq[1][DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[0][dt][r] = 0.f;
  const unsigned troff = (unsigned)((4 * half + ((lane >> 2) & 3)) * SB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  for (int j = 0; j < ntiles; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (j + 1 < ntiles) issue(j + 1);
    if constexpr (ACTIVE) {
      const char* sK = smem + (j & 1) * STAGE_B;
      const char* sV = sK + TILE_B;
      const unsigned vaK = smem_lds + (j & 1) * STAGE_B + troff;
      auto block = [&](auto KTc) {
        constexpr int KT = decltype(KTc)::value;
        const int k0 = j * 64 + 32 * KT;
        if (k0 >= Tk) return;
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mfma_prio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sK, 32 * KT, ks, lane), qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sV, 32 * KT, ks, lane), dof[ks], dp, 0, 0, 0);
        }
        mfma_prio(0);
        s16x4_t lo, hi;
        tr_issue<DH, 2 * KT, 0>(vaK, lo, hi);
        const bool ragged = k0 + 32 > Tk;
        f32x16_t ds[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = fast_exp2(__builtin_fmaf(s[r], c, -lse2));
          const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (ragged) p = key < Tk ? p : 0.f;
          float dpv = dp[r];                                // d(P_dropped) -> dP through the same mask
          if (DROP) dpv = mmf_keep(dkey, qidx + (unsigned)key, a.drop_thresh) ? dpv * a.inv_keep : 0.f;
          ds[0][r] = p * (dpv - delta);                     // dS^T (scale applied at the store)
        }
        bf16x8_t pf[1];
        PvStep<DH, 1, KT, 0>::run(vaK, lo, hi, ds, pf, dq);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    }
  }
  if constexpr (ACTIVE) {
    char* oslice = smem + (ntiles & 1) * STAGE_B + wave * (32 * SB);
    store_rows_lds<DH>(dq[0], a.scale, static_cast<unsigned short*>(P.dQ) + qoff, P.ldq, qs, Tq, lane, oslice);
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT, DQ_WAVES_PER_SIMD)
void attn_bwd_dq2_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B];
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi];
  const int bh = item / nchunk, q0 = (item % nchunk) * 128;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qs = q0 + 32 * wave, pidx = a.orig[pi];
  if (qs < P.Tq) dq2_wave<DH, DROP, true>(a, P, pidx, bh, qs, smem);
  else           dq2_wave<DH, DROP, false>(a, P, pidx, bh, qs, smem);
}

// dV^T += dO^T . P and dK^T += Q^T . dS for the 32 query rows of block QS: 4 DT fragment steps alternating the
// dO tile (va) and the Q tile (va - TILE_B... given separately), the next fragment in flight under this step's MFMA
template <int DH, int QS, int N>
struct DkvStep {
  static constexpr int DT = DH / 32, NF = 4 * DT;            // step N: ss = N / (2 DT), dt = (N / 2) % DT, operand N & 1
  static __device__ __forceinline__ void run(unsigned vaQ, unsigned vadO, s16x4_t lo, s16x4_t hi, const f32x16_t& pm,
                                             const f32x16_t& dsm, bf16x8_t& pf, bf16x8_t& dsf,
                                             f32x16_t (&dv)[DT], f32x16_t (&dk)[DT]) {
    s16x4_t nlo, nhi;
    if constexpr (N + 1 < NF) {
      constexpr int M = N + 1;
      tr_issue<DH, 2 * QS + M / (2 * DT), (M / 2) % DT>((M & 1) ? vaQ : vadO, nlo, nhi);
    }
    tr_wait<(N + 1 < NF) ? 2 : 0>(lo, hi);
    if constexpr (N == 0) mfma_prio(1);
    const bf16x8_t f = join(lo, hi);
    if constexpr (N % (2 * DT) == 0) { pf = acc_frag(pm, N / (2 * DT)); dsf = acc_frag(dsm, N / (2 * DT)); }
    constexpr int dt = (N / 2) % DT;
    if constexpr ((N & 1) == 0) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, pf, dv[dt], 0, 0, 0);
    else                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, dsf, dk[dt], 0, 0, 0);
    if constexpr (N + 1 < NF) DkvStep<DH, QS, N + 1>::run(vaQ, vadO, nlo, nhi, pm, dsm, pf, dsf, dv, dk);
    else mfma_prio(0);
  }
};

// Sweep split as in dq2_wave: a problem with <= 32 keys gives them to waves 0 and 1, wave w works on 32-row query
// block w of every tile; wave 1's partial dK^T, dV^T are added to wave 0's through LDS at the end.
template <int DH, bool DROP, bool ACTIVE>
__device__ __forceinline__ void dkv2_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
                                          const int k0, char* smem, const bool split_wg, const int sel) {
  constexpr int KS = DH / 16, DT = DH / 32, SB = (DH + 8) * 2, TILE_B = 64 * SB, STAGE_B = 2 * TILE_B;
  constexpr int STAT_OFF = 2 * STAGE_B;                      // [stage][lse 64 | delta 64] f32 behind the two stages
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;
  const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
  const unsigned short* dOg = static_cast<const unsigned short*>(P.dO) + (size_t)b * Tq * P.ldo + h * DH;
  const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Qg), 0, Tq * P.ldq * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsdO = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dOg), 0, Tq * P.ldo * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(P.LSE + (size_t)bh * Tq, 0, Tq * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(P.delta + (size_t)bh * Tq, 0, Tq * 4, 0x00020000);
  auto issue = [&](int j) {
    dma_pair<DH>(rsQ, rsdO, P.ldq, P.ldo, smem + (j & 1) * STAGE_B, j, wave, lane);
    // LSE and delta of the tile's 64 query rows: one 256-B piece each (4 B per lane; rows past Tq read as 0,
    // which is harmless: their Q and dO rows are zeros, so P = 1 meets dO = 0 and dS = 1 * (0 - 0))
    char* st = smem + STAT_OFF + (j & 1) * 512;
    if (wave == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (lds_void_t*)st, 4, (unsigned)(64 * j + lane) * 4u, 0, 0, 0);
    if (wave == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_t*)(st + 256), 4, (unsigned)(64 * j + lane) * 4u, 0, 0, 0);
  };
  const int ntiles = (Tq + 63) / 64;
  issue(0);

  const size_t koff = (size_t)b * Tk * P.ldk + h * DH, voff = (size_t)b * Tk * P.ldv + h * DH;
  char* slice = smem + STAGE_B + wave * (32 * SB);
  bf16x8_t kf[KS], vf[KS];
  if constexpr (ACTIVE) {
    load_row_frags_lds<DH>(kf, static_cast<const unsigned short*>(P.K) + koff, P.ldk, k0, Tk, lane, slice);
    load_row_frags_lds<DH>(vf, static_cast<const unsigned short*>(P.V) + voff, P.ldv, k0, Tk, lane, slice);
  }
  f32x16_t dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const float c = a.scale * LOG2E;
  const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;
  const unsigned kcol = (unsigned)(k0 + (lane & 31));
  const unsigned troff = (unsigned)((4 * half + ((lane >> 2) & 3)) * SB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  for (int j = 0; j < ntiles; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (j + 1 < ntiles) issue(j + 1);
    if constexpr (ACTIVE) {
      const char* sQ = smem + (j & 1) * STAGE_B;
      const char* sdO = sQ + TILE_B;
      const float* sl = reinterpret_cast<const float*>(smem + STAT_OFF + (j & 1) * 512);
      const unsigned vaQ = smem_lds + (j & 1) * STAGE_B + troff, vadO = vaQ + TILE_B;
      auto block = [&](auto QSc) {
        constexpr int QS = decltype(QSc)::value;
        const int q0 = j * 64 + 32 * QS;
        if (q0 >= Tq) return;
        if (sel >= 0 && QS != sel) return;                 // sweep split: this wave's half of the tile
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mfma_prio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sQ, 32 * QS, ks, lane), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sdO, 32 * QS, ks, lane), vf[ks], dp, 0, 0, 0);
        }
        mfma_prio(0);
        s16x4_t lo, hi;
        tr_issue<DH, 2 * QS, 0>(vadO, lo, hi);
        f32x16_t ds;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(sl + 32 * QS + 8 * g + 4 * half);
          const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(sl + 64 + 32 * QS + 8 * g + 4 * half);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float p = fast_exp2(__builtin_fmaf(s[4 * g + i], c, -l4[i] * LOG2E));
            float pd = p, dpv = dp[4 * g + i];
            if (DROP) {
              const unsigned q = (unsigned)(q0 + 8 * g + 4 * half + i);
              const bool keep = mmf_keep(dkey, q * (unsigned)Tk + kcol, a.drop_thresh);
              pd = keep ? p * a.inv_keep : 0.f;
              dpv = keep ? dpv * a.inv_keep : 0.f;
            }
            s[4 * g + i] = pd;                            // dV^T += dO^T . P_dropped
            ds[4 * g + i] = p * (dpv - d4[i]);
          }
        }
        bf16x8_t pf, dsf;
        DkvStep<DH, QS, 0>::run(vaQ, vadO, lo, hi, s, ds, pf, dsf, dv, dk);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    }
  }
  if (split_wg) {                                        // wave 1's partial sums -> wave 0 (the free stage holds them)
    float* red = reinterpret_cast<float*>(smem + (ntiles & 1) * STAGE_B);
    if (ACTIVE && sel == 1) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          red[(dt * 16 + r) * 64 + lane] = dk[dt][r];
          red[((DT + dt) * 16 + r) * 64 + lane] = dv[dt][r];
        }
    }
    __syncthreads();
    if (ACTIVE && sel == 0) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dk[dt][r] += red[(dt * 16 + r) * 64 + lane];
          dv[dt][r] += red[((DT + dt) * 16 + r) * 64 + lane];
        }
    }
    __syncthreads();
  }
  if constexpr (ACTIVE) {
    if (sel > 0) return;
    char* oslice = smem + (ntiles & 1) * STAGE_B + wave * (32 * SB);
    store_rows_lds<DH>(dk, a.scale, static_cast<unsigned short*>(P.dK) + koff, P.ldk, k0, Tk, lane, oslice);
    store_rows_lds<DH>(dv, 1.f, static_cast<unsigned short*>(P.dV) + voff, P.ldv, k0, Tk, lane, oslice);
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 2)      // dK^T, dV^T, K and V fragments alone are 144 registers: two waves per SIMD
void attn_bwd_dkv2_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B + 2 * 512];
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi];
  const int bh = item / nchunk, kc0 = (item % nchunk) * 128;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k0 = kc0 + 32 * wave, pidx = a.orig[pi];
  const bool split = a.split && P.Tk <= 32;                 // workgroup-uniform
  const bool act = split ? wave < 2 : k0 < P.Tk;
  if (act) dkv2_wave<DH, DROP, true>(a, P, pidx, bh, split ? kc0 : k0, smem, split, split ? wave : -1);
  else     dkv2_wave<DH, DROP, false>(a, P, pidx, bh, k0, smem, split, -1);
}

}  // namespace

int mmf_attn_bwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  if (int rc = check_ranges("mmf_attn_bwd_grouped", problems, n)) return rc;
  AttnArgs2 a;
  int total = fill_args2(a, problems, n, scale, drop_p, rng_state, site, 128, false, false);   // dQ (+ delta) first
  const bool dr = a.drop_thresh != 0u;
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_bwd_dq2_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dq2_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL((attn_bwd_dq2_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dq2_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
  MMF_CHECK_LAUNCH("mmf_attn_bwd_grouped(dq, v2)");
  total = fill_args2(a, problems, n, scale, drop_p, rng_state, site, 128, true, false);        // then dK/dV (reads delta)
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_bwd_dkv2_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dkv2_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL((attn_bwd_dkv2_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_bwd_dkv2_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
  MMF_CHECK_LAUNCH("mmf_attn_bwd_grouped(dkv, v2)");
  return MMF_OK;
}

// Called by mmf_attn_fwd_grouped_ex (attention.hip) after validation when the second generation is selected.
// idx (may be NULL): the caller's problem index of problems[i] — the dropout stream id the backward kernels use
int mmf_attn_fwd2_launch_indexed(const mmf_attn_problem* problems, const int* idx, int n, int head_dim, float scale,
                                 float drop_p, const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  if (int rc = check_ranges("mmf_attn_fwd_grouped", problems, n)) return rc;
  AttnArgs2 a;
  static const int rows = [] { const char* e = getenv("MMF_ATTN_FWD_ROWS"); return (e && atoi(e) == 256) ? 256 : 128; }();   // 128 (round 2): see attn_fwd2n_kernel
  const int total = fill_args2(a, problems, n, scale, drop_p, rng_state, site, rows, false, true);
  if (idx)
    for (int k = 0; k < n; ++k) a.orig[k] = (short)idx[a.orig[k]];
  const bool dr = a.drop_thresh != 0u;
  if (rows == 128) {
    if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd2n_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd2n_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
    else                { if (dr) hipLaunchKernelGGL((attn_fwd2n_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd2n_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
    MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped(v2n)");
    return MMF_OK;
  }
  static const int stages = [] { const char* e = getenv("MMF_ATTN_FWD_STAGES"); const int v = e ? atoi(e) : 2; return v == 3 ? 3 : 2; }();   // 3 measured equal (round 2): the per-tile chain, not the DMA, is what a workgroup waits for
  if (stages == 3) {
    if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd2_kernel<96, true, 3>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd2_kernel<96, false, 3>), dim3(total), dim3(NT), 0, s, a); }
    else                { if (dr) hipLaunchKernelGGL((attn_fwd2_kernel<64, true, 3>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd2_kernel<64, false, 3>), dim3(total), dim3(NT), 0, s, a); }
  } else {
    if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd2_kernel<96, true, 2>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd2_kernel<96, false, 2>), dim3(total), dim3(NT), 0, s, a); }
    else                { if (dr) hipLaunchKernelGGL((attn_fwd2_kernel<64, true, 2>), dim3(total), dim3(NT), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd2_kernel<64, false, 2>), dim3(total), dim3(NT), 0, s, a); }
  }
  MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped(v2)");
  return MMF_OK;
}

// Called by mmf_attn_fwd_grouped_ex (attention.hip) after validation when the second generation is selected.
int mmf_attn_fwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  return mmf_attn_fwd2_launch_indexed(problems, nullptr, n, head_dim, scale, drop_p, rng_state, site, s);
}
