// Grouped fused attention (forward, dQ, dK/dV): fat workgroups, LDS-DMA ring, one dual-use LDS image.
//
// Structure (rounds 1-2; measurements behind each choice in DESIGN.md section 5):
//   * all attention problems of a stage (the six cross blocks, or the three self-attentions, of a MulT pass — unequal
//     Tq / Tk) go out as ONE launch; workgroup ids are remapped so that the chunks of one (b, h) (and neighbouring heads
//     of one batch row) run on the SAME XCD (blockIdx % 8) and share its L2;
//   * products are oriented so that every probability tile stays in registers between its two uses: forward / dQ compute
//     S^T = K.Q^T (query on the lane: row max / sum / LSE / delta are per-lane scalars, P^T feeds O^T = V^T.P^T and dS^T
//     feeds dQ^T = K^T.dS^T from the accumulator), dK/dV computes S = Q.K^T (key on the lane; a wave keeps dK^T, dV^T
//     of its 32 keys in accumulators while the workgroup sweeps the queries; no atomics, deterministic);
//   * swept tiles (K/V in forward and dQ; Q/dO/LSE/delta in dK/dV) go HBM/L2 -> LDS by LDS-DMA into a 2-stage ring, the
//     next tile in flight under the current tile's MFMAs; rows past T come back as zeros from the buffer range check;
//   * softmax on raw scores, p = exp2(fma(s, c, -m)); the running maximum is raised only when a row's block maximum
//     exceeds it by 2^DEFER (guide T13), 32-key blocks, blocks that hold only padding are skipped;
//   * transposed operands by asm ds_read_b64_tr_b16 with counted lgkmcnt waits.
// Round 3 (each from a measurement, profiles/r03_attn_*):
//   * ONE LDS image for row reads and transposed reads (attn_helpers.h: 8-row x 32-column subtiles, XOR swizzle) instead
//     of 16-byte padded rows: the transposed reads were 2-way bank-conflicted, and the LDS port is as busy as the matrix
//     pipe in these kernels (~1 KiB of fragment reads per MFMA);
//   * prologue loads (Q; Q, dO, O; K, V) issued together behind one wait instead of 6 / 18 / 12 serial round trips;
//   * per-lane LDS-DMA source offsets computed once per wave, tiles addressed through per-tile descriptors (SALU only).
#include "attn2_common.h"

namespace {

// MMF_ATTN_STAMPS (build-time, measurement only; tools/attn2_bwd_stamps.py): s_memtime cycles per wave and segment of the
// backward kernels.  0 prologue, 1 vmcnt wait, 2 barrier, 3 DMA issue, 4 S / dP products, 5 P / dS arithmetic,
// 6 dQ^T (dQ kernel) or dV^T / dK^T (dK/dV kernel) chain, 7 epilogue, 8 HW_ID.  Every stamp drains the wave's LDS reads
// (s_memtime returns through lgkmcnt), so the segments are attributions, not the undisturbed schedule.
#ifdef MMF_ATTN_STAMPS
__device__ unsigned long long* g_bstamps = nullptr;       // [kernel 0 dQ / 1 dKdV][workgroup][4 waves][12]
#define BSTAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); seg[i] += t_ - tlast; tlast = t_; } while (0)
#define BSTAMP_DECL unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tlast = __builtin_readcyclecounter()
#define BSTAMP_STORE(kern) do { seg[8] = __builtin_amdgcn_s_getreg(63492); \
    if (g_bstamps && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; ++i_) \
      g_bstamps[(((size_t)(kern) * 65536 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 12 + i_] = seg[i_]; } } while (0)
#else
#define BSTAMP(i) do {} while (0)
#define BSTAMP_DECL do {} while (0)
#define BSTAMP_STORE(kern) do {} while (0)
#endif

// One wave of the forward: its 32-row query block at rows qs (ACTIVE), or staging and barriers only.
template <int DH, bool DROP, bool ACTIVE>
__device__ __forceinline__ void fwd2_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
                                          const int qs, char* smem) {
  constexpr int KS = DH / 16, DT = DH / 32, TILE_B = img_tile_bytes<DH>(), STAGE_B = 2 * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;

  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;
  TileDma<DH> dma;
  dma.init(P.ldk, P.ldv, wave, lane);
  auto issue = [&](int j) {
    dma.issue(Kg + (size_t)64 * j * P.ldk, Vg + (size_t)64 * j * P.ldv, P.ldk, P.ldv, Tk - 64 * j, smem + (j & 1) * STAGE_B, wave);
  };
  const int ntiles = (Tk + 63) / 64;
  issue(0);

  // Q fragments through the wave's slice of stage 1 (free until the loop issues tile 1 behind its first barrier)
  char* slice = smem + STAGE_B + wave * img_slice_bytes<DH>();
  bf16x8_t qf[KS];
  if constexpr (ACTIVE) {
    const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Qg), 0, Tq * P.ldq * 2, 0x00020000);
    RowLoad<DH> rl;
    rl.issue(rsQ, P.ldq, qs, lane);
    rl.commit(qf, lane, slice);
  }

  f32x16_t o[DT];
  float m = NEG_BIG, l = 0.f;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  const float c = a.scale * LOG2E;
  const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;
  const unsigned tlo = tr_lane_lo(lane), thi = tr_lane_hi(lane);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  for (int j = 0; j < ntiles; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of tile j have landed
    __builtin_amdgcn_s_barrier();                          // ... everyone's; the stage of tile j-1 is free
    asm volatile("" ::: "memory");
    if (j + 1 < ntiles) issue(j + 1);
    if constexpr (ACTIVE) {
      const char* sK = smem + (j & 1) * STAGE_B;
      const TrBase vaV = tr_base(smem_lds + (j & 1) * STAGE_B + TILE_B, tlo, thi);
      const int kb = j * 64;
      // one 32-key block: S^T = K.Q^T (raw scores), online softmax (lane = query), O^T += V^T.P^T
      auto block = [&](auto KTc) {
        constexpr int KT = decltype(KTc)::value;
        const int k0 = kb + 32 * KT;
        if (k0 >= Tk) return;                              // wave-uniform: nothing but masked keys
        f32x16_t s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        mfma_prio(1);
        // all KS fragment reads in flight before the first MFMA: one exposed LDS latency per block instead of KS / 2
        bf16x8_t kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = row_frag<DH>(sK, 32 * KT, ks, lane);
        __builtin_amdgcn_sched_barrier(0);                 // (the scheduler otherwise re-pairs them with the MFMAs)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // the asm V^T reads below are invisible to the compiler's lgkmcnt count
        mfma_prio(0);
        constexpr int TD = 2;                              // the first two V^T fragments land under the softmax
        s16x4_t lo[2 * DT], hi[2 * DT];
        PvStepD<DH, KT, TD, 0>::prime(vaV, lo, hi);
        if (k0 + 32 > Tk) {                                // ragged block
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
        if (!__all(mx <= m + DEFER)) {                     // wave-uniform: raise the running maximum
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
        if (DROP) {          // nn.MultiheadAttention(dropout=p): drop/rescale the probabilities fed to P.V only
          const unsigned qidx = (unsigned)(qs + (lane & 31)) * (unsigned)Tk;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned key = (unsigned)(k0 + (r & 3) + 8 * (r >> 2) + 4 * half);
            s[r] = mmf_keep(dkey, qidx + key, a.drop_thresh) ? s[r] * a.inv_keep : 0.f;
          }
        }
        bf16x8_t pf;
        PvStepD<DH, KT, TD, 0>::run(vaV, lo, hi, s, pf, o);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    }
  }

  if constexpr (ACTIVE) {
    // the stage the last tile did not use is free (every wave passed the last barrier): reuse the wave's slice there
    char* oslice = smem + (ntiles & 1) * STAGE_B + wave * img_slice_bytes<DH>();
    unsigned short* Og = static_cast<unsigned short*>(P.O) + (size_t)b * Tq * P.ldo + h * DH;
    const float lt = half_sum(l);
    int lane_e;                                            // (the lane id read again behind the sweep: see dq2_wave)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    store_rows_lds<DH>(o, 1.f / lt, Og, P.ldo, qs, Tq, lane_e, oslice);
    const int qrow = qs + (lane & 31);
    if (half == 0 && qrow < Tq) P.LSE[(size_t)bh * Tq + qrow] = m * LN2 + __logf(lt);
  }
}

// One 32-row query block per wave (128 rows per workgroup), <= 168 registers, so THREE workgroups share a CU (3 x 48 KiB
// of LDS) and a wave's LDS / MFMA dependency waits have two other waves on its SIMD to hide under (round 2: against two
// blocks per wave at two workgroups per CU, 102.3 -> 91.7 us per step).
template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 3)
void attn_fwd2n_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * img_tile_bytes<DH>();
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B];
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  // XCD x (= blockIdx % 8; blk_start is a multiple of 8) walks a contiguous range of the problem's work
  // items, so the chunks of one (b, h) and the heads of one batch row share that XCD's L2
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi], rpc = a.rpc[pi];         // rpc <= 128 here
  const int bh = item / nchunk, q0 = (item % nchunk) * rpc;
  const int nb = (min(P.Tq, q0 + rpc) - q0 + 31) >> 5;      // 32-row query blocks in this chunk (1..4)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qs = q0 + 32 * wave, pidx = a.orig[pi];
  if (wave < nb) fwd2_wave<DH, DROP, true>(a, P, pidx, bh, qs, smem);
  else           fwd2_wave<DH, DROP, false>(a, P, pidx, bh, qs, smem);
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
  constexpr int KS = DH / 16, DT = DH / 32, TILE_B = img_tile_bytes<DH>(), STAGE_B = 2 * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;
  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;
  TileDma<DH> dma;
  dma.init(P.ldk, P.ldv, wave, lane);
  auto issue = [&](int j) {
    dma.issue(Kg + (size_t)64 * j * P.ldk, Vg + (size_t)64 * j * P.ldv, P.ldk, P.ldv, Tk - 64 * j, smem + (j & 1) * STAGE_B, wave);
  };
  const int ntiles = (Tk + 63) / 64;
  BSTAMP_DECL;
  issue(0);

  const int qrow = qs + (lane & 31);
  const size_t qoff = (size_t)b * Tq * P.ldq + h * DH, ooff = (size_t)b * Tq * P.ldo + h * DH;
  char* slice = smem + STAGE_B + wave * img_slice_bytes<DH>();
  bf16x8_t qf[KS], dof[KS];
  float delta = 0.f, lse2 = 0.f;
  if constexpr (ACTIVE) {
    auto rsrc = [&](const void* base, size_t off, int ld) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(static_cast<const unsigned short*>(base) + off), 0, Tq * ld * 2, 0x00020000);
    };
    RowLoad<DH> rq, rdo, ro;                                  // 3 x 6 loads in flight, one wait (RowLoad, attn_helpers.h)
    rq.issue(rsrc(P.Q, qoff, P.ldq), P.ldq, qs, lane);
    rdo.issue(rsrc(P.dO, ooff, P.ldo), P.ldo, qs, lane);
    ro.issue(rsrc(P.O, ooff, P.ldo), P.ldo, qs, lane);
    rq.commit(qf, lane, slice);
    rdo.commit(dof, lane, slice);
    bf16x8_t of[KS];
    ro.commit(of, lane, slice);
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
  const unsigned tlo = tr_lane_lo(lane), thi = tr_lane_hi(lane);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  if constexpr (ACTIVE) BSTAMP(0);
  for (int j = 0; j < ntiles; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (ACTIVE) BSTAMP(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (ACTIVE) BSTAMP(2);
    if (j + 1 < ntiles) issue(j + 1);
    if constexpr (ACTIVE) BSTAMP(3);
    if constexpr (ACTIVE) {
      const char* sK = smem + (j & 1) * STAGE_B;
      const char* sV = sK + TILE_B;
      const TrBase vaK = tr_base(smem_lds + (j & 1) * STAGE_B, tlo, thi);
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
        BSTAMP(4);
        constexpr int TD = MMF_DQ_TRDEPTH;
        s16x4_t lo[2 * DT], hi[2 * DT];
        PvStepD<DH, KT, TD, 0>::prime(vaK, lo, hi);         // the first K^T fragments land under the dS arithmetic
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
        BSTAMP(5);
        bf16x8_t pf;
        PvStepD<DH, KT, TD, 0>::run(vaK, lo, hi, ds[0], pf, dq[0]);
        BSTAMP(6);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    }
  }
  if constexpr (ACTIVE) {
    char* oslice = smem + (ntiles & 1) * STAGE_B + wave * img_slice_bytes<DH>();
    // the store's per-lane LDS / global offsets depend on the lane only: formed at kernel entry they are carried (spilled) around the
    // sweep, so the lane id is read again here (guide: "recompute per block (v_mbcnt)")
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    store_rows_lds<DH>(dq[0], a.scale, static_cast<unsigned short*>(P.dQ) + qoff, P.ldq, qs, Tq, lane_e, oslice);
    BSTAMP(7);
    BSTAMP_STORE(0);
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT, DQ_WAVES_PER_SIMD)
void attn_bwd_dq2_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * img_tile_bytes<DH>();
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

// Row statistics of the dK/dV kernel through the matrix pipe (round 3).  In S = Q.K^T the key is the lane and the query row
// the accumulator register, so LSE and delta of a 32-row block are needed as 16 different values per lane.  Rounds 1-2
// staged them in LDS and read them back with eight broadcast ds_read_b128 per block in the middle of the P / dS
// arithmetic: four or five exposed LDS round trips per block (the "arith" segment of the stamped dK/dV kernel was 3x the
// dQ kernel's, profiles/r03_attn_bwd_stamps.txt) and a quarter of the kernel's LDS cycles.  Instead each lane keeps the two
// statistics of ITS OWN row (lane & 31) of the block in a register (prefetched from global memory one tile ahead) and
// ONE MFMA per statistic spreads them into the accumulator layout as an outer product with a ones fragment:
//     D[q][key] = sum_k A[q][k] B[k][key],  A[q][0..2] = the exact three-way bf16 split of x[q],  B[0..2][key] = 1,
// exact in f32 (8 + 8 + 8 mantissa bits, f32 accumulate).  With D as the initial accumulator of the S (and, without
// dropout, the dP) chain the subtraction is free:  S' = Q.K^T - LSE / scale,  p = exp2(c S');  dP' = dO.V^T - delta,
// dS = p dP'.
// Sweep split as in dq2_wave: a problem with <= 32 keys gives them to waves 0 and 1, wave w works on 32-row query
// block w of every tile; wave 1's partial dK^T, dV^T are added to wave 0's through LDS at the end.
template <int DH, bool DROP, bool ACTIVE>
__device__ __forceinline__ void dkv2_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
                                          const int k0, char* smem, const bool split_wg, const int sel) {
  constexpr int KS = DH / 16, DT = DH / 32, TILE_B = img_tile_bytes<DH>(), STAGE_B = 2 * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;
  const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
  const unsigned short* dOg = static_cast<const unsigned short*>(P.dO) + (size_t)b * Tq * P.ldo + h * DH;
  const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(P.LSE + (size_t)bh * Tq, 0, Tq * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(P.delta + (size_t)bh * Tq, 0, Tq * 4, 0x00020000);
  TileDma<DH> dma;
  dma.init(P.ldq, P.ldo, wave, lane);
  auto issue = [&](int j) {
    dma.issue(Qg + (size_t)64 * j * P.ldq, dOg + (size_t)64 * j * P.ldo, P.ldq, P.ldo, Tq - 64 * j, smem + (j & 1) * STAGE_B, wave);
  };
  // LSE and delta of this lane's row (lane & 31) of the two 32-row blocks of tile j (rows past Tq read as 0, which is
  // harmless: their Q and dO rows are zeros, so P = 1 meets dO = 0 and dS = 1 * (0 - 0))
  float st_l[2], st_d[2];
  auto stats = [&](int j) {
#pragma unroll
    for (int qs_ = 0; qs_ < 2; ++qs_) {
      const unsigned off = (unsigned)(64 * j + 32 * qs_ + (lane & 31)) * 4u;
      st_l[qs_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, off, 0, 0));
      st_d[qs_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, off, 0, 0));
    }
  };
  const int ntiles = (Tq + 63) / 64;
  BSTAMP_DECL;
  issue(0);
  if constexpr (ACTIVE) stats(0);

  const size_t koff = (size_t)b * Tk * P.ldk + h * DH, voff = (size_t)b * Tk * P.ldv + h * DH;
  char* slice = smem + STAGE_B + wave * img_slice_bytes<DH>();
  bf16x8_t kf[KS], vf[KS];
  if constexpr (ACTIVE) {
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(static_cast<const unsigned short*>(P.K) + koff), 0, Tk * P.ldk * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(static_cast<const unsigned short*>(P.V) + voff), 0, Tk * P.ldv * 2, 0x00020000);
    RowLoad<DH> rk, rv;                                       // 2 x 6 loads in flight, one wait (RowLoad, attn_helpers.h)
    rk.issue(rsK, P.ldk, k0, lane);
    rv.issue(rsV, P.ldv, k0, lane);
    rk.commit(kf, lane, slice);
    rv.commit(vf, lane, slice);
  }
  f32x16_t dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const float c = a.scale * LOG2E;
  const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;
  const unsigned kcol = (unsigned)(k0 + (lane & 31));
  const unsigned tlo = tr_lane_lo(lane), thi = tr_lane_hi(lane);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  if constexpr (ACTIVE) BSTAMP(0);
  for (int j = 0; j < ntiles; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (ACTIVE) BSTAMP(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (ACTIVE) BSTAMP(2);
    if (j + 1 < ntiles) issue(j + 1);
    if constexpr (ACTIVE) BSTAMP(3);
    if constexpr (ACTIVE) {
      const char* sQ = smem + (j & 1) * STAGE_B;
      const char* sdO = sQ + TILE_B;
      // this tile's row statistics as MFMA operands; the next tile's are requested now and land under this tile's work
      const float nls = -1.f / a.scale;
      const float cl[2] = {st_l[0] * nls, st_l[1] * nls}, cd[2] = {-st_d[0], -st_d[1]};
      const bf16x8_t one3 = ones3_frag(half == 0);
      if (j + 1 < ntiles) stats(j + 1);
      const TrBase vaQ = tr_base(smem_lds + (j & 1) * STAGE_B, tlo, thi), vadO = tr_base(smem_lds + (j & 1) * STAGE_B + TILE_B, tlo, thi);
      auto block = [&](auto QSc) {
        constexpr int QS = decltype(QSc)::value;
        const int q0 = j * 64 + 32 * QS;
        if (q0 >= Tq) return;
        if (sel >= 0 && QS != sel) return;                 // sweep split: this wave's half of the tile
        f32x16_t s, dp, z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        mfma_prio(1);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(split3_frag(cl[QS], half == 0), one3, z, 0, 0, 0);   // S' starts at -LSE / scale
        f32x16_t dm;                                                                      // -delta[q] in every key column
        dm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(split3_frag(cd[QS], half == 0), one3, z, 0, 0, 0);
        if constexpr (DROP) dp = z; else dp = dm;                                         // dP' starts at -delta (no dropout)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sQ, 32 * QS, ks, lane), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(sdO, 32 * QS, ks, lane), vf[ks], dp, 0, 0, 0);
        }
        mfma_prio(0);
        BSTAMP(4);
        constexpr int TD = MMF_DKV_TRDEPTH;
        s16x4_t lo[4 * DT], hi[4 * DT];
        DkvStepD<DH, QS, TD, 0>::prime(vaQ, vadO, lo, hi);  // the first fragments land under the P / dS arithmetic
        f32x16_t ds;
        unsigned qbase = (unsigned)(q0 + 4 * half) * (unsigned)Tk + kcol;
        if constexpr (DROP) asm volatile("" : "+v"(qbase));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(s[r] * c);
          if constexpr (DROP) {
            // element index q * Tk + key as one per-block lane value + a scalar multiple of Tk: written as (q0 + c_r + 4 half) * Tk + key,
            // hipcc hoists the sixteen (c_r + 4 half) * Tk + key out of the sweep and spills them (40 registers, reloaded every block)
            const bool keep = mmf_keep(dkey, qbase + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)Tk, a.drop_thresh);
            s[r] = keep ? p * a.inv_keep : 0.f;                                       // dV^T += dO^T . P_dropped
            ds[r] = p * ((keep ? dp[r] * a.inv_keep : 0.f) + dm[r]);
          } else {
            s[r] = p;
            ds[r] = p * dp[r];
          }
        }
        BSTAMP(5);
        bf16x8_t pf, dsf;
        DkvStepD<DH, QS, TD, 0>::run(vaQ, vadO, lo, hi, s, ds, pf, dsf, dv, dk);
        BSTAMP(6);
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
    char* oslice = smem + (ntiles & 1) * STAGE_B + wave * img_slice_bytes<DH>();
    int lane_e;                                            // (the lane id read again behind the sweep: see dq2_wave)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    store_rows_lds<DH>(dk, a.scale, static_cast<unsigned short*>(P.dK) + koff, P.ldk, k0, Tk, lane_e, oslice);
    store_rows_lds<DH>(dv, 1.f, static_cast<unsigned short*>(P.dV) + voff, P.ldv, k0, Tk, lane_e, oslice);
    BSTAMP(7);
    BSTAMP_STORE(1);
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 2)      // dK^T, dV^T, K and V fragments alone are 144 registers: two waves per SIMD
void attn_bwd_dkv2_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * img_tile_bytes<DH>();
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B];
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

#ifdef MMF_ATTN_STAMPS
extern "C" int mmf_debug_attn2_stamps(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

// MMF_ATTN_STUBS (build-time, measurement only; tools/build_variant.sh): MMF_ATTN_STUB bits 1 / 2 / 4 skip the forward / dQ / dK/dV
// launches, to read off what the step's time owes to each of them (the outputs are garbage).
#ifdef MMF_ATTN_STUBS
static const int g_attn_stub = [] { const char* e = getenv("MMF_ATTN_STUB"); return e ? atoi(e) : 0; }();
#else
constexpr int g_attn_stub = 0;
#endif

namespace {
template <typename K96T, typename K96F, typename K64T, typename K64F>
void launch4(int head_dim, bool dr, K96T k96t, K96F k96f, K64T k64t, K64F k64f, int total, const AttnArgs2& a, hipStream_t s) {
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL(k96t, dim3(total), dim3(NT), 0, s, a); else hipLaunchKernelGGL(k96f, dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL(k64t, dim3(total), dim3(NT), 0, s, a); else hipLaunchKernelGGL(k64f, dim3(total), dim3(NT), 0, s, a); }
}
}  // namespace

int mmf_attn_bwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  if (int rc = check_ranges("mmf_attn_bwd_grouped", problems, n)) return rc;
  AttnArgs2 a;
  if (!(g_attn_stub & 2)) {
    const int total = fill_args2(a, problems, n, scale, drop_p, rng_state, site, 128, false, false);   // dQ (+ delta) first
    launch4(head_dim, a.drop_thresh != 0u, attn_bwd_dq2_kernel<96, true>, attn_bwd_dq2_kernel<96, false>, attn_bwd_dq2_kernel<64, true>,
            attn_bwd_dq2_kernel<64, false>, total, a, s);
    MMF_CHECK_LAUNCH("mmf_attn_bwd_grouped(dq, v2)");
  }
  if (g_attn_stub & 4) return MMF_OK;
  const int total = fill_args2(a, problems, n, scale, drop_p, rng_state, site, 128, true, false);        // then dK/dV (reads delta)
  launch4(head_dim, a.drop_thresh != 0u, attn_bwd_dkv2_kernel<96, true>, attn_bwd_dkv2_kernel<96, false>, attn_bwd_dkv2_kernel<64, true>,
          attn_bwd_dkv2_kernel<64, false>, total, a, s);
  MMF_CHECK_LAUNCH("mmf_attn_bwd_grouped(dkv, v2)");
  return MMF_OK;
}

// Called by mmf_attn_fwd_grouped_ex (attention.hip) after validation.
int mmf_attn_fwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  if (int rc = check_ranges("mmf_attn_fwd_grouped", problems, n)) return rc;
  if (g_attn_stub & 1) return MMF_OK;
  AttnArgs2 a;
  const int total = fill_args2(a, problems, n, scale, drop_p, rng_state, site, 128, false, true);
  launch4(head_dim, a.drop_thresh != 0u, attn_fwd2n_kernel<96, true>, attn_fwd2n_kernel<96, false>, attn_fwd2n_kernel<64, true>,
          attn_fwd2n_kernel<64, false>, total, a, s);
  MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped");
  return MMF_OK;
}
