// Attention forward, one wave per SIMD (round 4): 64 query rows per wave, 256 per workgroup, the whole 512-register file.
//
// Why (measurements in DESIGN.md section 5): in the 32-rows-per-wave kernels of attention2.hip every K / V^T fragment read from LDS
// feeds ONE MFMA (1 KiB of LDS reads per MFMA: the LDS port is as busy as the matrix pipe) and the softmax of a block can only
// hide under ANOTHER wave's MFMAs, whose phases nothing aligns (two co-resident waves of a SIMD take 2,400-2,800 cycles per 32-key
// block each, 384 of them MFMA).  Here a wave owns two 32-row query blocks:
//   * every K fragment and every V^T fragment is read once and used for both query blocks (512 B of LDS reads per MFMA);
//   * the wave's own instruction stream carries the overlap: while the matrix pipe runs S^T(b+1) = K(b+1).Q^T (12 MFMAs) the
//     vector ALU turns S^T(b) into probabilities; while it runs O^T += V^T(b).P^T(b) (12 MFMAs) the vector ALU forms the row sums
//     of block b and the row maxima of block b+1.  The interleave is pinned with sched_group_barrier (one MFMA, then its share of
//     the vector work), LDS reads are inline asm behind counted waits (as in gemm6.hip), fragments are requested one phase ahead;
//   * K / V tiles (64 keys) arrive by LDS-DMA into a 4-stage ring, three tiles ahead, ONE barrier per tile (in the middle of the
//     tile: it publishes tile j+1 and frees tile j-1's stage for tile j+3);
//   * a launch of the MulT shapes is at most one wave per SIMD anyway (128 (b, h) x 512 rows / 64 = 1024 waves), so what the
//     step waits for is one wave's walk through its sweep; a second co-resident wave does not shorten that, a denser stream does.
// Arithmetic as attention2.hip (raw scores, p = exp2(fma(s, c, -m)), running maximum raised only past 2^DEFER, 32-key blocks,
// O^T accumulated transposed, LSE out); reference: F.multi_head_attention_forward as called at models/fusion_layers.py:161-163,204.
#include "attn2_common.h"

namespace {

constexpr int NS3 = 4;                       // K/V ring stages

// MMF_ATTN_STAMPS (build-time, measurement only; tools/attn3_stamps.py): s_memtime cycles per wave and segment.
// 0 prologue (to the first barrier), 1 S^T(0), 2 rescale decisions, 3 phase 1, 4 middle of the tile (vmcnt, barrier, DMA issue),
// 5 phase 2 (with the wait for the V^T fragments), 6 epilogue, 7 blocks
#ifdef MMF_ATTN_STAMPS
__device__ unsigned long long* g_f3stamps = nullptr;       // [workgroup][4 waves][8]
#define FSTAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); seg[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x008, SG_VALU = 0x002;

template <int OFF>
__device__ __forceinline__ u32x4_t lds_b128(unsigned addr) {
  u32x4_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// lane parts of a row-fragment address in the image (attn_helpers.h row_frag): k-substep even / odd
__device__ __forceinline__ unsigned row_lane_even(int lane, int dh) {
  const int r = lane & 31, half = lane >> 5;
  return (unsigned)((dh / 32) * 512 * (r >> 3) + 64 * (r & 7) + 16 * (half ^ ((r >> 2) & 3)));
}
__device__ __forceinline__ unsigned row_lane_odd(int lane, int dh) {
  const int r = lane & 31, half = lane >> 5;
  return (unsigned)((dh / 32) * 512 * (r >> 3) + 64 * (r & 7) + 16 * ((2 + half) ^ ((r >> 2) & 3)));
}
template <int DH, int KT, int KS_> constexpr int row_imm() { return (DH / 32) * 512 * 4 * KT + 512 * (KS_ >> 1); }

template <int DH, int KT, int KS_ = 0>
__device__ __forceinline__ void kfrag_issue(unsigned even, unsigned odd, u32x4_t (&kf)[DH / 16]) {
  if constexpr (KS_ < DH / 16) {
    kf[KS_] = lds_b128<row_imm<DH, KT, KS_>()>((KS_ & 1) ? odd : even);
    kfrag_issue<DH, KT, KS_ + 1>(even, odd, kf);
  }
}
template <int N> struct Tie;
__device__ __forceinline__ void lgkm0_tie(u32x4_t (&f)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
}
__device__ __forceinline__ void lgkm0_tie(u32x4_t (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
}
__device__ __forceinline__ void lgkm0_tie(s16x4_t (&lo)[6], s16x4_t (&hi)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]),
               "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]));
}
__device__ __forceinline__ void lgkm0_tie(s16x4_t (&lo)[4], s16x4_t (&hi)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}
// all 2 * DT transposed fragments of key block KT of a tile: fragment n = (k-substep n / DT, d-tile n % DT)
template <int DH, int KT, int N = 0>
__device__ __forceinline__ void vt_issue(TrBase va, s16x4_t (&lo)[2 * (DH / 32)], s16x4_t (&hi)[2 * (DH / 32)]) {
  constexpr int DT = DH / 32;
  if constexpr (N < 2 * DT) {
    tr_issue<DH, 2 * KT + N / DT, N % DT>(va, lo[N], hi[N]);
    vt_issue<DH, KT, N + 1>(va, lo, hi);
  }
}

// One wave: NQ (1 or 2) 32-row query blocks at rows qs, qs + 32; NQ = 0: staging and barriers only.
template <int DH, bool DROP, int NQ>
__device__ __forceinline__ void fwd3_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh, const int qs,
                                          char* smem) {
  constexpr int KS = DH / 16, DT = DH / 32, TILE_B = img_tile_bytes<DH>(), STAGE_B = 2 * TILE_B, SLICE = img_slice_bytes<DH>();
  constexpr int NI = TileDma<DH>::NI, NQA = NQ > 0 ? NQ : 1;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;
  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;
  TileDma<DH> dma;
  dma.init(P.ldk, P.ldv, wave, lane);
  const int ntiles = (Tk + 63) >> 6, nblk = (Tk + 31) >> 5;
#ifdef MMF_ATTN_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  // tiles past the end go out with an empty range: no memory traffic, and the vmcnt arithmetic stays uniform
  auto issue = [&](int j) {
    const int left = max(Tk - 64 * j, 0);
    dma.issue(Kg + (size_t)64 * j * P.ldk, Vg + (size_t)64 * j * P.ldv, P.ldk, P.ldv, left, smem + (j & (NS3 - 1)) * STAGE_B, wave);
  };
#pragma unroll
  for (int j = 0; j < NS3 - 1; ++j) issue(j);

  char* qslice = smem + NS3 * STAGE_B + wave * 2 * SLICE;
  bf16x8_t qf[NQA][KS];
  if constexpr (NQ > 0) {
    const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Qg), 0, Tq * P.ldq * 2, 0x00020000);
    RowLoad<DH> rl[NQA];
#pragma unroll
    for (int q = 0; q < NQ; ++q) rl[q].issue(rsQ, P.ldq, qs + 32 * q, lane);
#pragma unroll
    for (int q = 0; q < NQ; ++q) rl[q].commit(qf[q], lane, qslice + q * SLICE);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // tiles 0 .. NS3-2 (this wave's pieces)
  __builtin_amdgcn_s_barrier();                               // ... everyone's
  asm volatile("" ::: "memory");
  FSTAMP(0);
  if constexpr (NQ == 0) {
    for (int j = 0; j + 1 < ntiles; ++j) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(j + NS3 - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  } else {
    f32x16_t o[NQA][DT];
    float m[NQA], l[NQA], mx[NQA];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      m[q] = NEG_BIG; l[q] = 0.f; mx[q] = NEG_BIG;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[q][dt][r] = 0.f;
    }
    const float c = a.scale * LOG2E;
    const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;
    const unsigned tlo = tr_lane_lo(lane), thi = tr_lane_hi(lane);
    const unsigned rev = row_lane_even(lane, DH), rod = row_lane_odd(lane, DH);
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // raw row maxima of a finished S^T block (log2 domain); the last block of a ragged sweep is masked first
    auto maxima = [&](f32x16_t (&s)[NQA], int blk) {
      if (32 * blk + 32 > Tk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * half;
            s[q][r] = key < Tk ? s[q][r] : NEG_BIG;
          }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float v = s[q][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) v = fmaxf(v, s[q][r]);
        mx[q] = half_max(v) * c;
      }
    };

    u32x4_t kfb[KS];
    f32x16_t sA[NQA], sB[NQA];
    // S^T(0)
    kfrag_issue<DH, 0>(smem_lds + rev, smem_lds + rod, kfb);
    lgkm0_tie(kfb);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[q][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sA[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kfb[ks]), qf[q][ks], sA[q], 0, 0, 0);
    }
    if (nblk > 1) { kfrag_issue<DH, 1>(smem_lds + rev, smem_lds + rod, kfb); lgkm0_tie(kfb); }
    maxima(sA, 0);
    FSTAMP(1);

    // block (j, KT): sc = its finished S^T, sn = the next block's (formed here when HAS_NEXT)
    auto body = [&](auto KTc, auto HNc, const int j, f32x16_t (&sc)[NQA], f32x16_t (&sn)[NQA]) {
      constexpr int KT = decltype(KTc)::value;
      constexpr bool HAS_NEXT = decltype(HNc)::value;
      const int blk = 2 * j + KT, k0 = 32 * blk;
      const unsigned stage = smem_lds + (j & (NS3 - 1)) * STAGE_B;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (!__all(mx[q] <= m[q] + DEFER)) {                 // wave-uniform: raise the running maximum of query block q
          const float mnew = fmaxf(m[q], mx[q]);
          const float alpha = fast_exp2(m[q] - mnew);
          m[q] = mnew;
          l[q] *= alpha;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[q][dt][r] *= alpha;
        }
      }
      FSTAMP(2);
      // ---- phase 1: matrix pipe S^T(next) = K(next).Q^T, vector ALU P(this) = exp2(S c - m); V^T(this) fragments requested
      __builtin_amdgcn_sched_barrier(0);
      s16x4_t lo[2 * DT], hi[2 * DT];
      vt_issue<DH, KT>(tr_base(stage + TILE_B, tlo, thi), lo, hi);
      if constexpr (HAS_NEXT) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) sn[q][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            sn[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kfb[ks]), qf[q][ks], sn[q], 0, 0, 0);
      }
      bf16x8_t pf[NQA][2];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float nm = -m[q];
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[q][r] = fast_exp2(__builtin_fmaf(sc[q][r], c, nm));
        if constexpr (DROP) {                                 // the row sums below use the undropped probabilities
          const unsigned qidx = (unsigned)(qs + 32 * q + (lane & 31)) * (unsigned)Tk;
          f32x16_t pd;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned key = (unsigned)(k0 + (r & 3) + 8 * (r >> 2) + 4 * half);
            pd[r] = mmf_keep(dkey, qidx + key, a.drop_thresh) ? sc[q][r] * a.inv_keep : 0.f;
          }
          pf[q][0] = acc_frag(pd, 0); pf[q][1] = acc_frag(pd, 1);
        } else {
          pf[q][0] = acc_frag(sc[q], 0); pf[q][1] = acc_frag(sc[q], 1);
        }
      }
      // the probabilities (and the next block's scores) are pinned HERE: left alone, the compiler sinks the exponentials into the
      // block of their first use, behind the barrier below, where nothing runs beside them
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        asm volatile("" : "+v"(pf[q][0]), "+v"(pf[q][1]));
        if constexpr (HAS_NEXT) asm volatile("" : "+v"(sn[q]));
      }
      if constexpr (HAS_NEXT && !DROP) {
#pragma unroll
        for (int i = 0; i < KS * NQ; ++i) { SGB(SG_MFMA, 1); SGB(SG_VALU, (40 * NQ + KS * NQ - 1) / (KS * NQ)); }
      }
      __builtin_amdgcn_sched_barrier(0);
      FSTAMP(3);
      // ---- middle of the tile: tile j+1 becomes readable, tile j-1's stage goes to tile j+3
      if constexpr (KT == 0) {
        if (j + 1 < ntiles) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          issue(j + NS3 - 1);
        }
      }
      FSTAMP(4);
      // ---- phase 2: matrix pipe O^T += V^T.P^T, vector ALU row sums of this block and row maxima of the next
      __builtin_amdgcn_sched_barrier(0);
      lgkm0_tie(lo, hi);
      if (blk + 2 < nblk) {                                   // K fragments of block blk + 2 (tile j+1: published above / at its own middle)
        const unsigned nstage = smem_lds + ((j + 1) & (NS3 - 1)) * STAGE_B;
        kfrag_issue<DH, KT>(nstage + rev, nstage + rod, kfb);
      }
#pragma unroll
      for (int n = 0; n < 2 * DT; ++n)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          o[q][n % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(lo[n], hi[n]), pf[q][n / DT], o[q][n % DT], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) rs += sc[q][r];
        l[q] += rs;
      }
      if constexpr (HAS_NEXT) {                               // (a ragged last block is masked and its maxima redone below, outside the pinned schedule)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float v = sn[q][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) v = fmaxf(v, sn[q][r]);
          mx[q] = half_max(v) * c;
        }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(l[q]), "+v"(mx[q]));
      if constexpr (!DROP) {
#pragma unroll
        for (int i = 0; i < 2 * DT * NQ; ++i) { SGB(SG_MFMA, 1); SGB(SG_VALU, 4); }
      }
      __builtin_amdgcn_sched_barrier(0);
      lgkm0_tie(kfb);
      if constexpr (HAS_NEXT) { if (32 * (blk + 1) + 32 > Tk) maxima(sn, blk + 1); }
      FSTAMP(5);
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
    using Yes = std::integral_constant<bool, true>; using No = std::integral_constant<bool, false>;
    // blocks 0 .. nblk-2 have a successor; the loop takes whole tiles, the tail the last one or two blocks
    int j = 0;
    for (; 2 * j + 2 < nblk; ++j) { body(T0{}, Yes{}, j, sA, sB); body(T1{}, Yes{}, j, sB, sA); }
    if (2 * j + 2 == nblk) { body(T0{}, Yes{}, j, sA, sB); body(T1{}, No{}, j, sB, sA); }
    else                   { body(T0{}, No{}, j, sA, sB); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the empty-range tail DMAs

    unsigned short* Og = static_cast<unsigned short*>(P.O) + (size_t)b * Tq * P.ldo + h * DH;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float lt = half_sum(l[q]);
      store_rows_lds<DH>(o[q], 1.f / lt, Og, P.ldo, qs + 32 * q, Tq, lane, qslice + q * SLICE);
      const int qrow = qs + 32 * q + (lane & 31);
      if (half == 0 && qrow < Tq) P.LSE[(size_t)bh * Tq + qrow] = m[q] * LN2 + __logf(lt);
    }
#ifdef MMF_ATTN_STAMPS
    FSTAMP(6);
    seg[7] = nblk;
    if (g_f3stamps && lane == 0) for (int i = 0; i < 8; ++i) g_f3stamps[((size_t)blockIdx.x * 4 + wave) * 8 + i] = seg[i];
#endif
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 1)
void attn_fwd3_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * img_tile_bytes<DH>(), SLICE = img_slice_bytes<DH>();
  __shared__ __attribute__((aligned(1024))) char smem[NS3 * STAGE_B + 8 * SLICE];
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi], rpc = a.rpc[pi];         // rpc <= 256
  const int bh = item / nchunk, q0 = (item % nchunk) * rpc;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qs = q0 + 64 * wave, pidx = a.orig[pi];
  const int rows = min(P.Tq, q0 + rpc) - qs;                // this wave's query rows (<= 64; <= 0: none)
  if (rows > 32)     fwd3_wave<DH, DROP, 2>(a, P, pidx, bh, qs, smem);
  else if (rows > 0) fwd3_wave<DH, DROP, 1>(a, P, pidx, bh, qs, smem);
  else               fwd3_wave<DH, DROP, 0>(a, P, pidx, bh, qs, smem);
}

}  // namespace

#ifdef MMF_ATTN_STAMPS
extern "C" int mmf_debug_attn3_stamps(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_f3stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

int mmf_attn_fwd3_launch(const mmf_attn_problem* problems, const int* which, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  AttnArgs2 a;
  const int total = fill_args2(a, problems, n, scale, drop_p, rng_state, site, 256, false, true, which);
  const bool dr = a.drop_thresh != 0u;
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd3_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_fwd3_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL((attn_fwd3_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_fwd3_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
  MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped(v3)");
  return MMF_OK;
}
