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
#include "mmf_internal.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "attn_helpers.h"

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr float DEFER = 6.0f;          // log2 domain: P <= 64 before a rescale is forced

struct AttnArgs2 {
  int nprob;
  float scale;
  unsigned drop_thresh, site;
  float inv_keep;
  const unsigned long long* rng_state;
  int debug;      // timing ablations (MMF_ATTN2_DEBUG, results wrong by design): 1 no K/V DMA after tile 0, 2 no compute
  int blk_start[MMF_ATTN_MAX_PROBLEMS + 1];   // multiples of 8 (XCD alignment)
  int nwg[MMF_ATTN_MAX_PROBLEMS];             // real workgroups of the problem = B*H*nchunk
  short nchunk[MMF_ATTN_MAX_PROBLEMS];        // query chunks per (b, h)
  short rpc[MMF_ATTN_MAX_PROBLEMS];           // rows per chunk (multiple of 32, <= 256)
  short orig[MMF_ATTN_MAX_PROBLEMS];          // caller's problem index (dropout stream id)
  mmf_attn_problem p[MMF_ATTN_MAX_PROBLEMS];
};

typedef __attribute__((address_space(3))) void lds_void_t;

template <int OFF>
__device__ __forceinline__ s16x4_t tr_read_imm(unsigned addr) {
  s16x4_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// One V^T fragment (32 head-dim columns D of the 16 keys [16 G, 16 G + 16) of the tile whose LDS byte address
// plus this lane's tr offset is `va`): two ds_read_b64_tr_b16.  Issued from inline asm so that hipcc's LDS-DMA
// alias bookkeeping does not drain the next tile's DMA (s_waitcnt vmcnt(0)) before them; tr_wait<N>() ties the
// destination registers to the counted lgkmcnt wait, so no use can be scheduled above it.
template <int DH, int G, int D>
__device__ __forceinline__ void tr_issue(unsigned va, s16x4_t& lo, s16x4_t& hi) {
  constexpr int SB = (DH + 8) * 2, OFF = 16 * G * SB + 64 * D;
  lo = tr_read_imm<OFF>(va);
  hi = tr_read_imm<OFF + 8 * SB>(va);
}
template <int PENDING>
__device__ __forceinline__ void tr_wait(s16x4_t& lo, s16x4_t& hi) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(PENDING));
}
__device__ __forceinline__ bf16x8_t join(const s16x4_t& lo, const s16x4_t& hi) {
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// O^T += V^T . P^T for the 32 keys of block KT: 2 DT fragment steps, fragment N + 1 in flight under step N's MFMAs
template <int DH, int NQ, int KT, int N>
struct PvStep {
  static constexpr int DT = DH / 32, NF = 2 * DT;
  static __device__ __forceinline__ void run(unsigned va, s16x4_t lo, s16x4_t hi, const f32x16_t (&s)[NQ],
                                             bf16x8_t (&pf)[NQ], f32x16_t (&o)[NQ][DT]) {
    s16x4_t nlo, nhi;
    if constexpr (N + 1 < NF) tr_issue<DH, 2 * KT + (N + 1) / DT, (N + 1) % DT>(va, nlo, nhi);
    tr_wait<(N + 1 < NF) ? 2 : 0>(lo, hi);
    const bf16x8_t vf = join(lo, hi);
    if constexpr (N % DT == 0) {
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) pf[qb] = acc_frag(s[qb], N / DT);
    }
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
      o[qb][N % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb], o[qb][N % DT], 0, 0, 0);
    if constexpr (N + 1 < NF) PvStep<DH, NQ, KT, N + 1>::run(va, nlo, nhi, s, pf, o);
  }
};

// One wave of the forward: NQ (0, 1 or 2) query blocks of 32 rows at rows qs and qs + 128.  Waves with
// NQ == 0 only take part in the K/V staging and the barriers.
template <int DH, bool DROP, int NQ>
__device__ __forceinline__ void fwd2_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
                                          const int qs, char* smem) {
  constexpr int KS = DH / 16, DT = DH / 32, SB = (DH + 8) * 2, TILE_B = 64 * SB, STAGE_B = 2 * TILE_B;
  constexpr int CPR = DH / 8 + 1, PIECES = TILE_B / 1024, NI = (2 * PIECES + 3) / 4;
  static_assert(TILE_B % 1024 == 0, "a K or V tile must be a whole number of 1-KiB LDS-DMA pieces");
  constexpr int NQA = NQ > 0 ? NQ : 1;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;

  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Kg), 0, Tk * P.ldk * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Vg), 0, Tk * P.ldv * 2, 0x00020000);

  // LDS-DMA source offsets of this lane: piece p = wave + 4 i covers image chunks 64 pc .. 64 pc + 63 of the
  // K (p < PIECES) or V tile; chunk c is row c / CPR, 16-B column c % CPR (the last column is the pad).
  const unsigned stepK = 64u * P.ldk * 2u, stepV = 64u * P.ldv * 2u;
  auto issue = [&](int j) {
    char* st = smem + (j & 1) * STAGE_B;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = wave + 4 * i;
      if (p < 2 * PIECES) {
        const int isv = p >= PIECES, c = (p - isv * PIECES) * 64 + lane;
        const int row = c / CPR, ch = c % CPR;
        const unsigned off = ch == CPR - 1 ? OOB : (unsigned)(row * (isv ? P.ldv : P.ldk) * 2 + ch * 16);
        if (!isv) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_void_t*)(st + p * 1024), 16, off + j * stepK, 0, 0, 0);
        else      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_void_t*)(st + p * 1024), 16, off + j * stepV, 0, 0, 0);
      }
    }
  };
  const int ntiles = (Tk + 63) / 64;
  issue(0);

  // Q fragments through the wave's slice of stage 1 (free until the first barrier of the loop)
  char* slice = smem + STAGE_B + wave * (32 * SB);
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

  for (int j = 0; j < ntiles; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of tile j have landed
    __builtin_amdgcn_s_barrier();                          // ... everyone's; the other stage is free
    asm volatile("" ::: "memory");
    if (j + 1 < ntiles && !(a.debug & 1)) issue(j + 1);
    if constexpr (NQ > 0) {
      const char* sK = smem + (j & 1) * STAGE_B;
      const unsigned va = smem_lds + (j & 1) * STAGE_B + TILE_B + troff;
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
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t kf = row_frag<DH>(sK, 32 * KT, ks, lane);
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb)
            s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);
        }
        s16x4_t lo, hi;
        tr_issue<DH, 2 * KT, 0>(va, lo, hi);               // first V^T fragment lands under the softmax
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
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
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
        PvStep<DH, NQ, KT, 0>::run(va, lo, hi, s, pf, o);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    }
  }

  if constexpr (NQ > 0) {
    // the stage the last tile did not use is free (every wave passed the last barrier): reuse the wave's slice there
    char* oslice = smem + (ntiles & 1) * STAGE_B + wave * (32 * SB);
    unsigned short* Og = static_cast<unsigned short*>(P.O) + (size_t)b * Tq * P.ldo + h * DH;
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      const float lt = l[qb] + __shfl_xor(l[qb], 32, 64);
      store_rows_lds<DH>(o[qb], 1.f / lt, Og, P.ldo, qs + 128 * qb, Tq, lane, oslice);
      const int qrow = qs + 128 * qb + (lane & 31);
      if (half == 0 && qrow < Tq) P.LSE[(size_t)bh * Tq + qrow] = m[qb] * LN2 + __logf(lt);
    }
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT, 2)
void attn_fwd2_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * 64 * (DH + 8) * 2;
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
  const int nchunk = a.nchunk[pi], rpc = a.rpc[pi];
  const int bh = item / nchunk, q0 = (item % nchunk) * rpc;
  const int nb = (min(P.Tq, q0 + rpc) - q0 + 31) >> 5;      // 32-row query blocks in this chunk (1..8)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nq = (a.debug & 2) ? 0 : (wave < nb) + (wave + 4 < nb);   // blocks wave and wave + 4
  const int qs = q0 + 32 * wave, pidx = a.orig[pi];
  if (nq == 2)      fwd2_wave<DH, DROP, 2>(a, P, pidx, bh, qs, smem);
  else if (nq == 1) fwd2_wave<DH, DROP, 1>(a, P, pidx, bh, qs, smem);
  else              fwd2_wave<DH, DROP, 0>(a, P, pidx, bh, qs, smem);
}

}  // namespace

// Called by mmf_attn_fwd_grouped_ex (attention.hip) after validation when the second generation is selected.
int mmf_attn_fwd2_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  AttnArgs2 a;
  a.nprob = n; a.scale = scale;
  a.drop_thresh = (drop_p > 0.f && rng_state) ? mmf_drop_thresh(drop_p) : 0u;
  a.inv_keep = a.drop_thresh ? 1.f / (1.f - (float)a.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  a.site = site;
  a.rng_state = reinterpret_cast<const unsigned long long*>(rng_state);
  const char* dbg = getenv("MMF_ATTN2_DEBUG");
  a.debug = dbg ? atoi(dbg) : 0;
  int order[MMF_ATTN_MAX_PROBLEMS];
  for (int i = 0; i < n; ++i) order[i] = i;
  // heaviest first: work per workgroup ~ rows per chunk x Tk
  std::stable_sort(order, order + n, [&](int x, int y) {
    return (long long)std::min(problems[x].Tq, 256) * problems[x].Tk > (long long)std::min(problems[y].Tq, 256) * problems[y].Tk;
  });
  int total = 0;
  for (int k = 0; k < n; ++k) {
    const mmf_attn_problem& q = problems[order[k]];
    if ((long long)q.Tk * q.ldk * 2 >= 0x7fffffffLL || (long long)q.Tk * q.ldv * 2 >= 0x7fffffffLL)
      MMF_FAIL(MMF_E_SHAPE, "mmf_attn_fwd_grouped: Tk*ld exceeds the 2 GiB buffer-descriptor range");
    const int nchunk = (q.Tq + 255) / 256;
    const int rpc = (((q.Tq + nchunk - 1) / nchunk) + 31) / 32 * 32;
    a.blk_start[k] = total;
    a.nwg[k] = q.B * q.H * nchunk;
    a.nchunk[k] = (short)nchunk; a.rpc[k] = (short)rpc; a.orig[k] = (short)order[k];
    a.p[k] = q;
    total += (a.nwg[k] + 7) / 8 * 8;
  }
  a.blk_start[n] = total;
  const bool dr = a.drop_thresh != 0u;
  if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd2_kernel<96, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_fwd2_kernel<96, false>), dim3(total), dim3(NT), 0, s, a); }
  else                { if (dr) hipLaunchKernelGGL((attn_fwd2_kernel<64, true>), dim3(total), dim3(NT), 0, s, a);
                        else    hipLaunchKernelGGL((attn_fwd2_kernel<64, false>), dim3(total), dim3(NT), 0, s, a); }
  MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped(v2)");
  return MMF_OK;
}
