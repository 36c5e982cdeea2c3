// Attention forward, fourth generation: an 8-wave workgroup whose two waves per SIMD ALTERNATE between a matrix phase
// and a softmax phase ("ping-pong", guide section "Two waves per SIMD").
//
// Why (profiles/r02_attention_occupancy_pmc.txt): the second generation is bound by instruction
// issue — with three independent 4-wave workgroups per CU the SIMD is saturated by the sum of its waves' streams, and
// MFMA and VALU run together in only 42 % of the MFMA-busy cycles, because nothing makes one wave's softmax coincide
// with another's MFMAs.  At head dimension 96 a 64-key tile is 24 MFMAs (768 cycles of the matrix pipe) and ~150 vector
// instructions of softmax (~700 issue cycles): two streams of equal length that can run side by side if they are kept
// in opposite phases.  Here they are, by construction:
//
//   workgroup  512 threads = 8 waves = up to 256 query rows of one (b, h); wave w owns rows q0 + 32 w; waves w and
//              w + 4 share a SIMD, waves 0-3 are group 0, waves 4-7 group 1;
//   phases     per 64-key tile j a wave runs   M(j): O^T += V^T(j-1).P^T(j-1)  then  S^T(j) = K(j).Q^T      (24 MFMAs)
//                                              V(j): tile maximum, p = exp2(s c - m), row sums, pack P(j)  (VALU)
//              (M(0) has no P.V part, M(ntiles) only that); the scores and the packed probabilities stay in the
//              wave's registers between its phases;
//   slots      time is cut into slots by ONE s_barrier each; in slot s group 0 runs phase s and group 1 phase s - 1,
//              so on every SIMD one wave is in a matrix phase while its partner is in a softmax phase;
//   staging    K/V tiles by LDS-DMA into a 4-stage ring (4 x 26 KiB): tile t + 4 is issued when slot 2 t + 4 opens (the
//              last reader of tile t is group 1's M(t + 1) in slot 2 t + 3) and awaited, with a counted vmcnt that
//              leaves the younger tile in flight, when slot 2 t + 7 closes;
//   epilogue   O / l through a wave-private LDS slice (whole-row stores) in a ring stage nobody reads any more.
// The running maximum is decided once per 64-key tile, before any probability of the tile is formed (both blocks share
// one m; deferred by 2^DEFER as in attention2.hip).  Layouts, XCD work order and the dropout stream are those of the
// second generation, which keeps the narrow problems (Tq <= 128 or Tk <= 64: nothing to alternate with).
#include "attn2_common.h"

namespace {

constexpr int NT4 = 512, NST4 = 4;

// MMF_ATTN_STAMPS (build-time, measurement only; tools/attn4_stamps.py): s_memtime cycles per wave and segment
#ifdef MMF_ATTN_STAMPS
__device__ unsigned long long* g_astamps = nullptr;       // [workgroup][8 waves][12 segments]
#define ASTAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); seg[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define ASTAMP(i) do {} while (0)
#endif

// piece p = wave + 8 i of the 2 PIECES one-KiB pieces of a stage (dma_pair with eight waves)
template <int DH>
__device__ __forceinline__ void dma_pair8(__amdgpu_buffer_rsrc_t rsX, __amdgpu_buffer_rsrc_t rsY, int ldx, int ldy,
                                          char* st, int j, int wave, int lane) {
  constexpr int SB = (DH + 8) * 2, TILE_B = 64 * SB, CPR = DH / 8 + 1, PIECES = TILE_B / 1024, NI = (2 * PIECES + 7) / 8;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = wave + 8 * i;
    if (p < 2 * PIECES) {
      const int isy = p >= PIECES, c = (p - isy * PIECES) * 64 + lane;
      const int row = c / CPR, ch = c % CPR;
      const int ld = isy ? ldy : ldx;
      const unsigned off = ch == CPR - 1 ? OOB : (unsigned)((64 * j + row) * ld * 2 + ch * 16);
      if (!isy) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(st + p * 1024), 16, off, 0, 0, 0);
      else      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void_t*)(st + p * 1024), 16, off, 0, 0, 0);
    }
  }
}

// O^T += V^T . P^T for the 32 keys of block KT of the tile at `va`, from PACKED probabilities pf[16-key group]; the
// transposed V^T reads run two fragments ahead (fragments N and N + 1 arrive in flight).
template <int DH, int KT, int N>
struct PvPacked {
  static constexpr int DT = DH / 32, NF = 2 * DT;
  static __device__ __forceinline__ void run(unsigned va, s16x4_t lo, s16x4_t hi, s16x4_t lo1, s16x4_t hi1,
                                             const bf16x8_t (&pf)[2], f32x16_t (&o)[DT]) {
    s16x4_t lo2, hi2;
    if constexpr (N + 2 < NF) tr_issue<DH, 2 * KT + (N + 2) / DT, (N + 2) % DT>(va, lo2, hi2);
    tr_wait<(N + 2 < NF) ? 4 : (N + 1 < NF) ? 2 : 0>(lo, hi);
    const bf16x8_t vf = join(lo, hi);
    o[N % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[N / DT], o[N % DT], 0, 0, 0);
    if constexpr (N + 1 < NF) PvPacked<DH, KT, N + 1>::run(va, lo1, hi1, lo2, hi2, pf, o);
  }
};

// s_waitcnt vmcnt(k) for a wave-uniform runtime k in 0..12 (the immediate has to be a constant)
__device__ __forceinline__ void wait_vm(int k) {
  switch (k) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // any other count: wait for everything (never wrong)
  }
}

template <int DH, bool DROP, bool ACTIVE>
__device__ __forceinline__ void fwd4_wave(const AttnArgs2& a, const mmf_attn_problem& P, const int pidx, const int bh,
                                          const int qs, char* smem) {
  constexpr int KS = DH / 16, DT = DH / 32, SB = (DH + 8) * 2, TILE_B = 64 * SB, STAGE_B = 2 * TILE_B;
  constexpr int PIECES2 = 2 * (TILE_B / 1024);
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int Tq = P.Tq, Tk = P.Tk, H = P.H;
  const int b = bh / H, h = bh % H;
  const int npw = (PIECES2 - wave + 7) / 8;                 // LDS-DMA pieces this wave issues per tile

  const unsigned short* Kg = static_cast<const unsigned short*>(P.K) + (size_t)b * Tk * P.ldk + h * DH;
  const unsigned short* Vg = static_cast<const unsigned short*>(P.V) + (size_t)b * Tk * P.ldv + h * DH;
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Kg), 0, Tk * P.ldk * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Vg), 0, Tk * P.ldv * 2, 0x00020000);
  // Per-lane offsets of this wave's (at most four) pieces inside a tile, computed once; a tile is then fetched through a
  // descriptor whose base is the tile's first row and whose range ends after row Tk - 1 (rows past Tk read as zeros), so
  // the loop issues its DMA with scalar arithmetic only.
  constexpr int CPR = DH / 8 + 1, PIECES = TILE_B / 1024, NI = (2 * PIECES + 7) / 8;
  unsigned voff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = wave + 8 * i, isy = p >= PIECES, c = (p - isy * PIECES) * 64 + lane;
    const int row = c / CPR, ch = c % CPR;
    voff[i] = ch == CPR - 1 ? OOB : (unsigned)(row * (isy ? P.ldv : P.ldk) * 2 + ch * 16);
  }
  (void)rsK; (void)rsV;
  auto issue = [&](int j) {
    char* st = smem + (j & (NST4 - 1)) * STAGE_B;
    const int rows_left = max(0, Tk - 64 * j);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Kg + (size_t)64 * j * P.ldk), 0, rows_left * P.ldk * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Vg + (size_t)64 * j * P.ldv), 0, rows_left * P.ldv * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = wave + 8 * i;
      if (p < 2 * PIECES) {
        if (p < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_t*)(st + p * 1024), 16, voff[i], 0, 0, 0);
        else            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void_t*)(st + p * 1024), 16, voff[i], 0, 0, 0);
      }
    }
  };
  const int n = (Tk + 63) / 64;

  // the first tiles of the ring, then the Q fragments straight from HBM; the empty asm makes the compiler place its wait
  // for them HERE (a wait it would otherwise put in front of the first MFMA, inside the slot loop, where it would drain
  // the prefetched tiles every time round)
#ifdef MMF_ATTN_STAMPS
  unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  const int npro = min(n, 2);                             // tiles 2, 3, ... follow when slots 0, 2, ... open
  for (int j = 0; j < npro; ++j) if (!(a.debug & 32)) issue(j);
  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
  if (ACTIVE && !(a.debug & 16)) {
    const unsigned short* Qg = static_cast<const unsigned short*>(P.Q) + (size_t)b * Tq * P.ldq + h * DH;
    load_row_frags<DH>(qf, Qg, P.ldq, qs, Tq, lane);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  }

  f32x16_t o[DT], s[2];
  bf16x8_t pf[2][2];
  float m = NEG_BIG, l = 0.f;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  const float c = a.scale * LOG2E;
  const unsigned dkey = DROP ? mmf_rng_key(*a.rng_state, a.site, (unsigned)(pidx * 4096 + bh)) : 0u;
  const unsigned troff = (unsigned)((4 * half + ((lane >> 2) & 3)) * SB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // matrix phase j: P.V of tile j - 1 (if any), then the raw scores of tile j (if any)
  auto mphase = [&](int j) {
    // the K fragments of tile j (both blocks) are requested first, so that they land under the P.V chain of tile j - 1
    bf16x8_t kf[2][KS];
    const bool qk0 = j < n, qk1 = j < n && j * 64 + 32 < Tk;
    if (qk0) {
      const char* sK = smem + (j & (NST4 - 1)) * STAGE_B;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[0][ks] = row_frag<DH>(sK, 0, ks, lane);
      if (qk1) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[1][ks] = row_frag<DH>(sK, 32, ks, lane);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (j > 0) {
      const unsigned va = smem_lds + ((j - 1) & (NST4 - 1)) * STAGE_B + TILE_B + troff;
      const int kb = (j - 1) * 64;
      // (six fragments of read-ahead over both blocks were tried: P.V 820-1,050 -> 950-1,220 cycles — more LDS reads in flight
      //  are slower here, as in the wgrad kernel)
      auto pv = [&](auto KTc) {
        constexpr int KT = decltype(KTc)::value;
        if (kb + 32 * KT >= Tk) return;
        s16x4_t lo, hi, lo1, hi1;
        tr_issue<DH, 2 * KT, 0>(va, lo, hi);
        tr_issue<DH, 2 * KT + 1 / DT, 1 % DT>(va, lo1, hi1);
        PvPacked<DH, KT, 0>::run(va, lo, hi, lo1, hi1, pf[KT], o);
      };
      pv(std::integral_constant<int, 0>{});
      pv(std::integral_constant<int, 1>{});
      ASTAMP(11);
    }
    if (qk0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[0][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][ks], qf[ks], s[0], 0, 0, 0);
      if (qk1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[1][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][ks], qf[ks], s[1], 0, 0, 0);
      }
    }
  };
  // softmax phase j: one running-maximum decision for the whole tile, then the probabilities of its (one or two) blocks
  auto vphase = [&](int j) {
    const int kb = j * 64;
    const bool two = kb + 32 < Tk;                           // wave-uniform: the second block holds real keys
    if (kb + 64 > Tk) {                                      // ragged last tile (the empty asm keeps this a branch: if-converted
      asm volatile("" ::: "memory");                         //  it is 16-32 selects in every softmax phase)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
        s[0][r] = key < Tk ? s[0][r] : NEG_BIG;
        if (two) s[1][r] = key + 32 < Tk ? s[1][r] : NEG_BIG;
      }
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
    if (two) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    }
    mx = half_max(mx) * c;
    if (!__all(mx <= m + DEFER)) {                           // wave-uniform: raise the running maximum
      const float mnew = fmaxf(m, mx);
      const float alpha = fast_exp2(m - mnew);
      m = mnew;
      l *= alpha;
      // in place, from asm: written as `o *= alpha` the compiler keeps a second copy of all O registers alive across this
      // rare branch and moves the whole accumulator set (24 v_mov_b64) in EVERY softmax phase
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(o[dt][r]) : "v"(alpha));
    }
    const float nm = -m;
    ASTAMP(8);
    auto probs = [&](auto KTc) {
      constexpr int KT = decltype(KTc)::value;
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(__builtin_fmaf(s[KT][r], c, nm));
        s[KT][r] = p;
        rs += p;
      }
      l += rs;
      if (DROP) {
        const unsigned qidx = (unsigned)(qs + (lane & 31)) * (unsigned)Tk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned key = (unsigned)(kb + 32 * KT + (r & 3) + 8 * (r >> 2) + 4 * half);
          s[KT][r] = mmf_keep(dkey, qidx + key, a.drop_thresh) ? s[KT][r] * a.inv_keep : 0.f;
        }
      }
      pf[KT][0] = acc_frag(s[KT], 0);
      pf[KT][1] = acc_frag(s[KT], 1);
    };
    probs(std::integral_constant<int, 0>{});
    ASTAMP(9);
    if (two) probs(std::integral_constant<int, 1>{});
    ASTAMP(10);
  };

  // prologue: tile 0 (everybody's pieces) has landed
  wait_vm(ACTIVE ? 0 : (npro - 1) * npw);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ASTAMP(0);

  // slot bookkeeping (identical for every wave of the workgroup: the barrier count must not depend on group or activity)
  auto slot_open = [&](int sl) {
    if (!(sl & 1)) {                                         // slot 2 t - 4 opens: the stage of tile t - 4 is free (t >= 2)
      const int t = (sl + 4) >> 1;
      if (t < n && !(a.debug & 1)) issue(t);
      ASTAMP(5);
    }
  };
  auto slot_close = [&](int sl) {
    if (sl >= 2 * n) return;                                 // the last two slots need no rendezvous (no DMA, no stage reuse)
    if (sl & 1) {                                            // tile t is needed from the next slot on
      const int t = (sl + 1) >> 1;
      if (t < n) {
        const int newest = min(t + 1, n - 1);                // youngest tile issued so far
        wait_vm((newest - t) * npw);
      }
      ASTAMP(3);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ASTAMP(4);
  };
  if constexpr (ACTIVE) {
    // group g runs M(j) in slot 2 j + g and V(j) in slot 2 j + 1 + g: straight-line per group, no phase switch at run time
    // (with one the compiler shuffles the accumulator sets between register copies at every merge)
    auto run = [&](auto Gc) {
      constexpr int G = decltype(Gc)::value;
      if (G == 1) { slot_open(0); slot_close(0); }
      for (int j = 0; j < n; ++j) {
        slot_open(2 * j + G);
        if (!(a.debug & 4)) mphase(j);
        ASTAMP(1);
        slot_close(2 * j + G);
        slot_open(2 * j + 1 + G);
        if (!(a.debug & 2)) vphase(j);
        ASTAMP(2);
        slot_close(2 * j + 1 + G);
      }
      if (!(a.debug & 4)) mphase(n);                         // slot 2 n + g: the last P.V
      ASTAMP(1);
    };
    if (grp == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
  } else {
    for (int sl = 0; sl < 2 * n; ++sl) { slot_open(sl); slot_close(sl); }
  }

  if (ACTIVE && !(a.debug & 8)) {
    // group 0 finishes one slot before group 1, whose last phase still reads V of tile n - 1 (stage (n - 1) & 3):
    // stages n & 3 (group 0) and (n + 1) & 3 (group 1) are read by nobody and no DMA is pending anywhere
    char* oslice = smem + ((n + grp) & (NST4 - 1)) * STAGE_B + (wave & 3) * (32 * SB);
    unsigned short* Og = static_cast<unsigned short*>(P.O) + (size_t)b * Tq * P.ldo + h * DH;
    const float lt = half_sum(l);
    store_rows_lds<DH>(o, 1.f / lt, Og, P.ldo, qs, Tq, lane, oslice);
    const int qrow = qs + (lane & 31);
    if (half == 0 && qrow < Tq) P.LSE[(size_t)bh * Tq + qrow] = m * LN2 + __logf(lt);
  }
#ifdef MMF_ATTN_STAMPS
  ASTAMP(6);
  seg[7] = __builtin_amdgcn_s_getreg(63492);               // HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13]
  if (g_astamps && lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) g_astamps[((size_t)blockIdx.x * 8 + wave) * 12 + i] = seg[i];
  }
#endif
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NT4, 2)
void attn_fwd4_kernel(const AttnArgs2 a) {
  constexpr int STAGE_B = 2 * 64 * (DH + 8) * 2;
  __shared__ __attribute__((aligned(1024))) char smem[NST4 * STAGE_B];
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < a.nprob && bid >= a.blk_start[pi + 1]) ++pi;
  const int loc = bid - a.blk_start[pi], n8 = (a.blk_start[pi + 1] - a.blk_start[pi]) >> 3;
  const int item = (loc & 7) * n8 + (loc >> 3);
  if (item >= a.nwg[pi]) return;
  const mmf_attn_problem& P = a.p[pi];
  const int nchunk = a.nchunk[pi], rpc = a.rpc[pi];
  const int bh = item / nchunk, q0 = (item % nchunk) * rpc;
  const int nb = (min(P.Tq, q0 + rpc) - q0 + 31) >> 5;       // 32-row query blocks in this chunk (1..8)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qs = q0 + 32 * wave, pidx = a.orig[pi];
  if (wave < nb) fwd4_wave<DH, DROP, true>(a, P, pidx, bh, qs, smem);
  else           fwd4_wave<DH, DROP, false>(a, P, pidx, bh, qs, smem);
}

}  // namespace

int mmf_attn_fwd2_launch_indexed(const mmf_attn_problem* problems, const int* idx, int n, int head_dim, float scale,
                                 float drop_p, const uint64_t* rng_state, uint32_t site, hipStream_t s);
#ifdef MMF_ATTN_STAMPS
extern "C" int mmf_debug_attn4_stamps(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_astamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

// Wide problems on the fourth generation, the rest on the second; `orig` keeps the caller's problem index so that the
// dropout streams match the backward kernels.
int mmf_attn_fwd4_launch(const mmf_attn_problem* problems, int n, int head_dim, float scale, float drop_p,
                         const uint64_t* rng_state, uint32_t site, hipStream_t s) {
  if (int rc = check_ranges("mmf_attn_fwd_grouped", problems, n)) return rc;
  mmf_attn_problem wide[MMF_ATTN_MAX_PROBLEMS], narrow[MMF_ATTN_MAX_PROBLEMS];
  int wide_idx[MMF_ATTN_MAX_PROBLEMS], narrow_idx[MMF_ATTN_MAX_PROBLEMS], nw = 0, nn = 0;
  static const int all4 = [] { const char* e = getenv("MMF_ATTN4_ALL"); return e ? atoi(e) : 0; }();   // A/B: narrow problems too
  for (int i = 0; i < n; ++i) {
    if (all4 || (problems[i].Tq > 128 && problems[i].Tk > 64)) { wide[nw] = problems[i]; wide_idx[nw++] = i; }
    else                                             { narrow[nn] = problems[i]; narrow_idx[nn++] = i; }
  }
  if (nw) {
    AttnArgs2 a;
    const int total = fill_args2(a, wide, nw, scale, drop_p, rng_state, site, 256, false, true);
    for (int k = 0; k < nw; ++k) a.orig[k] = (short)wide_idx[a.orig[k]];
    const bool dr = a.drop_thresh != 0u;
    if (head_dim == 96) { if (dr) hipLaunchKernelGGL((attn_fwd4_kernel<96, true>), dim3(total), dim3(NT4), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd4_kernel<96, false>), dim3(total), dim3(NT4), 0, s, a); }
    else                { if (dr) hipLaunchKernelGGL((attn_fwd4_kernel<64, true>), dim3(total), dim3(NT4), 0, s, a);
                          else    hipLaunchKernelGGL((attn_fwd4_kernel<64, false>), dim3(total), dim3(NT4), 0, s, a); }
    MMF_CHECK_LAUNCH("mmf_attn_fwd_grouped(v4)");
  }
  if (nn) return mmf_attn_fwd2_launch_indexed(narrow, narrow_idx, nn, head_dim, scale, drop_p, rng_state, site, s);
  return MMF_OK;
}
