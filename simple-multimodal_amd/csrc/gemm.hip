// Grouped bf16 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), f32 accumulate: C-ABI entry points,
// validation and the choice of kernel generation.
//
// One launch = up to MMF_GEMM_MAX_PROBLEMS independent problems (the six cross-modal blocks'
// projections / FFNs of MulT, models/fusion_layers.py:146-153, are issued together so the grid
// has >> 256 workgroups even though each problem is small).  The problem table travels by value
// in the kernel arguments, so a launch is self-contained and hipGraph-capturable.
//
// Kernels: gemm2.hip (LDS-DMA ring, 256 x 128 tile: long-K NT), gemm4.hip (256 x 256 tile), gemm5.hip (256 x 128, 32-deep k-step, two workgroups per CU),
// gemm6.hip (256 x 256 tile, one wave per SIMD with 128 x 128 wave tiles: wgrad).  Rounds 1-2 kept a first-generation 128 x 128
// register-staged kernel here and a persistent variant in gemm3.hip behind mmf_gemm_select_impl(1 / 3); round 3
// removed them (git history has them, DESIGN.md section 5 their measurements).
#include "mmf_internal.h"
#include <stdlib.h>

int mmf_gemm2_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm2.hip: LDS-DMA ring kernel

int mmf_gemm4_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm4.hip: 256x256 tile
int mmf_gemm5_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm5.hip: NT, 32-deep k-step, 2 workgroups / CU
int mmf_gemm6_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm6.hip: one wave per SIMD, 128x128 wave tiles
bool mmf_gemm6_supports(const mmf_gemm_problem* problems, int num_problems, int layout);
bool mmf_gemm6_supports_epi(int epilogue, int out_f32);
int mmf_gemm7_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm7.hip: gemm6's tile, persistent workgroups
bool mmf_gemm7_supports(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue, int out_f32, const mmf_gemm_extra* extra);

// Implementation switch (A/B runs in one process: tools/gemm_bench.py): 0 = automatic (default), 1 =
// register-staged 128x128 kernel of this file, 2 = 256x128 LDS-DMA ring (gemm2.hip), 3 = its persistent
// form (gemm3.hip), 4 = 256x256 LDS-DMA ring (gemm4.hip).  Default from MMF_GEMM_IMPL, else automatic.
static int g_gemm_impl = [] {
  const char* e = getenv("MMF_GEMM_IMPL");
  int v = (e && e[0] >= '0' && e[0] <= '7') ? e[0] - '0' : 0;
  if (v == 1 || v == 3) v = 0;
  return v;
}();

// Automatic choice between the two ring kernels.  The 256x256 tile does 25 % fewer LDS fragment reads,
// 33 % fewer LDS-DMA issues per MFMA and half the L2 traffic (measured 1.27 vs 1.02 PFLOP/s at 4096^3,
// +22-26 % on the FFN1 / dH groups) but quantises coarser: it is used for NT / NN launches whose
// 256x256 tiling fills at least 80 % of the CU-rounds it occupies.  wgrad (TN) stays on 256x128
// (the transposed-read operands of the bigger wave tile do not fit 256 registers without spills).
static const int g_shortk_mode = [] { const char* e = getenv("MMF_GEMM_SHORTK"); return e ? atoi(e) : 1; }();
static const int g_shortk_nn = [] { const char* e = getenv("MMF_GEMM_SHORTK_NN"); return e ? atoi(e) : 0; }();
// MMF_GEMM_POLICY: 1 = round 1's rule (below), 2 = round 2's rule from the per-group microbenchmarks
// (profiles/r02_gemm_generations.txt: every MulT launch group x {256x128, 256x256, 256x128/32-deep} in isolation).
static const int g_tn5 = [] { const char* e = getenv("MMF_GEMM_TN5"); return e ? atoi(e) : 0; }();   // wgrad on the 32-deep two-workgroups-per-CU kernel
static const int g_tn6 = [] { const char* e = getenv("MMF_GEMM_TN6"); return e ? atoi(e) : 1; }();    // wgrad on the one-wave-per-SIMD kernel (round 3 default; 0: the 256x128 ring)
static const int g_g6_maxtiles = [] { const char* e = getenv("MMF_GEMM6_MAXTILES"); return e ? atoi(e) : 1 << 30; }();
static const int g_longk6 = [] { const char* e = getenv("MMF_GEMM_LONGK6"); return e ? atoi(e) : 512; }();   // round 3 default: every K >= 512 launch
static const int g_policy = [] { const char* e = getenv("MMF_GEMM_POLICY"); return e ? atoi(e) : 2; }();
// round 4: every NT / NN launch that gemm6 would take goes to its persistent form (gemm7.hip) if that kernel has the launch's flag set
// (mmf_gemm7_supports); MMF_GEMM_PERSIST=0: off.  MMF_GEMM_PERSIST_MINTILES (default 0): only launches with more tiles than this —
// launches of at most one tile per CU gain from gemm7's drain (outputs and aux through LDS as whole rows) alone: +4 ... +29 % on the
// step's one-round groups (profiles/r04_gemm7_vs_gemm6.txt)
static int g_persist = [] { const char* e = getenv("MMF_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
static const int g_persist_min = [] { const char* e = getenv("MMF_GEMM_PERSIST_MINTILES"); return e ? atoi(e) : 0; }();
static int auto_impl(const mmf_gemm_problem* p, int n, int layout, bool allow6 = true) {
  if (layout == MMF_GEMM_TN) {
    if (g_tn5) return 5;
    long t6 = 0;
    for (int i = 0; i < n; ++i) t6 += (long)((p[i].M + 255) / 256) * ((p[i].N + 255) / 256);
    static const int cus_tn = [] { int c = mmf_device_cu_count(); return c > 0 ? c : 256; }();
    // a launch that would leave more than half of the CUs without a 256 x 256 tile (a single layer's weight gradient) keeps the
    // 256 x 128 ring: twice the tiles
    return (allow6 && g_tn6 && 2 * t6 >= cus_tn && mmf_gemm6_supports(p, n, layout)) ? 6 : 2;
  }
  long tiles = 0;
  int kmax = 0, kmin = 1 << 30;
  for (int i = 0; i < n; ++i) {
    tiles += (long)((p[i].M + 255) / 256) * ((p[i].N + 255) / 256);
    kmax = p[i].K > kmax ? p[i].K : kmax;
    kmin = p[i].K < kmin ? p[i].K : kmin;
  }
  // The one-wave-per-SIMD kernel (gemm6.hip) for every NT / NN launch whose shortest reduction is at least MMF_GEMM_LONGK6 (default
  // 512; 0 = never): with its flag-specialised epilogues it is level with or ahead of the ring kernels on every MulT launch group
  // (profiles/r03_gemm_generations.txt) and worth 2.20 -> 2.12 ms in the step; launches that would leave more than half of the CUs
  // without a 256 x 256 tile keep round 2's rule (smaller tiles)
  static const int cus6 = [] { int c = mmf_device_cu_count(); return c > 0 ? c : 256; }();
  if (allow6 && g_longk6 > 0 && kmin >= g_longk6 && tiles <= g_g6_maxtiles && 2 * tiles >= cus6 && mmf_gemm6_supports(p, n, layout)) return 6;
  static const int cus = [] { int c = mmf_device_cu_count(); return c > 0 ? c : 256; }();
  const long rounds = (tiles + cus - 1) / cus;
  if (g_policy >= 2) {
    // The 256x256 tile (1.2+ PF steady state) wins whenever its tiling either fills its CU rounds (>= 80 %) or is one
    // partial round of at least half the chip: a launch that leaves CUs idle still finishes sooner than two rounds of
    // the 256x128 tile (FFN2 / dX / out-projection groups of three MulT blocks: 177 tiles, +22 ... +32 %).  Otherwise:
    // short reductions go to the two-workgroups-per-CU kernel (NT and NN), long ones to the 256x128 ring (NT) or,
    // for NN, again to the 32-deep kernel (930 vs 897 TF on the dX groups).
    const bool fills = tiles * 100 >= rounds * cus * 80;
    const bool one_round = tiles <= cus && 2 * tiles >= cus;
    if (fills || one_round) return 4;
    if (kmax <= 1024) return 5;
    return layout == MMF_GEMM_NN ? 5 : 2;
  }
  if (tiles >= 2L * cus && tiles * 100 >= rounds * cus * 80) return 4;
  // NT launches with a short reduction that the 256x256 tiling does not fill: the 32-deep, two-workgroups-per-CU
  // form of the 256x128 kernel (one workgroup's pipeline fill / output burst under the other's MFMA loop):
  // +2 ... +9 % on the in-projection and out-projection launches of MulT (MMF_GEMM_SHORTK=0: off)
  if (layout == MMF_GEMM_NT && g_shortk_mode && kmax <= 1024) return 5;
  // NN with a short reduction (round 2, MMF_GEMM_SHORTK_NN=1 to enable): the same two-workgroups-per-CU form
  if (layout == MMF_GEMM_NN && g_shortk_nn && kmax <= 1024) return 5;
  return 2;
}
static int gemm_impl() { return g_gemm_impl; }
static thread_local int t_last_impl = 0;
extern "C" int mmf_gemm_last_impl(void) { return t_last_impl; }
extern "C" int mmf_gemm_select_impl(int impl) {
  if (impl < 0 || impl > 7) MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_select_impl: %d not in 0..7", impl);
  if (impl == 1 || impl == 3)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_select_impl: generation %d was removed in round 3 (built: 2, 4, 5, 6)", impl);
  g_gemm_impl = impl;
  return MMF_OK;
}

extern "C" int mmf_gemm_grouped(const mmf_gemm_problem* problems, int num_problems, int layout,
                                int epilogue, int out_f32, void* stream) {
  return mmf_gemm_grouped_ex(problems, num_problems, layout, epilogue, out_f32, nullptr, stream);
}

extern "C" int mmf_gemm_grouped_ex(const mmf_gemm_problem* problems, int num_problems, int layout,
                                   int epilogue, int out_f32, const mmf_gemm_extra* extra, void* stream) {
  if (!problems || num_problems <= 0 || num_problems > MMF_GEMM_MAX_PROBLEMS)
    MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped: num_problems=%d out of range [1,%d]", num_problems,
             MMF_GEMM_MAX_PROBLEMS);
  if (layout < MMF_GEMM_NT || layout > MMF_GEMM_TN)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: unknown layout %d", layout);
  if ((epilogue & MMF_EPI_ACCUM) && !out_f32)
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: MMF_EPI_ACCUM needs f32 output");
  if ((epilogue & MMF_EPI_MASK_AUX) && (epilogue & MMF_EPI_ADD_AUX))
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: MASK_AUX and ADD_AUX are exclusive");
  if ((epilogue & MMF_EPI_COLSUM_A) && (layout != MMF_GEMM_TN || (epilogue & MMF_EPI_BIAS)))
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: COLSUM_A is a TN (wgrad) epilogue and excludes BIAS");
  if (epilogue & MMF_EPI_DROPOUT) {
    if (!extra || !extra->rng_state || !(extra->dropout_p >= 0.f) || extra->dropout_p >= 1.f)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped_ex: MMF_EPI_DROPOUT needs rng_state and 0 <= p < 1");
  }
  int impl = gemm_impl();
  if (impl == 0) impl = auto_impl(problems, num_problems, layout);
  if (impl == 6 && !(mmf_gemm6_supports(problems, num_problems, layout) && mmf_gemm6_supports_epi(epilogue, out_f32)))
    impl = auto_impl(problems, num_problems, layout, false);
  if (impl == 6 && gemm_impl() == 0 && g_persist && mmf_gemm7_supports(problems, num_problems, layout, epilogue, out_f32, extra)) {
    long tiles = 0;
    for (int i = 0; i < num_problems; ++i) tiles += (long)((problems[i].M + 255) / 256) * ((problems[i].N + 255) / 256);
    if (tiles > g_persist_min) impl = 7;
  }
  if (impl == 7 && !mmf_gemm7_supports(problems, num_problems, layout, epilogue, out_f32, extra))
    impl = mmf_gemm6_supports(problems, num_problems, layout) && mmf_gemm6_supports_epi(epilogue, out_f32) ? 6 : auto_impl(problems, num_problems, layout, false);
  t_last_impl = impl;
  for (int i = 0; i < num_problems; ++i) {
    const mmf_gemm_problem& p = problems[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: empty problem M=%d N=%d K=%d", i, p.M, p.N, p.K);
    if (!p.A || !p.B || !p.C) MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: null operand", i);
    if ((epilogue & (MMF_EPI_BIAS | MMF_EPI_COLSUM_A)) && !p.bias) MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: bias is null", i);
    if ((epilogue & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) && !p.aux)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: aux is null", i);
    // the contiguous extent of each operand must be a multiple of 8 elements (16-B vector loads)
    const int a_cols = (layout == MMF_GEMM_TN) ? p.M : p.K;
    const int b_cols = (layout == MMF_GEMM_NT) ? p.K : p.N;
    if ((a_cols & 7) || (b_cols & 7) || (p.lda & 7) || (p.ldb & 7) || (p.N & 3) || (p.ldc & 3) ||
        ((epilogue & (MMF_EPI_MASK_AUX | MMF_EPI_ADD_AUX)) && (p.ldaux & 3)))
      MMF_FAIL(MMF_E_ALIGN, "mmf_gemm_grouped[%d]: M=%d N=%d K=%d lda=%d ldb=%d ldc=%d ldaux=%d violate "
               "the 8-element (operands) / 4-element (output) granularity", i, p.M, p.N, p.K, p.lda,
               p.ldb, p.ldc, p.ldaux);
    if (p.lda < a_cols || p.ldb < b_cols || p.ldc < p.N)
      MMF_FAIL(MMF_E_SHAPE, "mmf_gemm_grouped[%d]: leading dimension smaller than the row", i);
    if (!mmf_aligned16(p.A) || !mmf_aligned16(p.B) || !mmf_aligned16(p.C) ||
        (p.bias && !mmf_aligned16(p.bias)) || (p.aux && (reinterpret_cast<uintptr_t>(p.aux) & 7)))
      MMF_FAIL(MMF_E_ALIGN, "mmf_gemm_grouped[%d]: operand pointers must be 16-byte aligned", i);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (impl == 2) return mmf_gemm2_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  if (impl == 4) return mmf_gemm4_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  if (impl == 5) return mmf_gemm5_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  if (impl == 6) return mmf_gemm6_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  if (impl == 7) return mmf_gemm7_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: kernel generation %d is not built (2, 4, 5, 6, 7)", impl);
}
