// Grouped bf16 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), f32 accumulate: C-ABI entry points,
// validation and the choice of kernel generation.
//
// One launch = up to MMF_GEMM_MAX_PROBLEMS independent problems (the six cross-modal blocks'
// projections / FFNs of MulT, models/fusion_layers.py:146-153, are issued together so the grid
// has >> 256 workgroups even though each problem is small).  The problem table travels by value
// in the kernel arguments, so a launch is self-contained and hipGraph-capturable.
//
// Kernels (round 4: three are left): gemm2.hip (LDS-DMA ring, 256 x 128 tile, eight waves: the general kernel — any K, small launches),
// gemm6.hip (256 x 256 tile, one wave per SIMD with 128 x 128 wave tiles: wgrad, and NT / NN launches the persistent form has no flag set
// for) and gemm7.hip (gemm6's tile walked by persistent workgroups, outputs through LDS: the NT / NN launches of the fusion step).  Earlier
// generations — the register-staged 128 x 128 kernel and gemm3 (rounds 1-2), gemm4 (256 x 256 ring, eight waves; stream-K) and gemm5
// (32-deep ring, two workgroups per CU) (rounds 2-3) — are in git history with their measurements in DESIGN.md section 5; the last two
// left when a same-box A/B of the selection rule without them came out level or ahead on all four workloads (MulT 2.116 -> 2.088 ms,
// hierarchical 2.75 = 2.75, training step 3.33 -> 3.34, MELD-shaped 1.076 -> 1.070).
#include "mmf_internal.h"
#include <stdlib.h>

int mmf_gemm2_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm2.hip: LDS-DMA ring kernel
int mmf_gemm6_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm6.hip: one wave per SIMD, 128x128 wave tiles
bool mmf_gemm6_supports(const mmf_gemm_problem* problems, int num_problems, int layout);
bool mmf_gemm6_supports_epi(int epilogue, int out_f32);
int mmf_gemm7_launch(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                     int out_f32, const mmf_gemm_extra* extra, hipStream_t s);   // gemm7.hip: gemm6's tile, persistent workgroups
bool mmf_gemm7_supports(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue, int out_f32, const mmf_gemm_extra* extra);

// Implementation switch (A/B runs in one process: tools/gemm7_bench.py): 0 = automatic (default), 2 / 6 / 7 = that kernel for every
// launch it supports.  Default from MMF_GEMM_IMPL, else automatic.
static bool impl_built(int v) { return v == 0 || v == 2 || v == 6 || v == 7; }
static int g_gemm_impl = [] {
  const char* e = getenv("MMF_GEMM_IMPL");
  const int v = (e && e[0] >= '0' && e[0] <= '9') ? e[0] - '0' : 0;
  return impl_built(v) ? v : 0;
}();
// MMF_GEMM_PERSIST=0: NT / NN launches stay on gemm6 (one tile per workgroup) instead of its persistent form
static const int g_persist = [] { const char* e = getenv("MMF_GEMM_PERSIST"); return e ? atoi(e) : 1; }();

// The automatic choice: the 256 x 256 one-wave-per-SIMD tile for every launch it supports that gives at least a quarter of the CUs a
// tile (wgrad: half — a single layer's weight gradient keeps the 256 x 128 ring and its twice as many tiles), gemm2 for the rest.
static int auto_impl(const mmf_gemm_problem* p, int n, int layout) {
  long tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (long)((p[i].M + 255) / 256) * ((p[i].N + 255) / 256);
  static const int cus = [] { int c = mmf_device_cu_count(); return c > 0 ? c : 256; }();
  const long need = layout == MMF_GEMM_TN ? (cus + 1) / 2 : (cus + 3) / 4;
  return (tiles >= need && mmf_gemm6_supports(p, n, layout)) ? 6 : 2;
}
static int gemm_impl() { return g_gemm_impl; }
static thread_local int t_last_impl = 0;
extern "C" int mmf_gemm_last_impl(void) { return t_last_impl; }
extern "C" int mmf_gemm_select_impl(int impl) {
  if (!impl_built(impl))
    MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_select_impl: generation %d is not built (0 = automatic, 2, 6, 7; 1 / 3 left in round 3, 4 / 5 in round 4)", impl);
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
  const bool pinned = impl != 0;
  if (impl == 0) impl = auto_impl(problems, num_problems, layout);
  const bool ok6 = mmf_gemm6_supports(problems, num_problems, layout) && mmf_gemm6_supports_epi(epilogue, out_f32);
  if (impl == 6 && !pinned && g_persist && mmf_gemm7_supports(problems, num_problems, layout, epilogue, out_f32, extra)) impl = 7;
  if (impl == 7 && !mmf_gemm7_supports(problems, num_problems, layout, epilogue, out_f32, extra)) impl = 6;
  if (impl == 6 && !ok6) impl = 2;
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
  if (impl == 6) return mmf_gemm6_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  if (impl == 7) return mmf_gemm7_launch(problems, num_problems, layout, epilogue, out_f32, extra, s);
  MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gemm_grouped: kernel generation %d is not built (2, 6, 7)", impl);
}
