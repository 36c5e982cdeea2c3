/* mmfusion.h — C ABI of libmmfusion.so: MI355X (gfx950) kernels for the cross-modal fusion path.
 *
 * The reference (nl1xx/simple-multimodal) is pure Python on stock torch.nn and has no FFI of its
 * own, so this ABI is what the reference-side binding for the fusion path
 * would call; each entry point names the reference arithmetic it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (activations, weights, outputs, saved statistics, workspaces); the library never
 *     allocates, frees or synchronises;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and is safe to capture into a hipGraph: descriptor tables travel by value
 *     in the kernel arguments;
 *   - activations / weights are bf16 (bit pattern of the upper half of an IEEE f32), row-major,
 *     leading dimensions in ELEMENTS; accumulation, softmax and LayerNorm statistics are f32;
 *   - return value: MMF_OK or a negative MMF_E_* code; mmf_last_error() gives the thread-local
 *     message.  No C++ exception crosses the boundary.
 *   - alignment: all base pointers 16-byte aligned; leading dimensions and K multiples of 8
 *     elements; N multiples of 4.  Violations return MMF_E_ALIGN / MMF_E_SHAPE, nothing is launched.
 */
#ifndef MMFUSION_H
#define MMFUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMF_ABI_VERSION 1

enum {
  MMF_OK = 0,
  MMF_E_SHAPE = -1,
  MMF_E_DTYPE = -2,
  MMF_E_ALIGN = -3,
  MMF_E_LAUNCH = -4,
  MMF_E_UNSUPPORTED = -5
};

int mmf_version(void);
const char* mmf_last_error(void);
/* number of compute units of the current device (grid sizing on the host side) */
int mmf_device_cu_count(void);

/* ------------------------------------------------------------------------------------------
 * Grouped GEMM, bf16 in / f32 accumulate on the MFMA units (v_mfma_f32_16x16x32_bf16).
 * Replaces every dense contraction of the path: nn.MultiheadAttention in/out projections
 * (models/fusion_layers.py:188-191,204), the FFN (:195-200,208), and all fusion MLPs
 * (:21-28,124-128,304-327,395-412,471-476), forward, dgrad and wgrad.
 *
 *   layout MMF_GEMM_NT:  C[m][n] = sum_k A[m][k]  * B[n][k]    (y = x W^T, forward)
 *   layout MMF_GEMM_NN:  C[m][n] = sum_k A[m][k]  * B[k][n]    (dx = dy W, dgrad)
 *   layout MMF_GEMM_TN:  C[m][n] = sum_k A[k][m]  * B[k][n]    (dW = dy^T x, wgrad)
 *
 * Epilogue, applied in this order on the f32 accumulator:
 *   + bias[n] (MMF_EPI_BIAS) -> relu (MMF_EPI_RELU) -> * (aux[m][n] > 0) (MMF_EPI_MASK_AUX)
 *   -> + aux[m][n] (MMF_EPI_ADD_AUX) -> + C_old[m][n] (MMF_EPI_ACCUM, f32 output only)
 * aux is bf16 with leading dimension ldaux.  Output is bf16 or f32 (out_f32 != 0).
 * One launch covers all problems (<= MMF_GEMM_MAX_PROBLEMS); problems must not alias outputs.
 * ------------------------------------------------------------------------------------------ */
#define MMF_GEMM_MAX_PROBLEMS 48
enum { MMF_GEMM_NT = 0, MMF_GEMM_NN = 1, MMF_GEMM_TN = 2 };
enum {
  MMF_EPI_BIAS = 1,
  MMF_EPI_RELU = 2,
  MMF_EPI_MASK_AUX = 4,
  MMF_EPI_ADD_AUX = 8,
  MMF_EPI_ACCUM = 16,
  /* TN only: additionally `bias[m] += sum_k A[k][m]` (f32 atomics) — the nn.Linear bias gradient
   * (column sums of dy) taken from the A tiles the wgrad kernel stages anyway; `bias` is then an
   * OUTPUT of length M and MMF_EPI_BIAS must not be set. */
  MMF_EPI_COLSUM_A = 32,
  /* mmf_gemm_grouped_ex only: after ReLU, zero each element with probability extra->dropout_p and scale
   * the kept ones by 1/(1-p) (nn.Dropout in training mode, reference models/fusion_layers.py:198);
   * the mask is a stateless hash of (*extra->rng_state, extra->site, problem index, m*N+n). */
  MMF_EPI_DROPOUT = 64
};

typedef struct mmf_gemm_problem {
  const void* A;      /* bf16 */
  const void* B;      /* bf16 */
  void* C;            /* bf16 or f32 */
  const float* bias;  /* f32 [N] or NULL (MMF_EPI_COLSUM_A: f32 [M], accumulated into) */
  const void* aux;    /* bf16 [M][ldaux] or NULL */
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldaux;
} mmf_gemm_problem;

int mmf_gemm_grouped(const mmf_gemm_problem* problems, int num_problems, int layout,
                     int epilogue, int out_f32, void* stream);
/* Extended form: `alpha` multiplies the result after the mask step (before +aux / accumulate) — the
 * 1/(1-p) of a dropout backward: dH = (dY W2) * (h_dropped > 0) * alpha needs no RNG because dropped
 * units are exactly 0 in the saved activations.  rng_state is a DEVICE pointer to the 64-bit dropout
 * state.  extra == NULL behaves like mmf_gemm_grouped. */
typedef struct mmf_gemm_extra {
  float alpha;
  float dropout_p;
  const uint64_t* rng_state;
  uint32_t site;
} mmf_gemm_extra;
int mmf_gemm_grouped_ex(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue,
                        int out_f32, const mmf_gemm_extra* extra, void* stream);

/* Tuning hook: which kernel mmf_gemm_grouped dispatches to (0 automatic [default]; 2 LDS-DMA ring 256x128; 4 LDS-DMA ring
 * 256x256; 5 256x128 with a 32-deep k-step and two workgroups per CU; 6 256x256 on one wave per SIMD with 128x128 wave tiles
 * [gemm6.hip: any K for TN, K % 32 == 0 for NT / NN, no aux epilogue with f32 output — otherwise the automatic choice]).
 * Automatic: 6 for TN (wgrad) and for NT / NN launches with K >= 512 that give at least half of the CUs a tile, else per launch
 * from the tile count and K (gemm.hip auto_impl); 7: see mmf_gemm_set_persistent_workgroups below.  1 and 3 (rounds 1-2: register-staged 128x128, persistent ring) were
 * removed in round 3 and are refused with MMF_E_SHAPE.  Results are identical up to f32 summation order; exists so that A/B
 * timings can be interleaved inside one process. */
int mmf_gemm_select_impl(int impl);
/* the kernel generation the calling thread's last mmf_gemm_grouped[_ex] call dispatched to (profiling labels) */
int mmf_gemm_last_impl(void);
/* Round 4, generation 7 (gemm7.hip): generation 6's tile on PERSISTENT workgroups — one per CU, each walking the tiles w, w + grid,
 * ... of the launch with its LDS ring running on across tile boundaries (no ring fill and no idle matrix pipe between tiles).  NT /
 * NN with bf16 output, K % 32 == 0 and K >= 160; the automatic choice takes it for every such launch generation 6 would get that
 * has more tiles than CUs (MMF_GEMM_PERSIST=0: off).  Results are bit-identical to generation 6.
 * mmf_gemm_set_persistent_workgroups(n): grid size of generation 7 (0 = the CU count [default]); a test / tuning hook: with a
 * small n a small problem exercises many tiles per workgroup. */
int mmf_gemm_set_persistent_workgroups(int n);

/* ------------------------------------------------------------------------------------------
 * Skinny-M linear layers (1 <= M <= 64 rows): the (B, d) MLPs of the Early / Contrastive / Adaptive /
 * Graph / meta branches (models/fusion_layers.py:21-28, 304-327, 395-412, 471-476) and MulT's pooled
 * projections.  Weight-streaming MFMA kernels with the weight tile on the MFMA row axis; the grouped GEMM
 * above would spend >90 % of a 256-row tile on padding.
 *   fwd:   Y[m][n] = act(sum_k X[m][k] W[n][k] + bias[n])           flags: MMF_EPI_BIAS | MMF_EPI_RELU
 *   dgrad: Y[m][k] = (sum_n X[m][n] W[n][k]) * (aux[m][k] > 0) * alpha   flags: MMF_EPI_MASK_AUX
 *          (X = dy [M][N], W [N][K], Y = dx [M][K]; N = W rows = reduction, K = W columns)
 * X, W, aux bf16; Y bf16 or f32; wgrad goes through mmf_gemm_grouped (TN).
 * ------------------------------------------------------------------------------------------ */
#define MMF_SKINNY_MAX_PROBLEMS 24
typedef struct mmf_skinny_problem {
  const void* X;      /* bf16 [M][ldx] */
  const void* W;      /* bf16 [N][ldw] */
  void* Y;            /* bf16 or f32 */
  const float* bias;  /* f32 [N] (fwd) or NULL */
  const void* aux;    /* bf16 [M][ldaux] (dgrad mask) or NULL */
  int32_t M, N, K;    /* W is N x K */
  int32_t ldx, ldw, ldy, ldaux;
} mmf_skinny_problem;
int mmf_skinny_linear_fwd(const mmf_skinny_problem* problems, int num_problems, int flags, int out_f32, void* stream);
int mmf_skinny_linear_dgrad(const mmf_skinny_problem* problems, int num_problems, int flags, float alpha, int out_f32,
                            void* stream);
/* Round 4: the launches AROUND a (B, d)-row linear folded into it — these rows cost a launch (~5 us) per kernel, not bandwidth
 * (models/fusion_layers.py:21-28,304-327,395-412,471-476: every Linear there is followed by ReLU and / or Dropout and fed by a cat
 * or a pooled f32 tensor).
 *   forward  X may be f32 (narrowed while loaded); MMF_EPI_DROPOUT: Y = dropout(act(X W^T + b)) with the build's counter-based
 *            masks (element m * N + n of stream `site`, problem i); Y2: a second copy of the output in the OTHER dtype (the bf16
 *            operand of the next linear beside the f32 value the module returns), or NULL; with an f32 X, `dz` (if not NULL)
 *            receives the narrowed input as bf16 [M][lddz] — the weight gradient's other operand.
 *   dgrad    dz = dY * (gate > 0) * gate_scale [* keep / (1 - p) with MMF_EPI_DROPOUT: the mask of the forward with the same site]
 *            is formed while dY (bf16 or f32) is loaded — gate = the forward's saved output (ReLU, or ReLU + dropout whose dropped
 *            units are exactly 0: gate_scale = 1 / (1 - p)) — and dx = dz W; `dz` (bf16, lddz) receives the gated gradient for the
 *            weight-gradient launch.  MMF_EPI_MASK_AUX / alpha act on dx as in mmf_skinny_linear_dgrad. */
typedef struct mmf_skinny_problem_ex {
  mmf_skinny_problem p;
  void* Y2;
  const void* gate;
  void* dz;
  int32_t ldy2, ldgate, lddz, reserved;
} mmf_skinny_problem_ex;
typedef struct mmf_skinny_extra {
  int32_t x_f32;        /* X (forward) / dY (dgrad) is f32 */
  int32_t gate_f32;     /* the gate tensor is f32 */
  float gate_scale;
  float dropout_p;
  const uint64_t* rng_state;
  uint32_t site;
  uint32_t reserved;
} mmf_skinny_extra;
int mmf_skinny_linear_fwd_ex(const mmf_skinny_problem_ex* problems, int num_problems, int flags, int out_f32,
                             const mmf_skinny_extra* extra, void* stream);
int mmf_skinny_linear_dgrad_ex(const mmf_skinny_problem_ex* problems, int num_problems, int flags, float alpha, int out_f32,
                               const mmf_skinny_extra* extra, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-step loss and ModalityDropout as one launch each (round 4; csrc/loss.hip).
 * mmf_fusion_loss: replaces the loss of training/advanced_trainer.py:139-166 —
 *   loss = mean_b CE_ls(logits[b], targets[b]) + sum_j extra_w[j] * extra[j][0]
 * with CE_ls = (1 - eps) * (-log p_y) + eps * (-(1 / C) sum_c log p_c) (torch's label_smoothing), C <= 64 — and writes
 * dlogits[b][c] = d loss / d logits (f32 [B][C], may be NULL).  extra[j]: device scalars (the contrastive / distillation losses).
 * mmf_modality_dropout: replaces models/encoders.py:289-321 — y_m[b] = x_m[b] * keep[b][m] for the three (B, d) f32 feature
 * tensors; draw != 0: keep[b][m] ~ Bernoulli(1 - p) from the counter-based hash of (*rng_state, site, 3 b + m), a sample with no
 * modality left gets one back, the masks are written to `keep` (f32 [B][3]); draw == 0: `keep` is applied as given (backward).
 * ------------------------------------------------------------------------------------------ */
int mmf_fusion_loss(const float* logits, int ldl, const int64_t* targets, int B, int C, float label_smoothing,
                    const float* const* extra, const float* extra_w, int n_extra, float* loss, float* dlogits, void* stream);
int mmf_modality_dropout(const float* const* x, float* const* y, float* keep, int B, int d, float p,
                         const uint64_t* rng_state, uint32_t site, int draw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped fused attention (flash-style: no (Tq,Tk) score matrix in HBM).
 * Replaces q*scale, QK^T, softmax, P.V of F.multi_head_attention_forward as called at
 * models/fusion_layers.py:161-163,204 (six cross blocks + three self blocks of MulT in ONE
 * launch) and models/encoders.py:152-154,236-238.
 * Row r = b*T + t of Q/K/V/O holds head h at columns [h*head_dim, (h+1)*head_dim).
 * head_dim in {64, 96}.  LSE (f32, [B][H][Tq]) = log sum_k exp(scale * q.k) is saved for backward.
 * Backward recomputes the probabilities; `delta` is a caller-provided f32 [B][H][Tq] workspace.
 * ------------------------------------------------------------------------------------------ */
#define MMF_ATTN_MAX_PROBLEMS 12
typedef struct mmf_attn_problem {
  const void* Q; const void* K; const void* V;   /* bf16 */
  void* O;                                       /* bf16 */
  float* LSE;                                    /* f32 [B*H*Tq] */
  /* backward only (NULL in forward) */
  const void* dO;                                /* bf16, same layout as O */
  float* delta;                                  /* f32 [B*H*Tq] workspace */
  void* dQ; void* dK; void* dV;                  /* bf16, same layouts/strides as Q, K, V */
  int32_t B, H, Tq, Tk;
  int32_t ldq, ldk, ldv, ldo;                    /* row strides (elements) of Q,K,V,O (and grads) */
} mmf_attn_problem;

int mmf_attn_fwd_grouped(const mmf_attn_problem* problems, int num_problems, int head_dim,
                         float scale, void* stream);
int mmf_attn_bwd_grouped(const mmf_attn_problem* problems, int num_problems, int head_dim,
                         float scale, void* stream);
/* With attention-probability dropout (nn.MultiheadAttention(dropout=p) in training mode): the
 * probabilities are dropped/rescaled before P.V, the softmax normaliser uses the undropped row sums;
 * the backward kernels regenerate the same mask from (*rng_state, site, problem, b, h, q, key). */
int mmf_attn_fwd_grouped_ex(const mmf_attn_problem* problems, int num_problems, int head_dim, float scale,
                            float dropout_p, const uint64_t* rng_state, uint32_t site, void* stream);
/* Tuning hook kept for ABI stability: 0 (automatic) and 2 select attention2.hip, the only implementation since round 3
 * (LDS-DMA ring, 128 query rows per workgroup, XCD-aware order); any other value is refused with MMF_E_SHAPE. */
int mmf_attn_select_impl(int impl);
int mmf_attn_bwd_grouped_ex(const mmf_attn_problem* problems, int num_problems, int head_dim, float scale,
                            float dropout_p, const uint64_t* rng_state, uint32_t site, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (eps inside the sqrt, biased variance, affine), one
 * wavefront per row, f32 statistics by wave reduction.  Replaces nn.LayerNorm at
 * models/fusion_layers.py:205,209 (the residual add is fused into the producing GEMM's epilogue).
 * Forward saves mean and rstd (f32 [rows]).  Backward writes dx (bf16) and ADDS the
 * per-column sums into dgamma/dbeta (f32 [d], caller zeroes or accumulates; problems of one call
 * must not share dgamma/dbeta).
 * ------------------------------------------------------------------------------------------ */
#define MMF_LN_MAX_PROBLEMS 8
typedef struct mmf_ln_problem {
  const void* x;        /* bf16 [rows][d] */
  void* y;              /* fwd: bf16 out.   bwd: unused */
  const float* gamma;   /* f32 [d] */
  const float* beta;    /* f32 [d] (fwd) */
  float* mean;          /* f32 [rows]: written by fwd, read by bwd */
  float* rstd;
  const void* dy;       /* bwd: bf16 [rows][d] */
  void* dx;             /* bwd: bf16 [rows][d] */
  float* dgamma;        /* bwd: f32 [d], accumulated into (dgamma += column sums) */
  float* dbeta;
  int32_t rows;
} mmf_ln_problem;

int mmf_layernorm_fwd_grouped(const mmf_ln_problem* problems, int num_problems, int d, float eps,
                              void* stream);
/* backward needs an f32 workspace of mmf_layernorm_bwd_workspace_bytes(d) bytes (per-workgroup
 * column partial sums; a second small kernel adds them into dgamma/dbeta — no same-address atomics). */
size_t mmf_layernorm_bwd_workspace_bytes(int d);
int mmf_layernorm_bwd_grouped(const mmf_ln_problem* problems, int num_problems, int d,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Streaming helpers (HBM-bound, 16-byte vector accesses)
 * ------------------------------------------------------------------------------------------ */
/* dst_bf16[i] = bf16(src_f32[i]); n multiple of 8 not required */
int mmf_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int mmf_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* row-strided source (rows x cols, ld_src floats between rows) -> contiguous bf16; cols, ld_src multiples of 4 */
int mmf_cast_f32_to_bf16_2d(const float* src, void* dst, int rows, int cols, int ld_src, void* stream);
/* dst_f32[i] = scale * f32(src_bf16[i]): the receive side of the bf16-compressed gradient all-reduce
 * (mmfusion/dp.py: widen the summed wire buffer and divide by the world size in one pass) */
int mmf_cast_bf16_to_f32_scaled(const void* src, float* dst, int64_t n, float scale, void* stream);
/* y = a + b + c (bf16); models/fusion_layers.py:156-158.  c may be NULL (y = a + b). */
int mmf_add3_bf16(const void* a, const void* b, const void* c, void* y, int64_t n, void* stream);
/* several three-operand sums in one launch (y_i = a_i + b_i + c_i, bf16, n_i elements each) */
#define MMF_ADD3_MAX 8
typedef struct mmf_add3_problem {
  const void* a; const void* b; const void* c;
  void* y;
  int64_t n;
} mmf_add3_problem;
int mmf_add3_grouped(const mmf_add3_problem* problems, int num_problems, void* stream);
/* y = x_0 + ... + x_{n-1} (2 <= n <= MMF_ADDN_MAX bf16 tensors of numel elements, f32 accumulate, bf16 or f32 out) in
 * one pass: the gradient of a tensor the forward used n times (MulT's input rows: models/fusion_layers.py:146-158). */
#define MMF_ADDN_MAX 8
int mmf_addn_bf16(const void* const* xs, int n, void* y, int64_t numel, int out_f32, void* stream);
/* several such sums in one launch (bf16 out): the three modalities' input-gradient sums at the end of MulT's backward were
 * three launches of 6 - 18 us with graph-node gaps between them, alone on the chip in front of the deferred wgrad launch
 * (round 3, profiles/r03_step_timeline.txt) */
#define MMF_ADDN_GROUP_MAX 4
typedef struct mmf_addn_problem {
  const void* x[MMF_ADDN_MAX];
  void* y;
  int64_t numel;
  int32_t n;
} mmf_addn_problem;
int mmf_addn_grouped(const mmf_addn_problem* problems, int num_problems, void* stream);
/* y[b][j] = mean_t x[b][t][j]  (models/fusion_layers.py:166-168); x bf16 [B][T][d], y bf16 with
 * row stride ldy (lets the three pooled modalities land side by side = torch.cat, :171). */
int mmf_meanpool_fwd(const void* x, void* y, int B, int T, int d, int ldy, void* stream);
/* dx[b][t][j] = dy[b][j] / T ; dy bf16 with row stride lddy */
int mmf_meanpool_bwd(const void* dy, void* dx, int B, int T, int d, int lddy, void* stream);
/* the n (<= MMF_POOL_MAX) modalities of :166-171 in one launch: problem i pools xs[i] (B, Ts[i], d) into columns
 * [i d, (i+1) d) of y; backward scatters dy's column blocks back (dxs[i] bf16 (B, Ts[i], d)).  xs / dxs / Ts are
 * HOST arrays. */
#define MMF_POOL_MAX 4
int mmf_meanpool_cat_fwd(const void* const* xs, const int* Ts, int n, void* y, int B, int d, int ldy, void* stream);
int mmf_meanpool_cat_bwd(const void* dy, void* const* dxs, const int* Ts, int n, int B, int d, int lddy, void* stream);
/* out[n] (+)= sum_m x[m][n]; x bf16 [M][ldx]; out f32, atomically accumulated (bias gradients:
 * the column sums of dy for nn.Linear biases).  The grouped form covers several matrices in one launch. */
int mmf_colsum_bf16(const void* x, float* out, int M, int N, int ldx, void* stream);
#define MMF_COLSUM_MAX_PROBLEMS 24
typedef struct mmf_colsum_problem {
  const void* x;      /* bf16 [M][ldx] */
  float* out;         /* f32 [N], accumulated */
  int32_t M, N, ldx;
} mmf_colsum_problem;
int mmf_colsum_grouped(const mmf_colsum_problem* problems, int num_problems, void* stream);
/* y = x * keep / (1-p) elementwise (nn.Dropout, training mode); is_f32 selects f32 or bf16 storage.
 * The same call with the same (*rng_state, site) on dy gives the backward.  In place allowed. */
int mmf_dropout(const void* x, void* y, int64_t n, int is_f32, float p, const uint64_t* rng_state,
                uint32_t site, void* stream);
/* base[starts[r] .. ends[r]) = 0 for up to MMF_ZERO_MAX_RANGES float ranges in ONE launch: the holes of the lazily
 * zeroed gradient arena (biases, LayerNorm vectors, gradients torch produces) between the matrices the wgrad GEMMs
 * overwrite.  base 16-byte aligned; starts / ends are HOST arrays of element offsets. */
#define MMF_ZERO_MAX_RANGES 48
int mmf_zero_ranges_f32(float* base, const int64_t* starts, const int64_t* ends, int n, void* stream);
/* relu backward on bf16: dx = dy * (y > 0) */
int mmf_relu_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, void* stream);
/* same with dy and / or y in f32 (flags non-zero): narrows and masks in one pass; dx is bf16 */
int mmf_relu_bwd_mixed(const void* dy, int dy_f32, const void* y, int y_f32, void* dx, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small fused kernels of the (B, d)-row branches (csrc/small.hip).  f32 unless said otherwise; every buffer
 * is caller-owned; parameter gradients (datt_*, dbias, dW2, db2, dW, db, demb) are ACCUMULATED into.
 * ------------------------------------------------------------------------------------------ */
/* Dense 3-node GAT layer = torch_geometric GATConv(heads, concat=False) on the directed 3-clique plus self
 * loops, as GraphFusion uses it (models/fusion_layers.py:223-232, 267-282): h [B][3][heads][C] is the node
 * features after the layer's linear map (an mmf_gemm / skinny launch);
 *   e[i,j,h] = leaky_relu(<h[i,h],att_dst[h]> + <h[j,h],att_src[h]>, negative_slope); alpha = softmax_j(e)
 *   (denominator + 1e-16, PyG), dropout(alpha, p); out[i] = mean_h sum_j alpha[i,j,h] h[j,h] + bias
 * y = relu(out) if relu (bf16 [B][3][C], the next layer's GEMM operand); pooled (optional, bf16 [B][C]) = mean over
 * the 3 nodes of y (global_mean_pool, :285-286).  alpha [B][3][3][heads] (before dropout) and sdots [B][2][3][heads]
 * are saved for the backward, which takes dy (bf16 [B][3][C]) and/or dpooled (bf16 [B][C]). */
typedef struct mmf_gat3_params {
  int32_t B, heads, C, relu;
  float negative_slope, dropout_p;
  const uint64_t* rng_state;
  uint32_t site;
} mmf_gat3_params;
int mmf_gat3_dense_fwd(const float* h, const float* att_src, const float* att_dst, const float* bias,
                       void* y_bf16, void* pooled_bf16, float* alpha, float* sdots,
                       const mmf_gat3_params* p, void* stream);
int mmf_gat3_dense_bwd(const float* h, const float* att_src, const float* att_dst, const void* y_bf16,
                       const float* alpha, const float* sdots, const void* dy_bf16, const void* dpooled_bf16,
                       float* dh, float* datt_src, float* datt_dst, float* dbias,
                       const mmf_gat3_params* p, void* stream);
/* n_m = z_m / max(||z_m||, 1e-12) for three (B, D) projections and, if losses != NULL, the symmetric InfoNCE of
 * the pairs (0,1) (0,2) (1,2): (CE(n_a n_b^T / T, arange) + CE(transposed, arange)) / 2 — ContrastiveFusion,
 * models/fusion_layers.py:338-347, 361-375.  B <= 64 (per-rank batch), D a multiple of 4.  inv_norm [3][B] and
 * lse [3][2][B] are saved for the backward; dn[m] / dloss[p] may be NULL (no gradient from that output). */
int mmf_infonce_fwd(const float* const z[3], float* const n[3], float* inv_norm, float* losses, float* lse,
                    int B, int D, float temperature, void* stream);
int mmf_infonce_bwd(const float* const n[3], const float* inv_norm, const float* lse, const float* const dn[3],
                    const float* const dloss[3], float* const dz[3], int B, int D, float temperature, void* stream);
/* AdaptiveFusion's weighting (:436-443): aw = softmax(hp W2^T + b2) [B][3]; weighted (bf16 [B][d]) =
 * sum_m attended[b][m][:] aw[b][m].  Backward: dweighted bf16, daw optional. */
int mmf_adaptive_combine_fwd(const float* hp, const float* W2, const float* b2, const float* attended,
                             float* aw, void* weighted_bf16, int B, int d, void* stream);
int mmf_adaptive_combine_bwd(const float* hp, const float* W2, const float* attended, const float* aw,
                             const void* dweighted_bf16, const float* daw, float* dattended, float* dhp,
                             float* dW2, float* db2, int B, int d, void* stream);
/* The head-averaged (B, 3, 3) attention weights AdaptiveFusion returns (:432-434; MultiheadAttention averages its
 * weights over the heads): qkv bf16 [B*3][3 heads head_dim] packed q | k | v.  No gradient (inspection only). */
int mmf_adaptive_attn_weights(const void* qkv_bf16, float* w, int B, int heads, int head_dim, void* stream);
/* The head-averaged (B, T, T) self-attention weights the reference's audio / video encoder heads return
 * (models/encoders.py:152-154,236-238): qkv bf16 [B*T][3 heads head_dim] packed q | k | v; T <= 2048, head_dim % 8 == 0.
 * No gradient (inspection only). */
int mmf_attn_weights_mean(const void* qkv_bf16, float* w, int B, int T, int heads, int head_dim, void* stream);
/* Narrow linear heads, 1 <= N <= 16 outputs, f32 masters (LateFusion :50-60, EmotionClassifier / valence /
 * arousal / uncertainty heads models/multimodal_model.py:56-60,186-219): y = x W^T + b.  dx may be NULL. */
int mmf_linear_narrow_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, void* stream);
int mmf_linear_narrow_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db,
                          int M, int N, int K, void* stream);
/* x[b][m][:] = feat_m[b][:] + emb[m][:] as bf16 rows (GraphFusion node stacking + type embedding, :255-264;
 * emb == NULL: plain stacking, AdaptiveFusion :427-429).  feat_m rows are ldf floats apart (3d when the three
 * are the thirds of one (B, 3d) buffer).  Backward: any of d0..d2, demb may be NULL; d_m rows ldd apart. */
int mmf_stack3_embed_fwd(const float* f0, const float* f1, const float* f2, const float* emb, void* x_bf16,
                         int B, int d, int ldf, void* stream);
int mmf_stack3_embed_bwd(const void* dx_bf16, float* d0, float* d1, float* d2, float* demb, int B, int d, int ldd,
                         void* stream);
/* y[b][:] = x[b][:] * mask[b]: ModalityDropout's per-sample keep masks (models/encoders.py:289-321, no rescale);
 * the same call on dy is the backward. */
int mmf_rowmask_apply(const float* x, const float* mask, float* y, int B, int d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-step tail on the flat arenas (reference recipe training/advanced_trainer.py:85-110,
 * 168-182: clip_grad_norm_(1.0) then AdamW).  out[0] += sum x^2 ;  AdamW with decoupled weight
 * decay, optional global-norm clipping from the device scalar gnorm_sq (NULL or hparams[7] <= 0:
 * none) and a gradient scale, updating the fp32 masters AND the bf16 shadow in one pass.
 * hparams is a DEVICE array of 9 floats: lr, beta1, beta2, eps, weight_decay, 1-beta1^t,
 * 1-beta2^t, max_grad_norm, grad_scale (device-resident so a captured graph can be replayed with
 * new values).
 * ------------------------------------------------------------------------------------------ */
int mmf_sqnorm_f32(const float* x, int64_t n, float* out, void* stream);
/* Device-side schedule (graph-capturable): *step += 1, then — for sched[0] == 1 — hparams[0] = the OneCycleLR(cos)
 * learning rate of optimiser step (*step - 1) and, for sched[6] != 0, hparams[1] = the cycled beta1 (OneCycleLR's
 * cycle_momentum, on by default and therefore part of the reference recipe, training/advanced_trainer.py:102-110);
 * finally hparams[5], [6] = 1 - beta^step from the current betas, as torch.optim.Adam forms them.
 * sched = {mode, max_lr, total_steps, pct_start, div_factor, final_div_factor, cycle_momentum, base_momentum,
 * max_momentum} as 9 doubles on the device; mode 0 leaves hparams[0], [1] alone. */
int mmf_adamw_advance(int64_t* step, float* hparams, const double* sched, void* stream);
int mmf_adamw_step(float* master, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                   int64_t n, const float* hparams, const float* gnorm_sq, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp32-storage parity mode (BASELINE.json north_star: "within 1e-3 fp32").  Every operand f32 in HBM, the exact f32
 * MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fmaf chain) instead of bf16 MFMA.  Same problem struct, layouts
 * and epilogue contract as mmf_gemm_grouped (A, B, C, aux are f32 here; MMF_EPI_DROPOUT is not available).
 * mmf_gemm_f32_batched runs ONE problem over an nb0 x nb1 batch: operand X of batch (i0, i1) starts at
 * X + i0 * strideX[0] + i1 * strideX[1] (elements) — the per-(batch, head) products of the explicit-scores attention
 * the reference computes (torch F.multi_head_attention_forward, need_weights path): S = Q K^T, softmax rows,
 * O = P V, and their backward products.  mmf_softmax_rows_f32: S <- softmax(scale * S) per row, in place;
 * mmf_softmax_bwd_rows_f32: dP <- scale * P * (dP - rowsum(dP * P)), in place.  mmf_layernorm_f32_*: nn.LayerNorm
 * with f32 rows (dgamma / dbeta are accumulated into with atomics).
 * ------------------------------------------------------------------------------------------ */
int mmf_gemm_f32_grouped(const mmf_gemm_problem* problems, int num_problems, int layout, int epilogue, float alpha,
                         void* stream);
int mmf_gemm_f32_batched(const mmf_gemm_problem* problem, int layout, int epilogue, float alpha, int nb0, int nb1,
                         const int64_t strideA[2], const int64_t strideB[2], const int64_t strideC[2], void* stream);
int mmf_softmax_rows_f32(float* S, int64_t rows, int cols, float scale, void* stream);
int mmf_softmax_bwd_rows_f32(const float* P, float* dP, int64_t rows, int cols, float scale, void* stream);
int mmf_layernorm_f32_fwd(const float* x, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                          int rows, int d, float eps, void* stream);
int mmf_layernorm_f32_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd,
                          float* dx, float* dgamma, float* dbeta, int rows, int d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bidirectional LSTM layer recurrence (replaces the per-time-step loop of torch.nn.LSTM / MIOpen behind
 * reference models/encoders.py:183-190,233: nn.LSTM(768, 384, num_layers=2, batch_first=True, bidirectional=True)).
 * The input projections of all time steps are the caller's grouped GEMM (gx); this entry point runs the T sequential
 * steps of both directions in ONE persistent launch with W_hh resident in registers (csrc/lstm.hip), and its backward
 * produces the gate pre-activation gradients dgates that the caller feeds to the grouped dgrad / wgrad GEMMs.
 * Time-major layout: row t*B + b.  Gate order i, f, g, o (torch).  Columns of the 2*4H-wide buffers:
 * [direction 0 | direction 1] x [gate] x [H]; of the 2H-wide buffers: [direction 0 H | direction 1 H].
 * y has a zero row block of B rows in front of time 0 and behind time T-1 (the caller zeroes them once): time t lives in
 * row block t + 1.  B <= 64 per call; H (hidden size per direction) in {64, 128, 384}.
 * workspace: >= mmf_bilstm_workspace_bytes() bytes of device memory, 4-byte aligned; zeroed by the call (stream-ordered);
 * after the launch int32 word [2] is 0, or 1 if a workgroup gave up waiting for its peers (results then invalid).
 * ------------------------------------------------------------------------------------------ */
typedef struct mmf_bilstm_args {
  const void* gx;          /* fwd: f32 (T*B, 2*4H)  x_t W_ih^T, no bias                                        */
  const void* w_hh[2];     /* bf16 (4H, H) per direction                                                      */
  const float* b_ih[2];    /* fwd: f32 (4H)                                                                   */
  const float* b_hh[2];    /* fwd: f32 (4H)                                                                   */
  void* y;                 /* fwd out: bf16 ((T+2)*B, 2H), h_t of time t in row block t+1                      */
  float* gates;            /* fwd out / bwd in: f32 (T*B, 2*4H) gate activations                               */
  float* cell;             /* fwd out / bwd in: f32 (T*B, 2H) cell state c_t                                   */
  const void* dy;          /* bwd in: bf16 (T*B, 2H) gradient of the layer output                              */
  void* dgates;            /* bwd out: bf16 (T*B, 2*4H) gradient of the gate pre-activations                   */
  int32_t T, B, H;
} mmf_bilstm_args;
size_t mmf_bilstm_workspace_bytes(void);
int mmf_bilstm_layer_fwd(const mmf_bilstm_args* args, void* workspace, size_t workspace_bytes, void* stream);
int mmf_bilstm_layer_bwd(const mmf_bilstm_args* args, void* workspace, size_t workspace_bytes, void* stream);

/* out[(i1 * n0 + i0) * d + :] = in[(i0 * n1 + i1) * d + :]  — swaps the two leading axes of an (n0, n1, d) tensor
 * ((B, T, d) <-> (T, B, d)); in_f32 selects an f32 source, the destination is bf16 unless out_f32. d % 8 == 0. */
int mmf_swap01(const void* in, void* out, int n0, int n1, int d, int in_f32, int out_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMFUSION_H */
