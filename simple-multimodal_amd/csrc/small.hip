// Small fused kernels of the (B, d)-row fusion branches (B ~ 16 per rank): dense 3-node GAT layer, L2-normalise +
// symmetric InfoNCE, adaptive softmax-weighted combination, narrow (N <= 16) linear heads, node stacking with
// type embedding, per-sample modality masks.  Each replaces 10-40 torch glue launches of a few microseconds
// (profiles/r01_hier_kernel_stats.csv: 364 torch launches, 1.14 ms per hier-seq step before these existed).
// They are latency-bound: one 256-thread workgroup per sample (or one workgroup in all for the B x B InfoNCE),
// f32 arithmetic, wave reductions + one LDS hop; parameter gradients are accumulated straight into the f32
// gradient arena (atomics across samples, 16 adders per address).
#include "mmf_internal.h"

namespace {

constexpr int SM_THREADS = 256;
constexpr int SM_WAVES = SM_THREADS / 64;

// Sum N per-thread partials over the workgroup; the totals land in red[0..N) (LDS, >= N floats) for every thread.
template <int N, int NW = SM_WAVES>
__device__ __forceinline__ void block_sum(float (&v)[N], float* red, float* scratch /* [NW][N] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float s = wave_sum(v[i]);
    if (lane == 0) scratch[wave * N + i] = s;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += scratch[w * N + threadIdx.x];
    red[threadIdx.x] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ float bf16_at(const unsigned short* p, size_t i) { return bf16_bits_to_f32(p[i]); }

// ================================================================================================
// Dense 3-node GAT layer (PyG GATConv(heads=H, concat=False) on the directed 3-clique + self loops,
// reference models/fusion_layers.py:267-282 through torch_geometric).
//   s_src[j,h] = <h[j,h,:], att_src[h,:]>, s_dst[i,h] = <h[i,h,:], att_dst[h,:]>
//   e[i,j,h] = leaky_relu(s_dst[i,h] + s_src[j,h], 0.2); alpha = softmax_j(e) (denominator + 1e-16); dropout(alpha)
//   out[i,c] = mean_h sum_j alpha[i,j,h] h[j,h,c] + bias[c];  y = relu(out) (bf16);  pooled = mean_i y
// ================================================================================================

struct Gat3Args {
  const float* h; const float* att_src; const float* att_dst; const float* bias;
  unsigned short* y; unsigned short* pooled; float* alpha; float* sdots;
  const unsigned short* dy; const unsigned short* dpool; float* dh; float* datt_src; float* datt_dst; float* dbias;
  int B, H, C, relu;
  float slope, inv_keep; unsigned drop_thresh, site; const unsigned long long* rng_state;
};

constexpr int GAT_THREADS = 512, GAT_WAVES = GAT_THREADS / 64;   // C = 768 columns: at most two per thread

template <int H>
__global__ __launch_bounds__(GAT_THREADS)
void gat3_fwd_kernel(const Gat3Args a) {
  __shared__ float red[6 * H], scratch[GAT_WAVES * 6 * H], alpha_s[9 * H];
  const int b = blockIdx.x, C = a.C, tid = threadIdx.x;
  const float* hb = a.h + (size_t)b * 3 * H * C;
  float part[6 * H];                                      // [src|dst][j][h]
#pragma unroll
  for (int i = 0; i < 6 * H; ++i) part[i] = 0.f;
  for (int c = tid; c < C; c += GAT_THREADS) {
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const float as = a.att_src[hh * C + c], ad = a.att_dst[hh * C + c];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float v = hb[(j * H + hh) * C + c];
        part[j * H + hh] += v * as;
        part[3 * H + j * H + hh] += v * ad;
      }
    }
  }
  block_sum<6 * H, GAT_WAVES>(part, red, scratch);
  if (tid < 6 * H) a.sdots[(size_t)b * 6 * H + tid] = red[tid];
  if (tid < 3 * H) {                                      // thread = (i, hh): softmax over the three sources j
    const int i = tid / H, hh = tid % H;
    float e[3], mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float z = red[3 * H + i * H + hh] + red[j * H + hh];
      e[j] = z > 0.f ? z : z * a.slope;
      mx = fmaxf(mx, e[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) { e[j] = __expf(e[j] - mx); sum += e[j]; }
    const float inv = 1.f / (sum + 1e-16f);
    const unsigned key = a.drop_thresh ? mmf_rng_key(*a.rng_state, a.site, (unsigned)b) : 0u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float al = e[j] * inv;
      a.alpha[((size_t)b * 9 + i * 3 + j) * H + hh] = al;
      float ad = al;
      if (a.drop_thresh) ad = mmf_keep(key, (unsigned)((i * 3 + j) * H + hh), a.drop_thresh) ? al * a.inv_keep : 0.f;
      alpha_s[(i * 3 + j) * H + hh] = ad;
    }
  }
  __syncthreads();
  const float invH = 1.f / H;
  for (int c = tid; c < C; c += GAT_THREADS) {
    float hv[3 * H];
#pragma unroll
    for (int q = 0; q < 3 * H; ++q) hv[q] = hb[q * C + c];
    float pool = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int hh = 0; hh < H; ++hh) o += alpha_s[(i * 3 + j) * H + hh] * hv[j * H + hh];
      o = o * invH + a.bias[c];
      if (a.relu) o = fmaxf(o, 0.f);
      a.y[((size_t)b * 3 + i) * C + c] = f32_to_bf16_bits(o);
      pool += o;
    }
    if (a.pooled) a.pooled[(size_t)b * C + c] = f32_to_bf16_bits(pool * (1.f / 3.f));
  }
}

template <int H>
__global__ __launch_bounds__(GAT_THREADS)
void gat3_bwd_kernel(const Gat3Args a) {
  __shared__ float red[9 * H], scratch[GAT_WAVES * 9 * H], alpha_d[9 * H], dssrc[3 * H], dsdst[3 * H];
  const int b = blockIdx.x, C = a.C, tid = threadIdx.x;
  const float* hb = a.h + (size_t)b * 3 * H * C;
  const float invH = 1.f / H;
  auto grad_out = [&](int i, int c) -> float {            // d(out[i,c]) through the pooling and the ReLU
    float g = 0.f;
    if (a.dy) g += bf16_at(a.dy, ((size_t)b * 3 + i) * C + c);
    if (a.dpool) g += bf16_at(a.dpool, (size_t)b * C + c) * (1.f / 3.f);
    if (a.relu && !(bf16_at(a.y, ((size_t)b * 3 + i) * C + c) > 0.f)) g = 0.f;
    return g;
  };
  float part[9 * H];                                      // d(alpha_dropped)[i][j][h] = sum_c g[i,c]/H * h[j,h,c]
#pragma unroll
  for (int q = 0; q < 9 * H; ++q) part[q] = 0.f;
  for (int c = tid; c < C; c += GAT_THREADS) {
    float g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = grad_out(i, c) * invH;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        const float v = hb[(j * H + hh) * C + c];
#pragma unroll
        for (int i = 0; i < 3; ++i) part[(i * 3 + j) * H + hh] += g[i] * v;
      }
  }
  block_sum<9 * H, GAT_WAVES>(part, red, scratch);
  if (tid < 3 * H) {                                      // thread = (i, hh): dropout, softmax and leaky-relu backward
    const int i = tid / H, hh = tid % H;
    const unsigned key = a.drop_thresh ? mmf_rng_key(*a.rng_state, a.site, (unsigned)b) : 0u;
    const float* sd = a.sdots + (size_t)b * 6 * H;
    float al[3], dal[3], dot = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      al[j] = a.alpha[((size_t)b * 9 + i * 3 + j) * H + hh];
      float keep = 1.f;
      if (a.drop_thresh) keep = mmf_keep(key, (unsigned)((i * 3 + j) * H + hh), a.drop_thresh) ? a.inv_keep : 0.f;
      alpha_d[(i * 3 + j) * H + hh] = al[j] * keep;
      dal[j] = red[(i * 3 + j) * H + hh] * keep;
      dot += al[j] * dal[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float z = sd[3 * H + i * H + hh] + sd[j * H + hh];
      const float dz = al[j] * (dal[j] - dot) * (z > 0.f ? 1.f : a.slope);
      red[(i * 3 + j) * H + hh] = dz;                     // own slots only: no other thread reads them before the barrier
    }
  }
  __syncthreads();
  if (tid < 3 * H) {                                      // thread = (node, hh)
    const int n = tid / H, hh = tid % H;
    float s = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { s += red[(k * 3 + n) * H + hh]; d += red[(n * 3 + k) * H + hh]; }
    dssrc[n * H + hh] = s;                                // d s_src[j = n]: sum over destinations i
    dsdst[n * H + hh] = d;                                // d s_dst[i = n]: sum over sources j
  }
  __syncthreads();
  for (int c = tid; c < C; c += GAT_THREADS) {
    float g[3], gb = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float go = grad_out(i, c); gb += go; g[i] = go * invH; }
    atomicAdd(a.dbias + c, gb);
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const float as = a.att_src[hh * C + c], ad = a.att_dst[hh * C + c];
      float das = 0.f, dad = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float v = hb[(j * H + hh) * C + c];
        float dv = dssrc[j * H + hh] * as + dsdst[j * H + hh] * ad;
#pragma unroll
        for (int i = 0; i < 3; ++i) dv += alpha_d[(i * 3 + j) * H + hh] * g[i];
        a.dh[((size_t)b * 3 * H + j * H + hh) * C + c] = dv;
        das += dssrc[j * H + hh] * v;
        dad += dsdst[j * H + hh] * v;
      }
      atomicAdd(a.datt_src + hh * C + c, das);
      atomicAdd(a.datt_dst + hh * C + c, dad);
    }
  }
}

// ================================================================================================
// L2-normalise three (B, D) projections and the symmetric InfoNCE of the three pairs (reference
// models/fusion_layers.py:338-347, 361-375): one workgroup, B <= 64 (per-rank batch; sims live in LDS).
//   n_m = z_m / max(||z_m||, 1e-12); sim_p = n_a n_b^T / T; loss_p = (CE(sim_p, arange) + CE(sim_p^T, arange)) / 2
// pairs p: (0,1) (0,2) (1,2)
// ================================================================================================
constexpr int NCE_MAXB = 64;
struct NceArgs {
  const float* z[3]; float* n[3]; float* inv_norm;        // inv_norm [3][B]
  float* losses;                                          // [3], may be NULL (normalise only)
  float* lse;                                             // [3][2][B] row / column log-sum-exp, saved for backward
  const float* dn[3];                                     // backward: grads w.r.t. the normalised outputs (may be NULL)
  const float* dloss[3];                                  // backward: device scalars (may be NULL)
  float* dz[3];
  int B, D; float inv_temp;
};

__device__ __forceinline__ void nce_pair(int p, int& ma, int& mb) { ma = p == 2 ? 1 : 0; mb = p == 0 ? 1 : 2; }

constexpr int NCE_THREADS = 1024, NCE_WAVES = NCE_THREADS / 64;

// Four similarities at once: x . y[j0 + u], u = 0..3 (rows past B are clamped; the caller ignores them).  The five
// row segments of each column step are independent loads, so a wave has them in flight together instead of
// walking a chain of L2 round trips (the first version of these kernels spent ~1,000 cycles per similarity).
__device__ __forceinline__ void wave_dot4(const float* __restrict__ x, const float* __restrict__ y, int j0, int B, int D,
                                          int lane, float (&out)[4]) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4_t u = *reinterpret_cast<const f32x4_t*>(x + c);
    f32x4_t v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f32x4_t*>(y + (size_t)min(j0 + t, B - 1) * D + c);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += u[0] * v[t][0] + u[1] * v[t][1] + u[2] * v[t][2] + u[3] * v[t][3];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) out[t] = wave_sum(acc[t]);
}

// forward: one 16-wave workgroup per pair p.  It normalises the two modalities of its pair (each modality is
// normalised by two workgroups, writing identical values), then one wave per row of similarities, four at a time,
// then the row / column log-sum-exps and the loss.
__global__ __launch_bounds__(NCE_THREADS)
void nce_fwd_kernel(const NceArgs a) {
  extern __shared__ float sim[];                          // [B][B]
  __shared__ float lsum;
  const int B = a.B, D = a.D, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = blockIdx.x;
  int ma, mb; nce_pair(p, ma, mb);
  for (int r = wave; r < 2 * B; r += NCE_WAVES) {         // one wave per row: norm, then the normalised row
    const int m = r < B ? ma : mb, i = r < B ? r : r - B;
    const float* z = a.z[m] + (size_t)i * D;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(z + c);
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    if (lane == 0) a.inv_norm[m * B + i] = inv;
    float* n = a.n[m] + (size_t)i * D;
    for (int c = lane * 4; c < D; c += 256)
      *reinterpret_cast<f32x4_t*>(n + c) = *reinterpret_cast<const f32x4_t*>(z + c) * inv;
  }
  if (!a.losses) return;
  if (tid == 0) lsum = 0.f;
  __syncthreads();                                        // this workgroup's own writes of n are visible to it
  for (int i = wave; i < B; i += NCE_WAVES) {
    for (int j0 = 0; j0 < B; j0 += 4) {
      float s4[4];
      wave_dot4(a.n[ma] + (size_t)i * D, a.n[mb], j0, B, D, lane, s4);
      if (lane < 4 && j0 + lane < B) sim[i * B + j0 + lane] = s4[lane] * a.inv_temp;
    }
  }
  __syncthreads();
  for (int t = tid; t < 2 * B; t += NCE_THREADS) {        // (row | col, i): log-sum-exp and the diagonal term
    const int col = t / B, i = t % B;
    float mx = -3.0e38f;
    for (int j = 0; j < B; ++j) mx = fmaxf(mx, col ? sim[j * B + i] : sim[i * B + j]);
    float sum = 0.f;
    for (int j = 0; j < B; ++j) sum += __expf((col ? sim[j * B + i] : sim[i * B + j]) - mx);
    const float lse = mx + __logf(sum);
    a.lse[p * 2 * B + t] = lse;
    atomicAdd(&lsum, (lse - sim[i * B + i]) * (0.5f / B));
  }
  __syncthreads();
  if (tid == 0) a.losses[p] = lsum;
}

// backward: 16-wave workgroup (m, block of 16 rows).  It first rebuilds d loss / d sim of the two pairs that contain
// modality m (one wave per similarity row, four similarities at a time), then one wave per row of dz_m:
// dn = dn_ext + sum over the pairs of dsim . n_other (float4 columns, four source rows in flight), and the
// projection of the normalisation.
__global__ __launch_bounds__(NCE_THREADS)
void nce_bwd_kernel(const NceArgs a) {
  extern __shared__ float dsim[];                         // [2][B][B]: the two pairs of this modality, 1/T included
  const int B = a.B, D = a.D, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = blockIdx.x;
  int pairs[2], q = 0;
  for (int p = 0; p < 3; ++p) { int ma, mb; nce_pair(p, ma, mb); if (ma == m || mb == m) pairs[q++] = p; }
  for (int u = wave; u < 2 * B; u += NCE_WAVES) {         // (pair k, row i of its first member)
    const int k = u / B, i = u % B, p = pairs[k];
    int ma, mb; nce_pair(p, ma, mb);
    const bool live = a.dloss[p] != nullptr;              // wave-uniform
    const float dl = live ? *a.dloss[p] * (0.5f / B) * a.inv_temp : 0.f;
    const float lr = live ? a.lse[(p * 2 + 0) * B + i] : 0.f;
    for (int j0 = 0; j0 < B; j0 += 4) {
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
      if (live) wave_dot4(a.n[ma] + (size_t)i * D, a.n[mb], j0, B, D, lane, s4);
      if (lane < 4 && j0 + lane < B) {
        const int j = j0 + lane;
        float g = 0.f;
        if (live) {
          const float s = s4[lane] * a.inv_temp;
          g = dl * (__expf(s - lr) + __expf(s - a.lse[(p * 2 + 1) * B + j]) - (i == j ? 2.f : 0.f));
        }
        dsim[(k * B + i) * B + j] = g;
      }
    }
  }
  __syncthreads();
  const int i = blockIdx.y * NCE_WAVES + wave;
  if (i >= B) return;
  const float* n = a.n[m] + (size_t)i * D;
  float* dz = a.dz[m] + (size_t)i * D;
  float dot = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    f32x4_t g = a.dn[m] ? *reinterpret_cast<const f32x4_t*>(a.dn[m] + (size_t)i * D + c) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int ma, mb; nce_pair(pairs[k], ma, mb);
      const bool first = ma == m;                         // m is the pair's first member: weights are row i of dsim
      const float* o = a.n[first ? mb : ma];
      const float* ds = dsim + k * B * B;
      for (int j0 = 0; j0 < B; j0 += 4) {
        f32x4_t v[4];
        float w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int j = min(j0 + t, B - 1);
          v[t] = *reinterpret_cast<const f32x4_t*>(o + (size_t)j * D + c);
          w[t] = j0 + t < B ? (first ? ds[i * B + j] : ds[j * B + i]) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) g += v[t] * w[t];
      }
    }
    *reinterpret_cast<f32x4_t*>(dz + c) = g;              // dn for now; projected below (same lane re-reads it)
    const f32x4_t nv = *reinterpret_cast<const f32x4_t*>(n + c);
    dot += g[0] * nv[0] + g[1] * nv[1] + g[2] * nv[2] + g[3] * nv[3];
  }
  dot = wave_sum(dot);
  const float inv = a.inv_norm[m * B + i];
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4_t g = *reinterpret_cast<const f32x4_t*>(dz + c), nv = *reinterpret_cast<const f32x4_t*>(n + c);
    *reinterpret_cast<f32x4_t*>(dz + c) = (g - nv * dot) * inv;
  }
}

// ================================================================================================
// Adaptive combination (reference models/fusion_layers.py:436-443): aw = softmax(hp W2^T + b2) over the three
// modalities, weighted = sum_m attended[:, m, :] aw[:, m]; one workgroup per sample.
// ================================================================================================
struct AdaArgs {
  const float* hp; const float* W2; const float* b2; const float* attended;
  float* aw; unsigned short* weighted;
  const unsigned short* dweighted; const float* daw_ext; float* dattended; float* dhp; float* dW2; float* db2;
  int B, d;
};

__global__ __launch_bounds__(SM_THREADS)
void ada_fwd_kernel(const AdaArgs a) {
  __shared__ float red[3], scratch[SM_WAVES * 3], aw_s[3];
  const int b = blockIdx.x, d = a.d, tid = threadIdx.x;
  float part[3] = {0.f, 0.f, 0.f};
  for (int c = tid; c < d; c += SM_THREADS) {
    const float x = a.hp[(size_t)b * d + c];
#pragma unroll
    for (int m = 0; m < 3; ++m) part[m] += x * a.W2[m * d + c];
  }
  block_sum<3>(part, red, scratch);
  if (tid == 0) {
    float l[3], mx = -3.0e38f, s = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) { l[m] = red[m] + a.b2[m]; mx = fmaxf(mx, l[m]); }
#pragma unroll
    for (int m = 0; m < 3; ++m) { l[m] = __expf(l[m] - mx); s += l[m]; }
#pragma unroll
    for (int m = 0; m < 3; ++m) { aw_s[m] = l[m] / s; a.aw[b * 3 + m] = aw_s[m]; }
  }
  __syncthreads();
  for (int c = tid; c < d; c += SM_THREADS) {
    float w = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) w += a.attended[((size_t)b * 3 + m) * d + c] * aw_s[m];
    a.weighted[(size_t)b * d + c] = f32_to_bf16_bits(w);
  }
}

__global__ __launch_bounds__(SM_THREADS)
void ada_bwd_kernel(const AdaArgs a) {
  __shared__ float red[3], scratch[SM_WAVES * 3], dl_s[3];
  const int b = blockIdx.x, d = a.d, tid = threadIdx.x;
  float aw[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) aw[m] = a.aw[b * 3 + m];
  float part[3] = {0.f, 0.f, 0.f};
  for (int c = tid; c < d; c += SM_THREADS) {
    const float g = a.dweighted ? bf16_at(a.dweighted, (size_t)b * d + c) : 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      part[m] += g * a.attended[((size_t)b * 3 + m) * d + c];
      a.dattended[((size_t)b * 3 + m) * d + c] = g * aw[m];
    }
  }
  block_sum<3>(part, red, scratch);
  if (tid == 0) {
    float da[3], dot = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) { da[m] = red[m] + (a.daw_ext ? a.daw_ext[b * 3 + m] : 0.f); dot += aw[m] * da[m]; }
#pragma unroll
    for (int m = 0; m < 3; ++m) { dl_s[m] = aw[m] * (da[m] - dot); atomicAdd(a.db2 + m, dl_s[m]); }
  }
  __syncthreads();
  for (int c = tid; c < d; c += SM_THREADS) {
    const float x = a.hp[(size_t)b * d + c];
    float g = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      g += dl_s[m] * a.W2[m * d + c];
      atomicAdd(a.dW2 + m * d + c, dl_s[m] * x);
    }
    a.dhp[(size_t)b * d + c] = g;
  }
}

// Head-averaged 3 x 3 attention weights of AdaptiveFusion's 3-token self-attention (reference :432-434 returns them
// for inspection; nn.MultiheadAttention(need_weights=True) averages over heads): qkv bf16 [B*3][3 d] packed
// (q | k | v), w[b][i][j] = mean_h softmax_j(q[b,i,h].k[b,j,h] / sqrt(dh)).  One workgroup per sample, no gradient.
constexpr int ADAW_MAXH = 16;
__global__ __launch_bounds__(SM_THREADS)
void ada_attn_weights_kernel(const unsigned short* __restrict__ qkv, float* __restrict__ w, int H, int dh, float scale) {
  __shared__ float sc[9 * ADAW_MAXH];
  const int b = blockIdx.x, d = H * dh, tid = threadIdx.x;
  if (tid < 9 * H) {
    const int i = tid / (3 * H), j = (tid / H) % 3, h = tid % H;
    const unsigned short* q = qkv + ((size_t)b * 3 + i) * 3 * d + h * dh;
    const unsigned short* kk = qkv + ((size_t)b * 3 + j) * 3 * d + d + h * dh;
    float s = 0.f;
    for (int c = 0; c < dh; ++c) s += bf16_bits_to_f32(q[c]) * bf16_bits_to_f32(kk[c]);
    sc[tid] = s * scale;
  }
  __syncthreads();
  if (tid < 3 * H) {                                      // (i, h): softmax over j in place
    const int i = tid / H, h = tid % H;
    float e[3], mx = -3.0e38f, sum = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) { e[j] = sc[(i * 3 + j) * H + h]; mx = fmaxf(mx, e[j]); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { e[j] = __expf(e[j] - mx); sum += e[j]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) sc[(i * 3 + j) * H + h] = e[j] / sum;
  }
  __syncthreads();
  if (tid < 9) {
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += sc[tid * H + h];
    w[(size_t)b * 9 + tid] = s / (float)H;
  }
}

// Head-averaged attention weights of a self-attention over T tokens, w[b][i][j] = mean_h softmax_j(q_i.k_j / sqrt(dh)):
// what nn.MultiheadAttention returns beside its output and the reference's audio / video encoders pass on
// (models/encoders.py:152-154,236-238).  qkv: bf16 rows [B*T][3 d] packed q | k | v.  One workgroup per (query i, b);
// a thread owns keys j = tid, tid + 256, ... (T <= 256 * AWM_MAXK); fp32 dot products straight from the bf16 rows
// (the row of q sits in LDS).  No gradient: inspection output, like the (B, 3, 3) weights above.
constexpr int AWM_MAXK = 8;
__global__ __launch_bounds__(SM_THREADS)
void attn_weights_mean_kernel(const unsigned short* __restrict__ qkv, float* __restrict__ w, int T, int H, int dh, float scale) {
  extern __shared__ float awm_q[];                       // d floats
  __shared__ float red[SM_WAVES];
  const int i = blockIdx.x, b = blockIdx.y, d = H * dh, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned short* qrow = qkv + ((size_t)b * T + i) * 3 * d;
  for (int c = tid; c < d; c += SM_THREADS) awm_q[c] = bf16_bits_to_f32(qrow[c]) * scale;
  __syncthreads();
  float outv[AWM_MAXK];
#pragma unroll
  for (int k = 0; k < AWM_MAXK; ++k) outv[k] = 0.f;
  for (int h = 0; h < H; ++h) {
    float sc[AWM_MAXK], mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < AWM_MAXK; ++k) {
      const int j = tid + SM_THREADS * k;
      sc[k] = -3.0e38f;
      if (j < T) {
        const unsigned short* kr = qkv + ((size_t)b * T + j) * 3 * d + d + h * dh;
        float s = 0.f;
        for (int c = 0; c < dh; c += 8) {
          const u32x4_t x = *reinterpret_cast<const u32x4_t*>(kr + c);
          const float* qq = awm_q + h * dh + c;
          s += bf16lo(x[0]) * qq[0] + bf16hi(x[0]) * qq[1] + bf16lo(x[1]) * qq[2] + bf16hi(x[1]) * qq[3] +
               bf16lo(x[2]) * qq[4] + bf16hi(x[2]) * qq[5] + bf16lo(x[3]) * qq[6] + bf16hi(x[3]) * qq[7];
        }
        sc[k] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < AWM_MAXK; ++k) {
      sc[k] = (tid + SM_THREADS * k < T) ? __expf(sc[k] - mx) : 0.f;
      sum += sc[k];
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[0] + red[1] + red[2] + red[3]) * (float)H);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < AWM_MAXK; ++k) outv[k] += sc[k] * inv;
  }
#pragma unroll
  for (int k = 0; k < AWM_MAXK; ++k) {
    const int j = tid + SM_THREADS * k;
    if (j < T) w[((size_t)b * T + i) * T + j] = outv[k];
  }
}

// ================================================================================================
// Narrow linear heads, N <= 16 outputs (LateFusion's three d -> 7 classifiers :50-60, EmotionClassifier's last
// layer and the valence / arousal / uncertainty heads, models/multimodal_model.py:56-60,186-219), f32 in / f32
// master weights / f32 out: y[m][n] = sum_k x[m][k] W[n][k] + b[n]
// ================================================================================================
constexpr int NARROW_MAXN = 16;
struct NarrowArgs {
  const float* x; const float* W; const float* b; float* y;
  const float* dy; float* dx; float* dW; float* db;
  int M, N, K;
};

__global__ __launch_bounds__(SM_THREADS)
void narrow_fwd_kernel(const NarrowArgs a) {
  __shared__ float red[NARROW_MAXN], scratch[SM_WAVES * NARROW_MAXN];
  const int m = blockIdx.x, K = a.K, N = a.N, tid = threadIdx.x;
  float part[NARROW_MAXN];
#pragma unroll
  for (int n = 0; n < NARROW_MAXN; ++n) part[n] = 0.f;
  for (int k = tid; k < K; k += SM_THREADS) {
    const float x = a.x[(size_t)m * K + k];
#pragma unroll
    for (int n = 0; n < NARROW_MAXN; ++n)
      if (n < N) part[n] += x * a.W[(size_t)n * K + k];
  }
  block_sum<NARROW_MAXN>(part, red, scratch);
  if (tid < N) a.y[(size_t)m * N + tid] = red[tid] + (a.b ? a.b[tid] : 0.f);
}

// thread = one input column k (for every row m and output n): dx[m][k], dW[n][k] += ..., block 0 also db[n]
__global__ __launch_bounds__(SM_THREADS)
void narrow_bwd_kernel(const NarrowArgs a) {
  const int k = blockIdx.x * SM_THREADS + threadIdx.x, K = a.K, N = a.N, M = a.M;
  if (blockIdx.x == 0 && threadIdx.x < N && a.db) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += a.dy[(size_t)m * N + threadIdx.x];
    a.db[threadIdx.x] += s;
  }
  if (k >= K) return;
  float w[NARROW_MAXN], dw[NARROW_MAXN];
#pragma unroll
  for (int n = 0; n < NARROW_MAXN; ++n) { w[n] = n < N ? a.W[(size_t)n * K + k] : 0.f; dw[n] = 0.f; }
  for (int m = 0; m < M; ++m) {
    const float x = a.x[(size_t)m * K + k];
    float g = 0.f;
#pragma unroll
    for (int n = 0; n < NARROW_MAXN; ++n)
      if (n < N) { const float d = a.dy[(size_t)m * N + n]; g += d * w[n]; dw[n] += d * x; }
    if (a.dx) a.dx[(size_t)m * K + k] = g;
  }
#pragma unroll
  for (int n = 0; n < NARROW_MAXN; ++n)
    if (n < N) a.dW[(size_t)n * K + k] += dw[n];
}

// ================================================================================================
// Node stacking + type embedding (reference :255-264): x[b][m][:] = feat_m[b][:] + emb[m][:]  -> bf16 rows
// backward: dfeat_m[b][:] = dx[b][m][:], demb[m][:] += sum_b dx[b][m][:]
// ================================================================================================
__global__ __launch_bounds__(SM_THREADS)
void stack3_fwd_kernel(const float* f0, const float* f1, const float* f2, const float* emb, unsigned short* x, int B, int d, int ldf) {
  const int64_t n = (int64_t)B * 3 * d;
  for (int64_t e = (int64_t)blockIdx.x * SM_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * SM_THREADS) {
    const int c = (int)(e % d), m = (int)((e / d) % 3), b = (int)(e / (3 * (int64_t)d));
    const float* f = m == 0 ? f0 : (m == 1 ? f1 : f2);
    x[e] = f32_to_bf16_bits(f[(size_t)b * ldf + c] + (emb ? emb[m * d + c] : 0.f));
  }
}
__global__ __launch_bounds__(SM_THREADS)
void stack3_bwd_kernel(const unsigned short* dx, float* d0, float* d1, float* d2, float* demb, int B, int d, int ldd) {
  const int e = blockIdx.x * SM_THREADS + threadIdx.x;     // (m, c)
  if (e >= 3 * d) return;
  const int m = e / d, c = e % d;
  float* dst = m == 0 ? d0 : (m == 1 ? d1 : d2);
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = bf16_at(dx, ((size_t)b * 3 + m) * d + c);
    if (dst) dst[(size_t)b * ldd + c] = g;
    s += g;
  }
  if (demb) demb[e] += s;
}

// y[b][:] = x[b][:] * mask[b]   (ModalityDropout keep-masks, models/encoders.py:289-321: no 1/(1-p) rescale)
__global__ __launch_bounds__(SM_THREADS)
void rowmask_kernel(const float* x, const float* mask, float* y, int B, int d) {
  const int64_t n = (int64_t)B * d;
  for (int64_t e = (int64_t)blockIdx.x * SM_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * SM_THREADS)
    y[e] = x[e] * mask[e / d];
}

template <typename F>
int dispatch_heads(int H, F&& f) {
  switch (H) {
    case 1: f(std::integral_constant<int, 1>{}); return MMF_OK;
    case 2: f(std::integral_constant<int, 2>{}); return MMF_OK;
    case 4: f(std::integral_constant<int, 4>{}); return MMF_OK;
    case 8: f(std::integral_constant<int, 8>{}); return MMF_OK;
  }
  MMF_FAIL(MMF_E_UNSUPPORTED, "mmf_gat3_dense: heads=%d (supported: 1, 2, 4, 8)", H);
}

int fill_gat(Gat3Args& a, const mmf_gat3_params* p) {
  if (!p || p->B <= 0 || p->C <= 0) MMF_FAIL(MMF_E_SHAPE, "mmf_gat3_dense: B=%d C=%d", p ? p->B : 0, p ? p->C : 0);
  if (!(p->dropout_p >= 0.f) || p->dropout_p >= 1.f || (p->dropout_p > 0.f && !p->rng_state))
    MMF_FAIL(MMF_E_SHAPE, "mmf_gat3_dense: dropout needs 0 <= p < 1 and an rng_state");
  a.B = p->B; a.H = p->heads; a.C = p->C; a.relu = p->relu; a.slope = p->negative_slope;
  a.drop_thresh = p->dropout_p > 0.f ? mmf_drop_thresh(p->dropout_p) : 0u;
  a.inv_keep = a.drop_thresh ? 1.f / (1.f - (float)a.drop_thresh * (1.f / 4294967296.f)) : 1.f;
  a.site = p->site;
  a.rng_state = reinterpret_cast<const unsigned long long*>(p->rng_state);
  return MMF_OK;
}

}  // namespace

extern "C" int mmf_gat3_dense_fwd(const float* h, const float* att_src, const float* att_dst, const float* bias,
                                  void* y_bf16, void* pooled_bf16, float* alpha, float* sdots,
                                  const mmf_gat3_params* p, void* stream) {
  Gat3Args a = {};
  if (int rc = fill_gat(a, p)) return rc;
  if (!h || !att_src || !att_dst || !bias || !y_bf16 || !alpha || !sdots) MMF_FAIL(MMF_E_SHAPE, "mmf_gat3_dense_fwd: null operand");
  a.h = h; a.att_src = att_src; a.att_dst = att_dst; a.bias = bias;
  a.y = static_cast<unsigned short*>(y_bf16); a.pooled = static_cast<unsigned short*>(pooled_bf16);
  a.alpha = alpha; a.sdots = sdots;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = dispatch_heads(a.H, [&](auto Hc) {
        hipLaunchKernelGGL((gat3_fwd_kernel<decltype(Hc)::value>), dim3(a.B), dim3(GAT_THREADS), 0, s, a); })) return rc;
  MMF_CHECK_LAUNCH("mmf_gat3_dense_fwd");
  return MMF_OK;
}

extern "C" int mmf_gat3_dense_bwd(const float* h, const float* att_src, const float* att_dst, const void* y_bf16,
                                  const float* alpha, const float* sdots, const void* dy_bf16, const void* dpooled_bf16,
                                  float* dh, float* datt_src, float* datt_dst, float* dbias,
                                  const mmf_gat3_params* p, void* stream) {
  Gat3Args a = {};
  if (int rc = fill_gat(a, p)) return rc;
  if (!h || !att_src || !att_dst || !y_bf16 || !alpha || !sdots || !dh || !datt_src || !datt_dst || !dbias || (!dy_bf16 && !dpooled_bf16))
    MMF_FAIL(MMF_E_SHAPE, "mmf_gat3_dense_bwd: null operand");
  a.h = h; a.att_src = att_src; a.att_dst = att_dst;
  a.y = const_cast<unsigned short*>(static_cast<const unsigned short*>(y_bf16));
  a.alpha = const_cast<float*>(alpha); a.sdots = const_cast<float*>(sdots);
  a.dy = static_cast<const unsigned short*>(dy_bf16); a.dpool = static_cast<const unsigned short*>(dpooled_bf16);
  a.dh = dh; a.datt_src = datt_src; a.datt_dst = datt_dst; a.dbias = dbias;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = dispatch_heads(a.H, [&](auto Hc) {
        hipLaunchKernelGGL((gat3_bwd_kernel<decltype(Hc)::value>), dim3(a.B), dim3(GAT_THREADS), 0, s, a); })) return rc;
  MMF_CHECK_LAUNCH("mmf_gat3_dense_bwd");
  return MMF_OK;
}

static int nce_fill(NceArgs& a, const char* who, int B, int D, float temperature) {
  if (B <= 0 || B > NCE_MAXB || D <= 0 || (D & 3)) MMF_FAIL(MMF_E_UNSUPPORTED, "%s: B=%d (1..%d) D=%d (multiple of 4)", who, B, NCE_MAXB, D);
  if (!(temperature > 0.f)) MMF_FAIL(MMF_E_SHAPE, "%s: temperature must be positive", who);
  a.B = B; a.D = D; a.inv_temp = 1.f / temperature;
  return MMF_OK;
}

extern "C" int mmf_infonce_fwd(const float* const z[3], float* const n[3], float* inv_norm, float* losses, float* lse,
                               int B, int D, float temperature, void* stream) {
  NceArgs a = {};
  if (int rc = nce_fill(a, "mmf_infonce_fwd", B, D, temperature)) return rc;
  for (int m = 0; m < 3; ++m) {
    if (!z[m] || !n[m] || !mmf_aligned16(z[m]) || !mmf_aligned16(n[m])) MMF_FAIL(MMF_E_ALIGN, "mmf_infonce_fwd: null or misaligned operand");
    a.z[m] = z[m]; a.n[m] = n[m];
  }
  if (!inv_norm || (losses && !lse)) MMF_FAIL(MMF_E_SHAPE, "mmf_infonce_fwd: null statistics buffer");
  a.inv_norm = inv_norm; a.losses = losses; a.lse = lse;
  hipLaunchKernelGGL(nce_fwd_kernel, dim3(3), dim3(NCE_THREADS), B * B * sizeof(float), static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_infonce_fwd");
  return MMF_OK;
}

extern "C" int mmf_infonce_bwd(const float* const n[3], const float* inv_norm, const float* lse, const float* const dn[3],
                               const float* const dloss[3], float* const dz[3], int B, int D, float temperature, void* stream) {
  NceArgs a = {};
  if (int rc = nce_fill(a, "mmf_infonce_bwd", B, D, temperature)) return rc;
  bool any_loss = false;
  for (int m = 0; m < 3; ++m) {
    if (!n[m] || !dz[m]) MMF_FAIL(MMF_E_SHAPE, "mmf_infonce_bwd: null operand");
    a.n[m] = const_cast<float*>(n[m]); a.dn[m] = dn ? dn[m] : nullptr; a.dloss[m] = dloss ? dloss[m] : nullptr; a.dz[m] = dz[m];
    any_loss = any_loss || a.dloss[m];
  }
  if (!inv_norm || (any_loss && !lse)) MMF_FAIL(MMF_E_SHAPE, "mmf_infonce_bwd: null statistics buffer");
  a.inv_norm = const_cast<float*>(inv_norm); a.lse = const_cast<float*>(lse);
  hipLaunchKernelGGL(nce_bwd_kernel, dim3(3, (B + NCE_WAVES - 1) / NCE_WAVES), dim3(NCE_THREADS), 2 * B * B * sizeof(float),
                     static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_infonce_bwd");
  return MMF_OK;
}

extern "C" int mmf_adaptive_combine_fwd(const float* hp, const float* W2, const float* b2, const float* attended,
                                        float* aw, void* weighted_bf16, int B, int d, void* stream) {
  if (!hp || !W2 || !b2 || !attended || !aw || !weighted_bf16 || B <= 0 || d <= 0) MMF_FAIL(MMF_E_SHAPE, "mmf_adaptive_combine_fwd: bad operand");
  AdaArgs a = {};
  a.hp = hp; a.W2 = W2; a.b2 = b2; a.attended = attended; a.aw = aw; a.weighted = static_cast<unsigned short*>(weighted_bf16);
  a.B = B; a.d = d;
  hipLaunchKernelGGL(ada_fwd_kernel, dim3(B), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_adaptive_combine_fwd");
  return MMF_OK;
}

extern "C" int mmf_adaptive_combine_bwd(const float* hp, const float* W2, const float* attended, const float* aw,
                                        const void* dweighted_bf16, const float* daw, float* dattended, float* dhp,
                                        float* dW2, float* db2, int B, int d, void* stream) {
  if (!hp || !W2 || !attended || !aw || !dattended || !dhp || !dW2 || !db2 || B <= 0 || d <= 0)
    MMF_FAIL(MMF_E_SHAPE, "mmf_adaptive_combine_bwd: bad operand");
  AdaArgs a = {};
  a.hp = hp; a.W2 = W2; a.attended = attended; a.aw = const_cast<float*>(aw);
  a.dweighted = static_cast<const unsigned short*>(dweighted_bf16); a.daw_ext = daw;
  a.dattended = dattended; a.dhp = dhp; a.dW2 = dW2; a.db2 = db2; a.B = B; a.d = d;
  hipLaunchKernelGGL(ada_bwd_kernel, dim3(B), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_adaptive_combine_bwd");
  return MMF_OK;
}

extern "C" int mmf_linear_narrow_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, void* stream) {
  if (!x || !W || !y || M <= 0 || K <= 0 || N <= 0 || N > NARROW_MAXN) MMF_FAIL(MMF_E_SHAPE, "mmf_linear_narrow_fwd: M=%d N=%d (1..%d) K=%d", M, N, NARROW_MAXN, K);
  NarrowArgs a = {};
  a.x = x; a.W = W; a.b = b; a.y = y; a.M = M; a.N = N; a.K = K;
  hipLaunchKernelGGL(narrow_fwd_kernel, dim3(M), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_linear_narrow_fwd");
  return MMF_OK;
}

extern "C" int mmf_linear_narrow_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db,
                                     int M, int N, int K, void* stream) {
  if (!x || !W || !dy || !dW || M <= 0 || K <= 0 || N <= 0 || N > NARROW_MAXN) MMF_FAIL(MMF_E_SHAPE, "mmf_linear_narrow_bwd: M=%d N=%d (1..%d) K=%d", M, N, NARROW_MAXN, K);
  NarrowArgs a = {};
  a.x = x; a.W = W; a.dy = dy; a.dx = dx; a.dW = dW; a.db = db; a.M = M; a.N = N; a.K = K;
  hipLaunchKernelGGL(narrow_bwd_kernel, dim3((K + SM_THREADS - 1) / SM_THREADS), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MMF_CHECK_LAUNCH("mmf_linear_narrow_bwd");
  return MMF_OK;
}

extern "C" int mmf_stack3_embed_fwd(const float* f0, const float* f1, const float* f2, const float* emb, void* x_bf16,
                                    int B, int d, int ldf, void* stream) {
  if (!f0 || !f1 || !f2 || !x_bf16 || B <= 0 || d <= 0 || ldf < d) MMF_FAIL(MMF_E_SHAPE, "mmf_stack3_embed_fwd: bad operand");
  const int64_t n = (int64_t)B * 3 * d;
  const int grid = (int)((n + SM_THREADS - 1) / SM_THREADS < 2048 ? (n + SM_THREADS - 1) / SM_THREADS : 2048);
  hipLaunchKernelGGL(stack3_fwd_kernel, dim3(grid), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream), f0, f1, f2, emb,
                     static_cast<unsigned short*>(x_bf16), B, d, ldf);
  MMF_CHECK_LAUNCH("mmf_stack3_embed_fwd");
  return MMF_OK;
}

extern "C" int mmf_stack3_embed_bwd(const void* dx_bf16, float* d0, float* d1, float* d2, float* demb, int B, int d, int ldd,
                                    void* stream) {
  if (!dx_bf16 || B <= 0 || d <= 0 || ldd < d) MMF_FAIL(MMF_E_SHAPE, "mmf_stack3_embed_bwd: bad operand");
  hipLaunchKernelGGL(stack3_bwd_kernel, dim3((3 * d + SM_THREADS - 1) / SM_THREADS), dim3(SM_THREADS), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(dx_bf16), d0, d1, d2, demb, B, d, ldd);
  MMF_CHECK_LAUNCH("mmf_stack3_embed_bwd");
  return MMF_OK;
}

extern "C" int mmf_rowmask_apply(const float* x, const float* mask, float* y, int B, int d, void* stream) {
  if (!x || !mask || !y || B <= 0 || d <= 0) MMF_FAIL(MMF_E_SHAPE, "mmf_rowmask_apply: bad operand");
  const int64_t n = (int64_t)B * d;
  const int grid = (int)((n + SM_THREADS - 1) / SM_THREADS < 2048 ? (n + SM_THREADS - 1) / SM_THREADS : 2048);
  hipLaunchKernelGGL(rowmask_kernel, dim3(grid), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream), x, mask, y, B, d);
  MMF_CHECK_LAUNCH("mmf_rowmask_apply");
  return MMF_OK;
}

extern "C" int mmf_attn_weights_mean(const void* qkv_bf16, float* w, int B, int T, int heads, int head_dim, void* stream) {
  if (!qkv_bf16 || !w || B <= 0 || T <= 0 || T > SM_THREADS * AWM_MAXK || heads <= 0 || head_dim <= 0 || (head_dim & 7) ||
      B > 65535 || heads * head_dim * 4 > 48 * 1024 || !mmf_aligned16(qkv_bf16))
    MMF_FAIL(MMF_E_SHAPE, "mmf_attn_weights_mean: B=%d T=%d (<= %d) heads=%d head_dim=%d (multiple of 8)", B, T,
             SM_THREADS * AWM_MAXK, heads, head_dim);
  hipLaunchKernelGGL(attn_weights_mean_kernel, dim3(T, B), dim3(SM_THREADS), (size_t)heads * head_dim * 4,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(qkv_bf16), w, T, heads, head_dim,
                     1.f / sqrtf((float)head_dim));
  MMF_CHECK_LAUNCH("mmf_attn_weights_mean");
  return MMF_OK;
}

extern "C" int mmf_adaptive_attn_weights(const void* qkv_bf16, float* w, int B, int heads, int head_dim, void* stream) {
  if (!qkv_bf16 || !w || B <= 0 || heads <= 0 || heads > ADAW_MAXH || head_dim <= 0)
    MMF_FAIL(MMF_E_SHAPE, "mmf_adaptive_attn_weights: B=%d heads=%d (1..%d) head_dim=%d", B, heads, ADAW_MAXH, head_dim);
  hipLaunchKernelGGL(ada_attn_weights_kernel, dim3(B), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(qkv_bf16), w, heads, head_dim, 1.f / sqrtf((float)head_dim));
  MMF_CHECK_LAUNCH("mmf_adaptive_attn_weights");
  return MMF_OK;
}
