// Do the matrix pipe and the vector ALU of one SIMD run side by side when they are fed by TWO DIFFERENT waves? (MI355X)
// 512-thread workgroups, one per CU (100 KiB of LDS each); waves w and w + 4 share a SIMD.  Role A (waves 0-3) issues only
// v_mfma_f32_32x32x16_bf16 (four independent accumulators), role B (waves 4-7) only VALU (16 independent fma -> exp chains).
// Timed: A alone, B alone, both.  "both ~ max" = the pipes overlap across waves; "both ~ sum" = they serialise.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/coexec_bench tools/coexec_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ unsigned long long g_cyc[2][4096];        // [role][workgroup * 4 + wave]: shader cycles (s_memtime) of the loop

template <int MODE, int VARIANT>
__global__ __launch_bounds__(512, 2) void k(int iters, float* out) {
  __shared__ char pad[100 * 1024];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x == 0) pad[0] = 1;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const bool roleA = wave < 4;
  if (roleA && (MODE & 1)) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[blockIdx.x] = s;
    if (lane == 0) g_cyc[0][blockIdx.x * 4 + wave] = __builtin_readcyclecounter() - t0;
  }
  if (!roleA && (MODE & 2)) {
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = 0.001f * (lane + r);
    const float c = 0.999f, d = -0.0001f * lane;
    for (int it = 0; it < iters; ++it) {
      if (VARIANT == 0) {            // per iteration 64 v_fma + 16 v_exp
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = __builtin_fmaf(v[r], c, d);
          t = __builtin_fmaf(t, c, d);
          t = __builtin_fmaf(t, c, d);
          t = __builtin_fmaf(t, c, d);
          v[r] = __builtin_amdgcn_exp2f(t * 0.001f) * 0.5f;
        }
      } else {                       // plain VALU only: 96 v_fma
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = v[r];
#pragma unroll
          for (int q = 0; q < 6; ++q) t = __builtin_fmaf(t, c, d);
          v[r] = t;
        }
      }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += v[r];
    if (s == 123.456f) out[blockIdx.x] = s;
    if (lane == 0) g_cyc[1][blockIdx.x * 4 + wave - 4] = __builtin_readcyclecounter() - t0;
  }
}

// ONE wave per SIMD carrying both streams: per MFMA, FILL independent v_fma (and one v_exp when EXP) placed behind it in
// program order (sched_group_barrier pins the interleave).
template <int FILL, bool EXP>
__global__ __launch_bounds__(512, 2) void k1(int iters, float* out) {
  __shared__ char pad[100 * 1024];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x == 0) pad[0] = 1;
  if (wave >= 4) return;
  const unsigned long long t0 = __builtin_readcyclecounter();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
  float v[16];
  for (int r = 0; r < 16; ++r) v[r] = 0.001f * (lane + r);
  const float c = 0.999f, d = -0.0001f * lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      float t = v[u];
#pragma unroll
      for (int q = 0; q < FILL; ++q) t = __builtin_fmaf(t, c, d);
      if (EXP) t = __builtin_amdgcn_exp2f(t);
      v[u] = t;
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, FILL + (EXP ? 1 : 0), 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int r = 0; r < 16; ++r) s += v[r];
  if (s == 123.456f) out[blockIdx.x] = s;
  if (lane == 0) g_cyc[0][blockIdx.x * 4 + wave] = __builtin_readcyclecounter() - t0;
}

template <typename F>
static double time_us(F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3;
}

int main() {
  float* out; (void)hipMalloc(&out, 4096 * 4);
  const int iters = 4000;
  auto run = [&](const char* name, auto kern) {
    const double us = time_us([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, iters, out); });
    static unsigned long long h[2][4096];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cyc), sizeof(h));
    double ca = 0, cb = 0;
    for (int i = 0; i < 1024; ++i) { ca += h[0][i]; cb += h[1][i]; }
    printf("%-44s %9.1f us | s_memtime cycles per iteration: role A %7.1f  role B %7.1f | implied clock %4.2f GHz\n", name, us,
           ca / 1024 / iters, cb / 1024 / iters, (ca > cb ? ca : cb) / 1024 / us / 1e3);
    (void)hipMemset(nullptr, 0, 0);
    static unsigned long long z[2][4096];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof(z));
    return us;
  };
  printf("per iteration: role A 16 MFMA 32x32x16 (512 cycles of the matrix pipe); role B 64 fma + 16 exp + 32 mul, or 96 fma\n");
  run("A alone (MFMA)", k<1, 0>);
  run("B alone (fma + exp)", k<2, 0>);
  run("A and B together (fma + exp)", k<3, 0>);
  run("B alone (fma only)", k<2, 1>);
  run("A and B together (fma only)", k<3, 1>);
  printf("one wave per SIMD, 16 MFMA per iteration with vector fillers behind each MFMA in the SAME stream:\n");
  run("  0 fillers", k1<0, false>);
  run("  2 fma per MFMA", k1<2, false>);
  run("  4 fma per MFMA", k1<4, false>);
  run("  6 fma per MFMA", k1<6, false>);
  run("  4 fma + 1 exp per MFMA", k1<4, true>);
  run("  8 fma per MFMA", k1<8, false>);
  return 0;
}
