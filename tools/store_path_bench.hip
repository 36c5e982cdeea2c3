// What a CU's store path does with the shapes a GEMM epilogue can emit (round 4): 256 workgroups x 4 waves (one per SIMD, as in
// gemm6 / gemm7), every wave stores a 128 x 128 bf16 quadrant (32 KiB) of a [rows][ld] matrix per iteration, 16 bytes per lane
// and instruction, in one of these shapes:
//   0  row per lane, as the accumulator layout gives it: lane l -> row (l & 31), 16 B at column 16 * (l >> 5)  (32 rows x 32 B per instruction)
//   1  whole rows: lane l -> row (l >> 4), 16 B at column 8 * (l & 15)  (4 rows x 256 B per instruction: after an LDS transposition)
//   2  as 1 with 128-B row pieces: lane l -> row (l >> 3), 16 B at 8 * (l & 7)  (8 rows x 128 B)
//   3  as 0 with 8-byte stores (16 rows... the un-widened form: 32 rows x 2 x 8 B)
// hipcc --offload-arch=gfx950 -O3 tools/store_path_bench.hip -o tools/bin/store_path_bench && tools/bin/store_path_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(unsigned short* C, int ld, int tiles_n, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const u32x4_t v = {(unsigned)lane, 1u, 2u, 3u};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int tile = blockIdx.x + it * gridDim.x;
    const int m0 = (tile / tiles_n) * 256 + 128 * wm, n0 = (tile % tiles_n) * 256 + 128 * wn;
    unsigned short* base = C + (size_t)m0 * ld + n0;
    if (SHAPE == 0) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int c = 0; c < 8; ++c)          // 8 x 32-byte column groups of the 128 columns
          *reinterpret_cast<u32x4_t*>(base + (size_t)(32 * tm + (lane & 31)) * ld + 16 * c + 8 * (lane >> 5)) = v;
    } else if (SHAPE == 1) {
#pragma unroll
      for (int r = 0; r < 32; ++r)
        *reinterpret_cast<u32x4_t*>(base + (size_t)(4 * r + (lane >> 4)) * ld + 8 * (lane & 15)) = v;
    } else if (SHAPE == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          *reinterpret_cast<u32x4_t*>(base + (size_t)(8 * r + (lane >> 3)) * ld + 64 * c + 8 * (lane & 7)) = v;
    } else {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int c = 0; c < 16; ++c)
          *reinterpret_cast<u32x2_t*>(base + (size_t)(32 * tm + (lane & 31)) * ld + 8 * c + 4 * (lane >> 5)) = u32x2_t{v[0], v[1]};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int M = 8192, N = 8192, ld = N;                    // 1024 tiles of 256 x 256: 4 per workgroup
  unsigned short* C; unsigned long long* cyc;
  hipMalloc(&C, (size_t)M * ld * 2); hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 0; shape < 4; ++shape)
    for (int grid : {256, 32}) {
      const int iters = 4;
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) k<0><<<grid, 256>>>(C, ld, N / 256, iters, cyc);
        if (shape == 1) k<1><<<grid, 256>>>(C, ld, N / 256, iters, cyc);
        if (shape == 2) k<2><<<grid, 256>>>(C, ld, N / 256, iters, cyc);
        if (shape == 3) k<3><<<grid, 256>>>(C, ld, N / 256, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
      const double bytes = (double)grid * iters * 131072;
      printf("shape %d grid %3d: %7.1f us  %6.2f TB/s  %8.0f cycles per workgroup = %5.1f B/clk/CU, %6.0f cycles per 128-KiB tile\n", shape, grid,
             best * 1e3, bytes / best / 1e9, avg, iters * 131072.0 / avg, avg / iters);
    }
  return 0;
}
