// L2 -> CU fill-rate microbenchmark (MI355X): how many bytes per clock one CU can pull from its XCD's L2
//   (a) with LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) and
//   (b) with plain buffer_load_dwordx4 into registers,
// every CU streaming over the same small (L2-resident) region.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/ldsdma_bw tools/ldsdma_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void k_dma(const char* buf, unsigned region, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) char smem[8 * DEPTH * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(buf), 0, (int)region, 0x00020000);
  unsigned off = ((blockIdx.x * nw + wave) * 1024u * DEPTH) % region;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + (wave * DEPTH + d) * 1024), 16, lane * 16u, off, 0, 0);
      off += 1024u; if (off >= region) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = smem[0];
}

template <int DEPTH>
__global__ __launch_bounds__(512) void k_reg(const char* buf, unsigned region, int iters, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(buf), 0, (int)region, 0x00020000);
  unsigned off = ((blockIdx.x * nw + wave) * 1024u * DEPTH) % region;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, off, 0);
      off += 1024u; if (off >= region) off = 0;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
  }
  if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x] = 1;
}

template <typename F>
static double time_us(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3;
}

int main() {
  const unsigned region = 1u << 20;   // 1 MiB: resident in every XCD's 4 MiB L2
  char* buf; unsigned* sink;
  hipMalloc(&buf, region); hipMemset(buf, 1, region); hipMalloc(&sink, 4096 * 4);
  const int iters = 2000;
  for (int threads : {256, 512}) {
    for (int blocks : {256, 512}) {
      const int nw = threads / 64;
      auto report = [&](const char* name, int depth, double us) {
        const double bytes = (double)blocks * nw * depth * 1024.0 * iters;
        printf("%-8s threads %3d blocks %3d depth %d : %8.1f us  %7.2f TB/s  = %5.1f B/clk/CU at 2.4 GHz\n", name, threads, blocks, depth, us,
               bytes / us / 1e6, bytes / us / 1e6 * 1e12 / 256 / 2.4e9 / 1.0);
      };
      report("lds-dma", 4, time_us([&] { hipLaunchKernelGGL(k_dma<4>, dim3(blocks), dim3(threads), 0, 0, buf, region, iters, sink); }));
      report("lds-dma", 8, time_us([&] { hipLaunchKernelGGL(k_dma<8>, dim3(blocks), dim3(threads), 0, 0, buf, region, iters, sink); }));
      report("regs", 4, time_us([&] { hipLaunchKernelGGL(k_reg<4>, dim3(blocks), dim3(threads), 0, 0, buf, region, iters, sink); }));
      report("regs", 8, time_us([&] { hipLaunchKernelGGL(k_reg<8>, dim3(blocks), dim3(threads), 0, 0, buf, region, iters, sink); }));
    }
  }
  return 0;
}
