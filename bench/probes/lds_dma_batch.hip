// Latency of a STAGE: one block of 8 waves per CU (the big-tile conv kernel's shape) fetches a batch of 64 KiB by LDS-DMA
// (every wave 8 instructions of 8 rows x 128 bytes), waits for ALL of it (vmcnt + barrier) and goes on - with 1, 2 or 3
// batches in flight, from buffers of different size (L2-resident ... HBM).  clocks per batch = what a K-stage costs when
// nothing but the loads is in the way.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_batch lds_dma_batch.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory");
}

template <int DEPTH>     // batches in flight
__global__ __launch_bounds__(512) void batch_kernel(const char* __restrict__ buf, unsigned bytes, int iters, unsigned reuse,
                                                    long long* clocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // DEPTH x 64 KiB... capped at 2 slots of 64 KiB
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long a = (unsigned long long)buf;
  u32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  rsrc[2] = __builtin_amdgcn_readfirstlane(bytes);
  rsrc[3] = 0x00020000u;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  const unsigned nbatch = bytes >> 16;                               // 64 KiB batches in the buffer
  // `reuse` consecutive blocks read the same batches (the N-tiles of one M-tile read the same activation rows)
  unsigned b = (blockIdx.x / reuse) % nbatch;
  const unsigned bstep = (gridDim.x + reuse - 1) / reuse;
  auto issue = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // wave w, instruction j: rows (w*8 + j)*8 .. +7 of the batch's 512 rows of 128 bytes
      const unsigned row = (unsigned)(wid * 8 + j) * 8u + (unsigned)(lane >> 3);
      lds_dma16(rsrc, lds0 + (unsigned)(slot & 1) * 65536u + (unsigned)(wid * 8 + j) * 1024u,
                (b << 16) + row * 128u + (unsigned)(lane & 7) * 16u);
    }
    b += bstep; if (b >= nbatch) b %= nbatch;
  };
  for (int d = 0; d < DEPTH - 1; ++d) issue(d);
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    issue(it + DEPTH - 1);
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

int main() {
  long long* clocks; hipMalloc(&clocks, 256 * 8);
  long long host[256];
  const int iters = 2000;
  hipFuncSetAttribute((const void*)batch_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)batch_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)batch_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  printf("one block of 8 waves per CU, 64 KiB per batch; 100 MHz wall clock ticks converted to 2.4 GHz shader clocks\n");
  printf("%10s %6s | %28s | %28s | %28s\n", "buffer", "reuse", "1 in flight: clk/batch B/clk", "2 in flight", "3 in flight");
  for (size_t mb : {16ul, 64ul, 512ul, 4096ul}) {
    const size_t bytes = mb << 20;
    char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    for (unsigned reuse : {1u, 2u, 8u}) {
      printf("%7zu MB %6u |", mb, reuse);
      for (int depth = 1; depth <= 3; ++depth) {
        for (int rep = 0; rep < 2; ++rep) {
          if (depth == 1) hipLaunchKernelGGL(batch_kernel<1>, dim3(256), dim3(512), 131072, 0, buf, (unsigned)(bytes > 0xFFFF0000ul ? 0xFFFF0000ul : bytes), iters, reuse, clocks);
          if (depth == 2) hipLaunchKernelGGL(batch_kernel<2>, dim3(256), dim3(512), 131072, 0, buf, (unsigned)(bytes > 0xFFFF0000ul ? 0xFFFF0000ul : bytes), iters, reuse, clocks);
          if (depth == 3) hipLaunchKernelGGL(batch_kernel<3>, dim3(256), dim3(512), 131072, 0, buf, (unsigned)(bytes > 0xFFFF0000ul ? 0xFFFF0000ul : bytes), iters, reuse, clocks);
          hipDeviceSynchronize();
        }
        hipMemcpy(host, clocks, sizeof(host), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)host[i]; mean /= 256;
        const double clk = mean / iters * 24.0;                  // 100 MHz ticks -> 2.4 GHz clocks
        printf(" %12.0f %13.1f |", clk, 65536.0 / clk);
      }
      printf("\n");
    }
    hipFree(buf);
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
