// How fast can the whole chip pull L2-resident (or Infinity-Cache- / HBM-resident) data into LDS with LDS-DMA
// (buffer_load_dwordx4 ... lds), and into registers with plain 16-byte loads?  The im2col-style staging of the conv
// kernels re-reads every activation once per filter tap THROUGH L2, so this number is their ceiling.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_lds_bw l2_lds_bw.hip ; run: ./l2_lds_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory");
}

// every wave streams 1 KiB slices (64 lanes x 16 B) of a buffer of `bytes` bytes, 8 slices in flight, `iters` rounds
template <bool DMA>
__global__ __launch_bounds__(256) void pull_kernel(const char* __restrict__ buf, unsigned bytes, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 8 * 1024];      // 8 KiB per wave
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned waves = gridDim.x * 4u, me = blockIdx.x * 4u + (unsigned)wid;
  const unsigned slices = bytes >> 10;
  const unsigned long long a = (unsigned long long)buf;
  u32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  rsrc[2] = __builtin_amdgcn_readfirstlane(bytes);
  rsrc[3] = 0x00020000u;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * 8192u;
  unsigned s = me % slices;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (DMA) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lds_dma16(rsrc, lds0 + (unsigned)j * 1024u, (s << 10) + (unsigned)lane * 16u);
        s += waves; if (s >= slices) s -= slices * (s / slices);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = *reinterpret_cast<const u32x4*>(buf + ((size_t)s << 10) + (size_t)lane * 16);
        s += waves; if (s >= slices) s -= slices * (s / slices);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
  }
  if (DMA) acc = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  unsigned* sink; hipMalloc(&sink, 64);
  const int grid = 256 * 4;                       // 4 blocks (16 waves, 128 KiB in flight) per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%10s %14s %14s\n", "buffer", "LDS-DMA TB/s", "registers TB/s");
  for (size_t mb : {1ul, 4ul, 16ul, 32ul, 64ul, 128ul, 512ul, 2048ul}) {
    const size_t bytes = mb << 20;
    char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    const int iters = 256;                        // 256 rounds x 8 KiB per wave = 2 MiB per wave, 8 GiB per launch
    const double moved = (double)grid * 4 * iters * 8192.0;
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 4; ++rep) {
      for (int k = 0; k < 2; ++k) {
        hipEventRecord(e0);
        if (k == 0) hipLaunchKernelGGL(pull_kernel<true>, dim3(grid), dim3(256), 0, 0, buf, (unsigned)bytes, iters, sink);
        else        hipLaunchKernelGGL(pull_kernel<false>, dim3(grid), dim3(256), 0, 0, buf, (unsigned)bytes, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best[k]) best[k] = ms;
      }
    }
    printf("%7zu MB %14.2f %14.2f\n", mb, moved / best[0] * 1e-9, moved / best[1] * 1e-9);
    hipFree(buf);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
  return 0;
}
