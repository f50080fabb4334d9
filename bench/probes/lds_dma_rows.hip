// What does the SHAPE of an LDS-DMA request cost?  Every wave-instruction (buffer_load_dwordx4 ... lds) moves 1 KiB;
// the conv kernels fetch it as R rows of 1024 / R bytes from R different pixels (16 x 64 B with BK = 32, 8 x 128 B with
// BK = 64, 4 x 256 B in the filter-gradient kernel), the bandwidth probe (l2_lds_bw.hip) as ONE contiguous KiB.
// Same bytes, same L2-resident buffer, 16 waves per CU, 8 instructions in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_rows lds_dma_rows.hip ; run: ./lds_dma_rows
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory");
}

// rows_log2: the KiB of one instruction is 2^rows_log2 rows; row r of slice s starts at (s * R + r) * pitch (bytes).
// reads_per_iter: ds_read_b128 per lane issued between the DMA batches (the fragment reads of a conv K-step).
__global__ __launch_bounds__(256) void rows_kernel(const char* __restrict__ buf, unsigned bytes, int iters, int rows_log2,
                                                   unsigned pitch, int reads_per_iter, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 8 * 1024];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned waves = gridDim.x * 4u, me = blockIdx.x * 4u + (unsigned)wid;
  const unsigned R = 1u << rows_log2, row_bytes = 1024u >> rows_log2;
  const unsigned lpr = row_bytes / 16u;                       // lanes per row
  const unsigned lrow = (unsigned)lane / lpr, lcol = ((unsigned)lane % lpr) * 16u;
  const unsigned slices = bytes / (R * pitch);                // a slice = R rows of the buffer
  const unsigned long long a = (unsigned long long)buf;
  u32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  rsrc[2] = __builtin_amdgcn_readfirstlane(bytes);
  rsrc[3] = 0x00020000u;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + (unsigned)wid * 8192u;
  unsigned s = me % slices;
  u32x4 accv = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // column window: successive instructions of a wave walk the columns of the same rows (the K loop of a conv)
      const unsigned col0 = ((unsigned)(it * 8 + j) * row_bytes) % pitch;
      lds_dma16(rsrc, lds0 + (unsigned)j * 1024u, (s * R + lrow) * pitch + ((col0 + lcol) % pitch));
    }
    for (int r = 0; r < reads_per_iter; ++r) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((threadIdx.x * 16 + r * 4096) & (32 * 1024 - 1)));
      accv ^= v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((it & 31) == 31) { s += waves; if (s >= slices) s %= slices; }
  }
  const unsigned acc = accv[0] ^ accv[1] ^ accv[2] ^ accv[3];
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  unsigned* sink; hipMalloc(&sink, 64);
  const int grid = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t bytes = 24ul << 20;                 // 24 MiB: stays in the L2s + memory-side cache
  char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
  const int iters = 256;
  const double moved = (double)grid * 4 * iters * 8192.0;
  printf("LDS-DMA, 1 KiB per wave-instruction as R rows (chip-wide TB/s; bytes per clock per CU at 2.4 GHz)\n");
  printf("%6s %10s %8s %12s | %12s %12s\n", "rows", "row bytes", "pitch", "no ds_read", "+8 ds_read", "+16 ds_read");
  for (int rl = 0; rl <= 4; ++rl) {
    for (unsigned pitch : {1024u, 4096u}) {
      if ((1024u >> rl) > pitch) continue;
      double tb[3];
      int k = 0;
      for (int reads : {0, 8, 16}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(rows_kernel, dim3(grid), dim3(256), 0, 0, buf, (unsigned)bytes, iters, rl, pitch, reads, sink);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        tb[k++] = moved / best * 1e-9;
      }
      printf("%6d %10u %8u %7.2f (%4.1f) | %7.2f (%4.1f) %7.2f (%4.1f)\n", 1 << rl, 1024u >> rl, pitch, tb[0],
             tb[0] * 1e12 / 256 / 2.4e9, tb[1], tb[1] * 1e12 / 256 / 2.4e9, tb[2], tb[2] * 1e12 / 256 / 2.4e9);
    }
  }
  // ---- bytes in flight: waves per CU x 8 KiB each (every wave waits for its own 8 instructions before the next batch)
  printf("\nbytes in flight per CU (8 x 128-byte rows per instruction, pitch 1024): blocks of 256 threads per CU x 32 KiB\n");
  printf("%12s %12s %14s %22s\n", "blocks / CU", "KiB in flight", "TB/s (B/clk/CU)", "latency = inflight/rate");
  for (int bpc : {1, 2, 3, 4}) {
    const int g = 256 * bpc;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(rows_kernel, dim3(g), dim3(256), 0, 0, buf, (unsigned)bytes, iters, 3, 1024u, 0, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double tbs = (double)g * 4 * iters * 8192.0 / best * 1e-9;
    const double bpcc = tbs * 1e12 / 256 / 2.4e9;
    printf("%12d %12d %8.2f (%4.1f) %16.0f clocks\n", bpc, bpc * 32, tbs, bpcc, bpc * 32768.0 / bpcc);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
  return 0;
}
