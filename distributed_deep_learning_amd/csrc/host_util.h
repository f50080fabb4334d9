// Host-side helpers shared by the launchers.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

static inline FastDiv make_fastdiv(unsigned int d) {
  FastDiv f;
  f.d = d;
  if (d <= 1u) { f.magic = 0u; f.shift = 0u; return f; }
  unsigned int s = 0;
  while ((1ull << s) < (unsigned long long)d) ++s;           // s = ceil(log2 d), s >= 1
  const unsigned long long num = 1ull << (31 + s);
  f.magic = (unsigned int)((num + d - 1) / d);                // < 2^32 because 2^(s-1) < d
  f.shift = s - 1;
  return f;
}

// Dynamic-LDS opt-in, once per (kernel, device).
struct SmemAttrOnce {
  bool done[64] = {false};
  int ensure(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) dev = 0;
    if (done[dev]) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done[dev] = true;
    return 0;
  }
};

static inline unsigned int grid_for(long long work_items, int per_block, int max_blocks = 2048) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (unsigned int)b;
}

// Zero-fill as a KERNEL, never hipMemsetAsync: inside a captured hipGraph a memset NODE on a small buffer of the
// graph's private pool went wrong after torch.cuda.empty_cache() (replays produced garbage; found with LeNet-5's
// padded-bias path, the only one that zero-filled through a memset while captured).  A fill kernel is an ordinary
// kernel node.  `bytes` must be a multiple of 4.
static __global__ void ddl_zero_kernel(unsigned int* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t ddl_zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(ddl_zero_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, st, (unsigned int*)p, n);
  return hipGetLastError();
}
